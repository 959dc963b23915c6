#!/usr/bin/env python3
"""Per-problem-shape breakdown of the GEMM launches of ONE training iteration (development aid).

    python benchmarks/step_breakdown.py [per_gpu_batch] [c1|c2]

Runs a few warm-up iterations of the bench workload (128x128; c1 fp32, c2 bf16), then one iteration with HIP
events around every conv C-ABI call, and prints time / launches / TFLOP/s per (kind, shape),
largest first.  Tells which layer shapes hold the GEMM roofline fraction down.
"""
import contextlib
import io
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench  # noqa: E402
from hipdwc import ops, host, synth  # noqa: E402

PEAK_TF = 157.3


def main():
    global PEAK_TF
    conf = sys.argv[2] if len(sys.argv) > 2 else "c1"
    B = int(sys.argv[1]) if len(sys.argv) > 1 else bench.CONFIGS[conf]["per_gpu_batch"]
    ops.set_precision(bench.CONFIGS[conf]["precision"])
    PEAK_TF = bench.MFMA_PEAK_TFLOPS[bench.CONFIGS[conf]["precision"]]
    dev = torch.device("cuda:0")
    from solver import Solver
    cfg = synth.make_config(image_size=128)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    host.set_noise(host.DeviceNoise())
    batch = synth.make_batch(B, 128, seed=1, device=dev)
    batch["txt_lens"] = batch["txt_lens"].cpu()
    for it in range(3):
        bench.run_iteration(trainer, batch, cfg, it)
    torch.cuda.synchronize()
    import ctypes
    from hipdwc import _lib
    lib = _lib.load()
    probe = getattr(lib, "dwc_debug_clock_probe", None)      # only in `make PROBE=1` builds
    buf = (ctypes.c_ulonglong * 2)()
    if probe is not None:
        probe(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(3, 8):
        bench.run_iteration(trainer, batch, cfg, it)
    b.record()
    torch.cuda.synchronize()
    step_ms = a.elapsed_time(b) / 5
    if probe is not None:
        probe(buf, 0)
        print("average shader clock seen by workgroup 0 of the GEMM launches: %.0f MHz" % (100.0 * buf[0] / max(buf[1], 1)))
    ops.TIMER = ops.KernelTimer()
    bench.run_iteration(trainer, batch, cfg, 4)
    timer, ops.TIMER = ops.TIMER, None
    torch.cuda.synchronize()
    rows = {}
    for (tag, flops, e0, e1), detail in zip(timer.spans, timer.details):
        ent = rows.setdefault(detail or tag, [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += e0.elapsed_time(e1)
        ent[2] += flops
    tot_ms = sum(v[1] for v in rows.values())
    tot_fl = sum(v[2] for v in rows.values())
    print("step %.1f ms (untimed run); timed GEMM spans %.1f ms, %.1f TF avg" % (step_ms, tot_ms, tot_fl / tot_ms / 1e9))
    print("%-44s %5s %8s %6s %7s %6s" % ("problem", "n", "ms", "%", "TF", "%peak"))
    for k, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        tf = fl / ms / 1e9
        print("%-44s %5d %8.3f %6.1f %7.1f %6.1f" % (k, n, ms, 100 * ms / tot_ms, tf, 100 * tf / PEAK_TF))


if __name__ == "__main__":
    main()
