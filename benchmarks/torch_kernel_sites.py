#!/usr/bin/env python3
"""Which Python call sites launch the stock-torch kernels of one training iteration (development aid for the launch diet).

    python benchmarks/torch_kernel_sites.py [c1|c2]

Profiles ONE iteration with torch.profiler (with_stack) and prints, per (aten op, innermost frame inside this repository),
the number of GPU kernels launched and their device time.
"""
import collections
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


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "c1"
    B = bench.CONFIGS[conf]["per_gpu_batch"]
    ops.set_precision(bench.CONFIGS[conf]["precision"])
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
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        bench.run_iteration(trainer, batch, cfg, 3)
        torch.cuda.synchronize()
    sites = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        kt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        nk = len(getattr(ev, "kernels", []) or [])
        if not nk or not ev.name.startswith("aten::"):
            continue
        frame = "?"
        for fr in (ev.stack or []):
            if "/dwc-gan_amd/" in fr or "/bench.py" in fr:
                frame = fr.split("/dwc-gan_amd/")[-1] if "/dwc-gan_amd/" in fr else fr.split("/")[-1]
                break
        shp = str(getattr(ev, "input_shapes", ""))[:60]
        key = (ev.name, frame + " " + shp)
        sites[key][0] += nk
        sites[key][1] += sum(k.duration for k in ev.kernels)
    tot_n = sum(v[0] for v in sites.values())
    tot_t = sum(v[1] for v in sites.values())
    print("stock-torch kernels in one iteration: %d launches, %.2f ms" % (tot_n, tot_t / 1e3))
    for (name, frame), (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%5d  %8.1f us  %-28s %s" % (n, t, name, frame))


if __name__ == "__main__":
    main()
