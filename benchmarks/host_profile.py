#!/usr/bin/env python3
"""Where the HOST spends its time issuing one training iteration (cProfile over N iterations, GPU running behind).
    python benchmarks/host_profile.py [c1|c2] [n]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench  # noqa: E402
from hipdwc import ops, host, synth  # noqa: E402


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "c1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
    for it in range(4):
        bench.run_iteration(trainer, batch, cfg, it)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for it in range(4, 4 + n):
        bench.run_iteration(trainer, batch, cfg, it)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(28)
    print("per iteration (x %d):" % n)
    print(s.getvalue()[:7000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
    print(s.getvalue()[:7000])


if __name__ == "__main__":
    main()
