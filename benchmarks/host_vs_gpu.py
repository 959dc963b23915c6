#!/usr/bin/env python3
"""Is the step host-bound?  Issues N iterations without synchronising and compares the time the host needed to ISSUE
them with the time until the GPU finished them (development aid)."""
import contextlib
import io
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench  # noqa: E402
from hipdwc import host, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128     # a small S makes the GPU fast: what remains is host overhead
    dev = torch.device("cuda:0")
    from solver import Solver
    cfg = synth.make_config(image_size=S)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    host.set_noise(host.DeviceNoise())
    batch = synth.make_batch(B, S, seed=1, device=dev)
    batch["txt_lens"] = batch["txt_lens"].cpu()
    for it in range(4):
        bench.run_iteration(trainer, batch, cfg, it)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for it in range(4, 4 + n):
        bench.run_iteration(trainer, batch, cfg, it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d S=%d: host issue %.1f ms/iteration, until GPU done %.1f ms/iteration" % (B, S, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))


if __name__ == "__main__":
    main()
