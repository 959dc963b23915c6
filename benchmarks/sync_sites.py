#!/usr/bin/env python3
"""Which Python lines make the host WAIT for the GPU inside one training iteration (torch.cuda.set_sync_debug_mode: every
synchronising call -- .item(), copies from / to pageable memory, nonzero-shaped ops -- raises a warning with its stack).

    python benchmarks/sync_sites.py [c1|c2]
"""
import collections
import contextlib
import io
import os
import sys
import traceback
import warnings

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
    sites = collections.Counter()

    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message).lower():
            return
        frame = "?"
        for fs in reversed(traceback.extract_stack(limit=30)[:-1]):
            if "/dwc-gan_amd/" in fs.filename or fs.filename.endswith("bench.py"):
                frame = "%s:%d %s" % (fs.filename.split("/repo/")[-1], fs.lineno, fs.name)
                break
        if frame == "?":
            frame = "? " + " <- ".join("%s:%d" % (fs.filename.split("/")[-1], fs.lineno) for fs in reversed(traceback.extract_stack(limit=12)[:-1]))
        sites[frame] += 1
    old = warnings.showwarning
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode(1)
    try:
        bench.run_iteration(trainer, batch, cfg, 3)
    finally:
        torch.cuda.set_sync_debug_mode(0)
        warnings.showwarning = old
    torch.cuda.synchronize()
    print("%s: host-synchronising calls in one iteration: %d" % (conf, sum(sites.values())))
    for k, n in sites.most_common():
        print("  %3d  %s" % (n, k))


if __name__ == "__main__":
    main()
