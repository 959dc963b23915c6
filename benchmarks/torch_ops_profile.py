#!/usr/bin/env python3
"""Which aten operators launch the small non-HIP-library kernels of a training iteration (development aid).

    python benchmarks/torch_ops_profile.py [per_gpu_batch]
"""
import contextlib
import io
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench  # noqa: E402
from hipdwc import host, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for it in range(3, 5):
            bench.run_iteration(trainer, batch, cfg, it)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))


if __name__ == "__main__":
    main()
