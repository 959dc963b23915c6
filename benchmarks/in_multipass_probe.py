"""Instance norm on the planes the resident kernels do not take (64x64 and 128x128), batch 16, fp32: forward + backward, 20 times.
Run under `rocprofv3 --kernel-trace --stats` to read the per-kernel times.  usage: python benchmarks/in_multipass_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    for C, H in ((128, 64), (64, 128)):
        x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for _ in range(20):
            y = ops.instance_norm(x, None, None, relu=True)
            y.backward(torch.ones_like(y))
            x.grad = None
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
