"""nn.Linear on few rows: csrc/linear_small.hip against the 1x1-convolution path it replaces (DWC_LINEAR_SMALL=0), per shape of the
AdaIN-parameter MLP / style mapping at the bench configurations.  Forward and forward+backward time per call by HIP events over 200 calls.

    python benchmarks/linear_small_bench.py
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import ops          # noqa: E402

SHAPES = [(16, 64, 256, "relu"), (16, 256, 256, "relu"), (16, 256, 4096, "none"), (48, 64, 256, "relu"), (48, 256, 4096, "none"),
          (128, 256, 256, "relu"), (128, 256, 4096, "none"), (384, 256, 4096, "none")]


def time_it(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    ops.set_precision("fp32")
    print("%-28s %12s %12s %12s %12s" % ("rows, in > out", "fwd small", "fwd 1x1", "f+b small", "f+b 1x1"))
    for M, K, N, act in SHAPES:
        x = torch.randn(M, K, device=dev, requires_grad=True)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_(True)
        b = torch.zeros(N, device=dev, requires_grad=True)
        gy = torch.randn(M, N, device=dev)
        row = []
        for small in (1, 0):
            ops.LINEAR_SMALL = small

            def fwd():
                with torch.no_grad():
                    ops.linear(x, w, b, act)

            def both():
                ops.linear(x, w, b, act).backward(gy)
                x.grad = w.grad = b.grad = None

            row.append((time_it(fwd), time_it(both)))
        print("%-28s %9.1f us %9.1f us %9.1f us %9.1f us" % ("%d, %d > %d %s" % (M, K, N, act), row[0][0], row[1][0], row[0][1], row[1][1]))
    ops.LINEAR_SMALL = 1


if __name__ == "__main__":
    main()
