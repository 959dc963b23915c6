#!/usr/bin/env python3
"""Per-kernel timings of the bf16 conv kernels on one MI355X (development aid; bench.py --config c2 is the contract).

    python benchmarks/kernel_bench_bf16.py [B] [name-filter]

Times forward / data-gradient / weight-gradient of the layer shapes of the 128x128 configuration through the autograd
ops (so the planner, the ring strips and the reduces are included) with HIP events on the launch stream, random bf16
operands (zero-filled operands clock higher: cdna_hip_programming.md rule 25), and prints achieved TFLOP/s against the
dense bf16 MFMA peak (2500 TF, MI355X_MICROARCH.md).
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import ops  # noqa: E402

PEAK_TF = 2500.0

# name, Cin, Cout, H, k, stride, pad
LAYERS = [
    ("res3x3 256>256 @32", 256, 256, 32, 3, 1, 1),
    ("up5x5 256>128 @64", 256, 128, 64, 5, 1, 2),
    ("up5x5 128>64 @128", 128, 64, 128, 5, 1, 2),
    ("down4x4 64>128 @128", 64, 128, 128, 4, 2, 1),
    ("down4x4 128>256 @64", 128, 256, 64, 4, 2, 1),
    ("D 4x4 64>128 @64", 64, 128, 64, 4, 2, 1),
    ("D 4x4 128>256 @32", 128, 256, 32, 4, 2, 1),
    ("D 4x4 256>512 @16", 256, 512, 16, 4, 2, 1),
    ("D 4x4 512>512 @8", 512, 512, 8, 4, 2, 1),
    ("style 4x4 256>256 @8", 256, 256, 8, 4, 2, 1),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda:0")
    ops.set_precision("bf16")
    print("B=%d  bf16 MFMA peak %.0f TF" % (B, PEAK_TF))
    for name, ci, co, H, k, s, p in LAYERS:
        if only not in name:
            continue
        x = torch.randn(B, ci, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, k, k, device=dev) * 0.05)
        b = torch.zeros(co, device=dev)
        flops = 2.0 * B * ((H + 2 * p - k) // s + 1) ** 2 * co * ci * k * k

        def fwd():
            with torch.no_grad():
                return ops.conv2d(x, w, b, s, p, "relu")
        tf = timeit(fwd)
        # backward pieces separately: only dx, only dw
        xg = x.clone().requires_grad_(True)
        y = ops.conv2d(xg, w, b, s, p, "none")
        dy = torch.randn_like(y)
        td = timeit(lambda: torch.autograd.grad(y, xg, dy, retain_graph=True))
        wg = w.clone().requires_grad_(True)
        y2 = ops.conv2d(x, wg, None, s, p, "none")
        tw = timeit(lambda: torch.autograd.grad(y2, wg, dy, retain_graph=True))
        print("%-22s %8.2f GFLOP | fwd %8.1f us %6.0f TF (%4.1f%%) | dgrad %8.1f us %6.0f TF (%4.1f%%) | wgrad %8.1f us %6.0f TF (%4.1f%%)" % (
            name, flops / 1e9, tf * 1e6, flops / tf / 1e12, 100 * flops / tf / 1e12 / PEAK_TF,
            td * 1e6, flops / td / 1e12, 100 * flops / td / 1e12 / PEAK_TF,
            tw * 1e6, flops / tw / 1e12, 100 * flops / tw / 1e12 / PEAK_TF))


if __name__ == "__main__":
    main()
