#!/usr/bin/env python3
"""Achieved HBM bandwidth of the norm / pointwise kernels at the shapes of the bench workloads (development aid).

    python benchmarks/hbm_kernels_bench.py [bf16|fp32] [B]

Times each op through the autograd wrappers (HIP events on the launch stream) and prints microseconds and the ALGORITHMIC
bytes (every tensor the op must read or write, once) over that time, against the 8 TB/s spec / ~6.3 TB/s achievable of
MI355X_MICROARCH.md.
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
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
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ops.set_precision(prec)
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    es = 2 if prec == "bf16" else 4
    dev = torch.device("cuda:0")

    def feat(b, c, h):
        return torch.randn(b, c, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)

    def row(name, t, nbytes):
        print("%-44s %9.1f us  %7.2f GB  %6.2f TB/s" % (name, t * 1e6, nbytes / 1e9, nbytes / t / 1e12))

    for bb in (B, 3 * B):
        # instance norm (AdaIN + ReLU, AdaIN + residual) on the ResBlock maps
        x = feat(bb, 256, 32).requires_grad_(True)
        n = x.numel()
        g, be = torch.rand(bb * 256, device=dev) + 0.5, torch.randn(bb * 256, device=dev)
        res = feat(bb, 256, 32)
        row("IN fwd adain+relu %dx256x32x32" % bb, timeit(lambda: ops.instance_norm(x.detach(), g, be, relu=True)), 3 * n * es)
        row("IN fwd adain+res  %dx256x32x32" % bb, timeit(lambda: ops.instance_norm(x.detach(), g, be, residual=res)), 4 * n * es)
        y = ops.instance_norm(x, g, be, relu=True)
        dy = torch.randn_like(y)
        row("IN bwd adain+relu %dx256x32x32" % bb, timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True)), 5 * n * es)
        # layer norm on the upsampled maps
        x2 = feat(bb, 128, 64).requires_grad_(True)
        n2 = x2.numel()
        ga, bt = torch.rand(128, device=dev), torch.randn(128, device=dev)
        row("LN fwd relu %dx128x64x64" % bb, timeit(lambda: ops.layer_norm_munit(x2.detach(), ga, bt, relu=True)), 3 * n2 * es)
        y2 = ops.layer_norm_munit(x2, ga, bt, relu=True)
        dy2 = torch.randn_like(y2)
        row("LN bwd relu %dx128x64x64" % bb, timeit(lambda: torch.autograd.grad(y2, x2, dy2, retain_graph=True)), 5 * n2 * es)
        # bilinear x2
        row("upsample2x fwd %dx256x32x32" % bb, timeit(lambda: ops.upsample2x(x.detach())), 5 * n * es)
        u = ops.upsample2x(x)
        du = torch.randn_like(u)
        row("upsample2x bwd %dx256x32x32" % bb, timeit(lambda: torch.autograd.grad(u, x, du, retain_graph=True)), 5 * n * es)
    print("(spec 8 TB/s, float4-copy 6.29 TB/s)")


if __name__ == "__main__":
    main()
