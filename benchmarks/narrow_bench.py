#!/usr/bin/env python3
"""The 7x7 layers with a 3/4-plane side (image heads 64 -> image, stems image -> 64) in isolation: forward, data gradient, weight
gradient of both, through the autograd ops, HIP events over 20 calls, against each launch's HBM floor (operands read once, results written
once at 5 TB/s).  DWC_HIP_LIB=<ablation build> DWC_NARROW_DBG=<bits> times the ablations of conv_narrow_kernel (make ABLATIONS=1).

    python benchmarks/narrow_bench.py [bf16|fp32] [B ...]
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
    return a.elapsed_time(b) / iters * 1e3


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    batches = [int(b) for b in sys.argv[2:]] or [128, 384]
    ops.set_precision(prec)
    dev = torch.device("cuda", 0)
    dt = ops.BF16 if prec == "bf16" else torch.float32
    P = ops.image_planes(dt)
    el = 2 if prec == "bf16" else 4
    H = 128
    for B in batches:
        feat = ops.empty_cl(B, 64, H, H, dev, dt)
        feat.copy_(torch.randn(B, 64, H, H, device=dev))
        feat.requires_grad_(True)
        w4 = torch.zeros(P, 64, 7, 7, device=dev)
        w4[:4] = torch.randn(4, 64, 7, 7, device=dev) * 0.02
        w4.requires_grad_(True)
        b4 = torch.zeros(P, device=dev, requires_grad=True)
        img = ops.empty_cl(B, P, H, H, dev, dt)
        img.copy_(torch.randn(B, P, H, H, device=dev))
        img.requires_grad_(True)
        ws = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).requires_grad_(True)
        bs = torch.zeros(64, device=dev, requires_grad=True)
        big, small = B * H * H * 64 * el, B * H * H * P * el

        def heads_fwd():
            with torch.no_grad():
                return ops.conv2d_heads(feat, w4, b4)

        y = ops.conv2d_heads(feat, w4, b4)
        gy = torch.randn_like(y)

        def heads_fb():
            ops.conv2d_heads(feat, w4, b4).backward(gy)
            feat.grad = w4.grad = b4.grad = None

        def stem_fwd():
            with torch.no_grad():
                return ops.conv2d(img, ws, bs, 1, 3, "relu")

        z = ops.conv2d(img, ws, bs, 1, 3, "relu")
        gz = torch.randn_like(z)

        def stem_fb():
            ops.conv2d(img, ws, bs, 1, 3, "relu").backward(gz)
            img.grad = ws.grad = bs.grad = None

        floor = (big + small) / 5e12 * 1e6
        t_hf, t_hfb, t_sf, t_sfb = timeit(heads_fwd), timeit(heads_fb), timeit(stem_fwd), timeit(stem_fb)
        print("%s B=%d  (one pass over the 64-channel tensor + the image at 5 TB/s: %.0f us)" % (prec, B, floor))
        print("  heads 64>%d forward            %8.1f us" % (P, t_hf))
        print("  heads forward + backward       %8.1f us   (backward alone %.1f: image-plane dY -> 64 dgrad + small-k wgrad)" % (t_hfb, t_hfb - t_hf))
        print("  stem %d>64 forward              %8.1f us" % (P, t_sf))
        print("  stem forward + backward        %8.1f us   (backward alone %.1f: act-bwd + 64 -> image dgrad + small-k wgrad)" % (t_sfb, t_sfb - t_sf))


if __name__ == "__main__":
    main()
