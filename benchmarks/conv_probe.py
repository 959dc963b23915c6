#!/usr/bin/env python3
"""Sustained-clock timing of single conv launches (development aid): each shape is launched a few hundred times
back to back so that the GPU clock has ramped (short bursts run at ~2.0 GHz and understate a kernel by 15-20 %).

    python benchmarks/conv_probe.py            # the comparison set used in DESIGN.md
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import ops, _lib  # noqa: E402

PEAK_TF = 157.3
# name, B, Cin, Cout, H, k, stride, pad
SHAPES = [
    ("1x1 2304>256 @32 B32 (pure GEMM 32768x256x2304)", 32, 2304, 256, 32, 1, 1, 0),
    ("3x3 256>256 @32 B32", 32, 256, 256, 32, 3, 1, 1),
    ("3x3 256>256 @32 B48", 48, 256, 256, 32, 3, 1, 1),
    ("3x3 256>256 @32 B16", 16, 256, 256, 32, 3, 1, 1),
    ("5x5 256>128 @64 B16", 16, 256, 128, 64, 5, 1, 2),
    ("5x5 128>64 @128 B16", 16, 128, 64, 128, 5, 1, 2),
]


def timeit(fn, iters):
    for _ in range(iters // 2):
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
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    iters = int(os.environ.get("ITERS", 300))
    for name, B, ci, co, H, k, s, p in SHAPES:
        x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(B, co, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn_like(y)
        dw = torch.empty_like(w)
        w_f = ops._prepped(w, "fwd", co, ci, s)
        w_d = ops._prepped(w, "dgrad", co, ci, s)
        ws = ops.workspace(max(lib.dwc_conv2d_bwd_weight_ws_bytes(B, H, H, ci, co, k, k, s, p),
                               lib.dwc_conv2d_fwd_ws_bytes(B, H, H, ci, co, k, k, s, p),
                               lib.dwc_conv2d_bwd_data_ws_bytes(B, H, H, ci, co, k, k, s, p), 256), dev)
        dxp = torch.empty(B * (H + 2 * p) * (H + 2 * p) * ci, device=dev)
        flops = 2.0 * B * Ho * Ho * co * ci * k * k
        tf = timeit(lambda: lib.dwc_conv2d_fwd(x.data_ptr(), w_f.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, ci, co, k, k,
                                               s, p, 1, ws.data_ptr(), ws.numel(), st), iters)
        td = timeit(lambda: lib.dwc_conv2d_bwd_data(dy.data_ptr(), w_d.data_ptr(), dxp.data_ptr(), B, H, H, ci, co, k, k, s, p,
                                                    ws.data_ptr(), ws.numel(), st), iters)
        tw = timeit(lambda: lib.dwc_conv2d_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, H, ci, co, k, k, s, p,
                                                      ci, co, ws.data_ptr(), ws.numel(), st), iters)
        print("%-50s fwd %6.1f TF %5.1f%% | dgrad %6.1f TF %5.1f%% | wgrad %6.1f TF %5.1f%%" % (
            name, flops / tf / 1e12, 100 * flops / tf / 1e12 / PEAK_TF, flops / td / 1e12, 100 * flops / td / 1e12 / PEAK_TF,
            flops / tw / 1e12, 100 * flops / tw / 1e12 / PEAK_TF), flush=True)


if __name__ == "__main__":
    main()
