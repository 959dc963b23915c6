#!/usr/bin/env python3
"""Per-kernel timings of the conv stack on one MI355X (development aid; bench.py is the contract).

    python benchmarks/kernel_bench.py [B]

Times forward / data-gradient / weight-gradient of the layer shapes of the 128x128
configuration with HIP events on the launch stream and prints achieved TFLOP/s against
the fp32 MFMA peak (157.3 TF, MI355X_MICROARCH.md).
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import ops, _lib  # noqa: E402

PEAK_TF = 157.3

# name, Cin, Cout, H, k, stride, pad
LAYERS = [
    ("res3x3 256>256 @32", 256, 256, 32, 3, 1, 1),
    ("up5x5 256>128 @64", 256, 128, 64, 5, 1, 2),
    ("up5x5 128>64 @128", 128, 64, 128, 5, 1, 2),
    ("heads7x7 64>4 @128", 64, 4, 128, 7, 1, 3),
    ("stem7x7 4>64 @128", 4, 64, 128, 7, 1, 3),
    ("down4x4 64>128 @128", 64, 128, 128, 4, 2, 1),
    ("down4x4 128>256 @64", 128, 256, 64, 4, 2, 1),
    ("D 4x4 256>512 @16", 256, 512, 16, 4, 2, 1),
    ("D 4x4 512>512 @8", 512, 512, 8, 4, 2, 1),
    ("D 4x4 128>256 @32", 128, 256, 32, 4, 2, 1),
    ("style 4x4 256>256 @8", 256, 256, 8, 4, 2, 1),
]


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    print("B=%d  fp32 MFMA peak %.1f TF" % (B, PEAK_TF))
    for name, ci, co, H, k, s, p in LAYERS:
        if only not in name:
            continue
        x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(B, co, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        w_hwio = ops._prepped(w, "fwd", co, ci, s)
        w_dg = ops._prepped(w, "dgrad", co, ci, s)
        ws = ops.workspace(max(lib.dwc_conv2d_bwd_weight_ws_bytes(B, H, H, ci, co, k, k, s, p),
                               lib.dwc_conv2d_fwd_ws_bytes(B, H, H, ci, co, k, k, s, p),
                               lib.dwc_conv2d_bwd_data_ws_bytes(B, H, H, ci, co, k, k, s, p)), dev)
        dxp = torch.empty(B * (H + 2 * p) * (H + 2 * p) * ci, device=dev)
        flops = 2.0 * B * Ho * Ho * co * ci * k * k
        tf = timeit(lambda: lib.dwc_conv2d_fwd(x.data_ptr(), w_hwio.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, ci, co,
                                               k, k, s, p, 1, ws.data_ptr(), ws.numel(), st))
        td = timeit(lambda: lib.dwc_conv2d_bwd_data(dy.data_ptr(), w_dg.data_ptr(), dxp.data_ptr(), B, H, H, ci, co, k, k, s,
                                                    p, ws.data_ptr(), ws.numel(), st))
        tw = timeit(lambda: lib.dwc_conv2d_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, H, ci, co, k, k, s, p,
                                                      ci, co, ws.data_ptr(), ws.numel(), st))
        print("%-22s %7.2f GFLOP | fwd %8.1f us %5.1f TF (%4.1f%%) | dgrad %8.1f us %5.1f TF | wgrad %8.1f us %5.1f TF" % (
            name, flops / 1e9, tf * 1e6, flops / tf / 1e12, 100 * flops / tf / 1e12 / PEAK_TF,
            td * 1e6, flops / td / 1e12, tw * 1e6, flops / tw / 1e12))


if __name__ == "__main__":
    main()
