"""Forward time of the split-bf16 (x3) fp32 convolution against the native fp32 path of ops.conv2d, per layer shape.
usage: python benchmarks/kernel_bench_x3.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402

LAYERS = [("res3x3 256>256 @32", 256, 256, 32, 3), ("up5x5 256>128 @64", 256, 128, 64, 5), ("up5x5 128>64 @128", 128, 64, 128, 5)]


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
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for B in [int(v) for v in sys.argv[1:]] or [16, 32, 48]:
        for name, ci, co, H, k in LAYERS:
            x = torch.randn(B, H, H, ci, device=dev)
            w = torch.randn(co, ci, k, k, device=dev) * 0.05
            b = torch.zeros(co, device=dev)
            wp = torch.empty(lib.dwc_x3_weight_prepared_elems(co, ci, k), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.dwc_x3_weight_prepare(w.data_ptr(), wp.data_ptr(), co, ci, k, co, 0, st), "prep")
            y = torch.empty(B, H, H, co, device=dev)
            flops = 2.0 * B * H * H * co * ci * k * k
            t3 = timeit(lambda: _lib.check(lib.dwc_x3_conv2d_same(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, ci,
                                                                  co, co, k, 1, 1, st), "x3"))
            xc = x.permute(0, 3, 1, 2)          # channels_last view for ops
            old = ops.X3
            ops.X3 = 0
            with torch.no_grad():
                tn = timeit(lambda: ops.conv2d(xc, w, b, 1, k // 2, "relu"))
            ops.X3 = old
            # weight gradient: x3 against the native path (ops with X3 off)
            dy = torch.randn(B, H, H, co, device=dev)
            nws = lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, H, ci, co, k)
            tw3 = twn = float("nan")
            if nws:
                wsb = torch.empty(nws, dtype=torch.uint8, device=dev)
                dw = torch.empty(co, ci, k, k, device=dev)
                tw3 = timeit(lambda: _lib.check(lib.dwc_x3_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, H, ci, co, k, ci,
                                                                        co, wsb.data_ptr(), nws, st), "x3 wgrad"))
            old = ops.X3
            ops.X3 = 0
            wg = w.clone().requires_grad_(True)
            y2 = ops.conv2d(xc, wg, None, 1, k // 2, "none")
            dyc = dy.permute(0, 3, 1, 2)
            twn = timeit(lambda: torch.autograd.grad(y2, wg, dyc, retain_graph=True))
            ops.X3 = old
            print("      wgrad: x3 %8.1f us %6.1f TF | native %8.1f us %6.1f TF | x%.2f" % (tw3 * 1e6, flops / tw3 / 1e12, twn * 1e6,
                                                                                          flops / twn / 1e12, twn / tw3))
            print("B=%-3d %-20s %7.2f GFLOP | x3 %8.1f us %6.1f TF (bf16 MFMA %4.1f%%) | native fp32 path %8.1f us %6.1f TF | x%.2f" % (
                B, name, flops / 1e9, t3 * 1e6, flops / t3 / 1e12, 100 * 6 * flops / t3 / 2.5e15, tn * 1e6, flops / tn / 1e12, tn / t3))


if __name__ == "__main__":
    main()
