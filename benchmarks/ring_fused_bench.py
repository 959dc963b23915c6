"""Data gradient of the stride-1 reflect-padded layers, bf16: ONE fused launch (border ring inside the halo kernel, r06) against the
two-call form (halo interior, then strip GEMM + fold), through the C ABI at the c2 shapes.  usage: python benchmarks/ring_fused_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402

LAYERS = [("3x3 256>256 @32", 256, 256, 32, 3), ("5x5 256>128 @64", 256, 128, 64, 5), ("5x5 128>64 @128", 128, 64, 128, 5)]


def med(fn, n=24, skip=4):
    ts = []
    for it in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if it >= skip:
            ts.append(a.elapsed_time(e) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ops.set_precision("bf16")
    for B in (128, 384):
        for name, ci, co, H, k in LAYERS:
            w = (torch.randn(co, ci, k, k, device=dev) * 0.05)
            g = torch.randn(B, H, H, co, device=dev).to(torch.bfloat16)
            dx = torch.zeros(B, H, H, ci, device=dev).to(torch.bfloat16)
            w_dg = ops._prepped(w, "dgrad", co, ci, 1, None, True)
            w_dg_t = ops._prepped(w, "dgrad_t", co, ci, 1, None, True)
            nws = lib.dwc_bf16_conv2d_bwd_data_same_ws_bytes(B, H, H, ci, co, k, k, k // 2)
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)

            def interior():
                _lib.check(lib.dwc_bf16_conv2d_same_halo_add(g.data_ptr(), w_dg.data_ptr(), None, None, dx.data_ptr(), B, H, H, co, ci, k, 0, 0, st), "halo")

            def two_call():
                interior()
                _lib.check(lib.dwc_bf16_conv2d_bwd_data_ring(g.data_ptr(), w_dg.data_ptr(), w_dg_t.data_ptr(), dx.data_ptr(), B, H, H, ci, co, k, k,
                                                              k // 2, ws.data_ptr(), nws, st), "ring")

            def fused():
                _lib.check(lib.dwc_bf16_conv2d_bwd_data_same_fused(g.data_ptr(), w_dg.data_ptr(), None, dx.data_ptr(), B, H, H, ci, co, k, st), "fused")
            assert lib.dwc_bf16_conv2d_bwd_data_same_fused_ok(B, H, H, ci, co, k)
            ti, t2, tf = med(interior), med(two_call), med(fused)
            t2b, tfb = med(two_call), med(fused)
            gf = 2.0 * B * H * H * ci * co * k * k
            print("bf16 B%-3d %-16s interior %7.1f us | two-call %7.1f / %7.1f us | fused %7.1f / %7.1f us (%.3f of 2.5 PF)" % (
                B, name, ti * 1e6, t2 * 1e6, t2b * 1e6, tf * 1e6, tfb * 1e6, gf / min(tf, tfb) / 2.5e15))


if __name__ == "__main__":
    main()
