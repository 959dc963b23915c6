"""Border-ring step of the "same" data gradients (strip GEMMs + fold) through the C ABI, per layer shape and batch, both precisions.
usage: python benchmarks/ring_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402

LAYERS = [("3x3 256>256 @32", 256, 256, 32, 3), ("5x5 256>128 @64", 256, 128, 64, 5), ("5x5 128>64 @128", 128, 64, 128, 5)]
S2 = [("4x4s2 64>128 @128", 64, 128, 128), ("4x4s2 128>256 @64", 128, 256, 64), ("4x4s2 256>256 @32", 256, 256, 32)]


def med(fn, n=15, skip=3):
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
    for half, batches in ((False, (16, 48)), (True, (128, 384))):
        dt = torch.bfloat16 if half else torch.float32
        pre = "bf16_" if half else ""
        for B in batches:
            for name, ci, co, H, k in LAYERS:
                w = (torch.randn(co, ci, k, k, device=dev) * 0.05)
                g = torch.randn(B, H, H, co, device=dev).to(dt)
                dx = torch.zeros(B, H, H, ci, device=dev).to(dt)
                w_dg = ops._prepped(w, "dgrad", co, ci, 1, None, half)
                w_dg_t = ops._prepped(w, "dgrad_t", co, ci, 1, None, half)
                nws = getattr(lib, "dwc_%sconv2d_bwd_data_same_ws_bytes" % pre)(B, H, H, ci, co, k, k, k // 2)
                ws = torch.empty(nws, dtype=torch.uint8, device=dev)
                fn = getattr(lib, "dwc_%sconv2d_bwd_data_ring" % pre)
                t = med(lambda: _lib.check(fn(g.data_ptr(), w_dg.data_ptr(), w_dg_t.data_ptr(), dx.data_ptr(), B, H, H, ci, co, k, k, k // 2,
                                              ws.data_ptr(), nws, st), "ring"))
                print("  %s B%-3d %-18s ring %7.1f us" % ("bf16" if half else "fp32", B, name, t * 1e6))
            for name, ci, co, H in S2:
                w = (torch.randn(co, ci, 4, 4, device=dev) * 0.05)
                g = torch.randn(B, H // 2, H // 2, co, device=dev).to(dt)
                dx = torch.zeros(B, H, H, ci, device=dev).to(dt)
                dxp = torch.zeros(B, H + 2, H + 2, ci, device=dev).to(dt)
                w_dg = ops._prepped(w, "dgrad", co, ci, 2, None, half)
                fn = getattr(lib, "dwc_%sconv2d_bwd_data_s2_ring" % pre)
                t = med(lambda: _lib.check(fn(g.data_ptr(), w_dg.data_ptr(), dxp.data_ptr(), dx.data_ptr(), B, H, H, ci, co, st), "s2 ring"))
                print("  %s B%-3d %-18s ring %7.1f us" % ("bf16" if half else "fp32", B, name, t * 1e6))


if __name__ == "__main__":
    main()
