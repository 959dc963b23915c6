"""5x5 halo kernel, bf16: the shipped tiles against the 4 x 1-wave two-per-CU tile (DWC_H16_WM4=1, read at the first launch: run once per
setting).  Forward (reflect) and data-gradient interior (zero rule) of both upsampling layers.  usage: DWC_H16_WM4=0|1 python benchmarks/halo5_tile_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402

# (name, gathered channels, output channels, H)
LAYERS = [("fwd 256>128 @64", 256, 128, 64), ("fwd 128>64 @128", 128, 64, 128), ("dgrad 128>256 @64", 128, 256, 64), ("dgrad 64>128 @128", 64, 128, 128)]


def med(fn, n=20, skip=4):
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
    print("DWC_H16_WM4 =", os.environ.get("DWC_H16_WM4", "0"))
    for B in (128, 384):
        for name, ci, co, H in LAYERS:
            fwd = name.startswith("fwd")
            w = torch.randn(co, ci, 5, 5, device=dev) * 0.05 if fwd else torch.randn(ci, co, 5, 5, device=dev) * 0.05
            x = torch.randn(B, H, H, ci, device=dev).to(torch.bfloat16)
            y = torch.zeros(B, H, H, co, device=dev).to(torch.bfloat16)
            wp = ops._prepped(w, "fwd" if fwd else "dgrad", co if fwd else ci, ci if fwd else co, 1, None, True)
            t = med(lambda: _lib.check(lib.dwc_bf16_conv2d_same_halo(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), B, H, H, ci, co, 5, 0,
                                                                     1 if fwd else 0, st), "halo"))
            gf = 2.0 * B * H * H * ci * co * 25
            print("B%-3d %-18s %8.1f us  %.3f of 2.5 PF  checksum %.6e" % (B, name, t * 1e6, gf / t / 2.5e15, float(y.float().abs().mean())))


if __name__ == "__main__":
    main()
