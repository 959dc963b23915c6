"""Timing-only ablations of the two-plane halo kernel (WRONG results; library built with -DDWC_DEV_ABLATIONS, loaded through
DWC_HIP_LIB): DWC_X3_DBG = 1 no MFMA, 2 no fragment reads, 4 no weight staging, 8 no barrier, 16 no patch refresh / flush,
32 no conversion at the slab boundary, 64 no flush, 128 no patch DMA, 3 = 1 + 2,
31 = empty skeleton.  usage: DWC_HIP_LIB=.../libdwcgan_hip_abl.so DWC_X3_DBG=<n> python benchmarks/h2_ablation_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402


REPS = 8


def med(fn, n=15, skip=3):
    ts = []
    for it in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):                    # back-to-back launches: the event pair's own ~10 us are shared
            fn()
        e.record()
        torch.cuda.synchronize()
        if it >= skip:
            ts.append(a.elapsed_time(e) * 1e3 / REPS)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for name, B, ci, co, H, k in (("3x3 256>256 @32 B48", 48, 256, 256, 32, 3), ("5x5 256>128 @64 B16", 16, 256, 128, 64, 5),
                                  ("5x5 128>64 @128 B16", 16, 128, 64, 128, 5)):
        x = torch.randn(B, H, H, ci, device=dev)
        w = torch.randn(co, ci, k, k, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        y = torch.empty(B, H, H, co, device=dev)
        hp = torch.zeros(lib.dwc_h2_weight_prepared_elems(co, ci, k), dtype=torch.float16, device=dev)
        wsl, wep = ops.amax_slot(dev)
        _lib.check(lib.dwc_absmax(w.data_ptr(), w.numel(), wsl, wep, st), "absmax w")
        _lib.check(lib.dwc_h2_weight_prepare(w.data_ptr(), hp.data_ptr(), co, ci, k, co, 0, wsl, wep, st), "h2 prep")
        xsl, xep = ops.amax_slot(dev)
        _lib.check(lib.dwc_absmax(x.data_ptr(), x.numel(), xsl, xep, st), "absmax x")
        t = med(lambda: _lib.check(lib.dwc_h2_conv2d_same_add_ws(x.data_ptr(), xsl, xep, hp.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, 0,
                                                                 B, H, H, ci, co, co, k, 1, 1, None, 0, None, st), "h2"))
        out.append("%s %.1f us" % (name, t))
    print("DWC_X3_DBG=%s: %s" % (os.environ.get("DWC_X3_DBG", "0"), " | ".join(out)))


if __name__ == "__main__":
    main()
