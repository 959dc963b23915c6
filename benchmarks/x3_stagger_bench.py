"""Duo-form split-product kernels under a phase offset between the two workgroups of a CU (DWC_X3_STAGGER, 10 ns ticks; the
library reads the knob once per process: run this script once per value).  Random operands, median of 12 after 3 warm-up.
usage: DWC_X3_STAGGER=2000 python benchmarks/x3_stagger_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib  # noqa: E402

LAYERS = [("3x3 256>256 @32", 256, 256, 32, 3), ("5x5 256>128 @64", 256, 128, 64, 5), ("5x5 128>64 @128", 128, 64, 128, 5)]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    print("DWC_X3_STAGGER =", os.environ.get("DWC_X3_STAGGER", "0"))
    for B in (16, 32, 48):
        for name, ci, co, H, k in LAYERS:
            x = torch.randn(B, H, H, ci, device=dev)
            w = torch.randn(co, ci, k, k, device=dev) * 0.05
            b = torch.zeros(co, device=dev)
            wp = torch.empty(lib.dwc_x3_weight_prepared_elems(co, ci, k), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.dwc_x3_weight_prepare(w.data_ptr(), wp.data_ptr(), co, ci, k, co, 0, st), "prep")
            y = torch.empty(B, H, H, co, device=dev)
            flops = 2.0 * B * H * H * co * ci * k * k
            ts = []
            for it in range(15):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _lib.check(lib.dwc_x3_conv2d_same(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, ci, co, co, k, 1, 1, st), "x3")
                e.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts.append(a.elapsed_time(e) * 1e-3)
            ts.sort()
            t = ts[len(ts) // 2]
            wgs = B * (H // 16) ** 2 * (co // 64)
            print("  B%-3d %-18s %5d WGs  med %8.1f us  %6.1f TF fp32-equiv  %.3f of 2.5PF executed  checksum %.6e" % (
                B, name, wgs, t * 1e6, flops / t / 1e12, 6 * flops / t / 2.5e15, float(y.double().sum())))


if __name__ == "__main__":
    main()
