"""Split-product kernels through the C ABI per layer shape of c1: forward (conv_halo_x3_kernel) and weight gradient
(wgrad_x3_kernel + reduce), stride-1 3x3 / 5x5 and the stride-2 4x4 forms.  Random operands, median of 12 after 3 warm-up.
usage: python benchmarks/x3_kernels_bench.py [B ...]      (also the workload of benchmarks/x3_sq_counters.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib  # noqa: E402

LAYERS = [("3x3 256>256 @32", 256, 256, 32, 3, 1), ("5x5 256>128 @64", 256, 128, 64, 5, 1), ("5x5 128>64 @128", 128, 64, 128, 5, 1),
          ("4x4s2 64>128 @128", 64, 128, 128, 4, 2), ("4x4s2 128>256 @64", 128, 256, 64, 4, 2)]


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
    for B in [int(v) for v in sys.argv[1:]] or [16, 32, 48]:
        for name, ci, co, H, k, s in LAYERS:
            Ho = H // s
            x = torch.randn(B, H, H, ci, device=dev)
            dy = torch.randn(B, Ho, Ho, co, device=dev)
            w = torch.randn(co, ci, k, k, device=dev) * 0.05
            b = torch.zeros(co, device=dev)
            wp = torch.empty(lib.dwc_x3_weight_prepared_elems(co, ci, k), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.dwc_x3_weight_prepare(w.data_ptr(), wp.data_ptr(), co, ci, k, co, 0, st), "prep")
            y = torch.empty(B, Ho, Ho, co, device=dev)
            flops = 2.0 * B * Ho * Ho * co * ci * k * k
            # scratch of the contraction split of small launches (<= 256 tiles; DWC_X3_KSPLIT=0: off)
            need = lib.dwc_x3_conv2d_ksplit_ws_bytes(B, H, H, ci, co, k, s)
            ksw = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            kst = torch.zeros(lib.dwc_x3_conv2d_ksplit_ticket_words(), dtype=torch.int32, device=dev)
            if s == 1:
                tf = med(lambda: _lib.check(lib.dwc_x3_conv2d_same_add_ws(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, H, H,
                                                                          ci, co, co, k, 1, 1, ksw.data_ptr(), need, kst.data_ptr(), st), "x3"))
            else:
                tf = med(lambda: _lib.check(lib.dwc_x3_conv2d_s2_ws(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, ci, co, co,
                                                                    1, ksw.data_ptr(), need, kst.data_ptr(), st), "x3s2"))
            nws = lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, H, ci, co, k)
            tw = float("nan")
            if nws:
                wsb = torch.empty(nws, dtype=torch.uint8, device=dev)
                dw = torch.empty(co, ci, k, k, device=dev)
                try:
                    tw = med(lambda: _lib.check(lib.dwc_x3_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, H, ci, co, k, ci, co,
                                                                        wsb.data_ptr(), nws, st), "x3 wgrad"))
                except _lib.HipKernelError:          # (a knob selected a variant this shape has no instantiation of)
                    pass
            print("  B%-3d %-18s%s fwd %8.1f us  %.3f of 2.5PF | wgrad+reduce %8.1f us  %.3f of 2.5PF   checksum %.6e" % (
                B, name, " (split)" if need else "        ", tf * 1e6, 6 * flops / tf / 2.5e15, tw * 1e6, 6 * flops / tw / 2.5e15, float(y.double().sum())))
            # the same layer as two-plane f16 split products (three MFMAs per fp32 MFMA-equivalent), absmax passes timed apart
            from hipdwc import ops
            hp = torch.empty(lib.dwc_h2_weight_prepared_elems(co, ci, k), dtype=torch.float16, device=dev)
            wsl, wep = ops.amax_slot(dev)
            _lib.check(lib.dwc_absmax(w.data_ptr(), w.numel(), wsl, wep, st), "absmax w")
            _lib.check(lib.dwc_h2_weight_prepare(w.data_ptr(), hp.data_ptr(), co, ci, k, co, 0, wsl, wep, st), "h2 prep")
            xsl, xep = ops.amax_slot(dev)
            dsl, dep = ops.amax_slot(dev)
            ta = med(lambda: _lib.check(lib.dwc_absmax(x.data_ptr(), x.numel(), xsl, xep, st), "absmax x"))
            _lib.check(lib.dwc_absmax(dy.data_ptr(), dy.numel(), dsl, dep, st), "absmax dy")
            if s == 1:
                tf2 = med(lambda: _lib.check(lib.dwc_h2_conv2d_same_add_ws(x.data_ptr(), xsl, xep, hp.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, 0,
                                                                           B, H, H, ci, co, co, k, 1, 1, ksw.data_ptr(), need, kst.data_ptr(), st), "h2"))
            else:
                tf2 = med(lambda: _lib.check(lib.dwc_h2_conv2d_s2_ws(x.data_ptr(), xsl, xep, hp.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, B, H, H, ci,
                                                                     co, co, 1, ksw.data_ptr(), need, kst.data_ptr(), st), "h2s2"))
            tw2 = float("nan")
            if nws:
                tw2 = med(lambda: _lib.check(lib.dwc_h2_conv2d_wgrad(x.data_ptr(), xsl, xep, dy.data_ptr(), dsl, dep, dw.data_ptr(), B, H, H, ci, co,
                                                                     k, ci, co, wsb.data_ptr(), nws, st), "h2 wgrad"))
            print("       %-18s  h2 fwd %8.1f us  %.3f of 2.5PF (%.2fx) | wgrad+reduce %8.1f us  %.3f of 2.5PF (%.2fx) | absmax(x) %6.1f us  checksum %.6e" % (
                "", tf2 * 1e6, 3 * flops / tf2 / 2.5e15, tf / tf2, tw2 * 1e6, 3 * flops / tw2 / 2.5e15, tw / tw2, ta * 1e6, float(y.double().sum())))


if __name__ == "__main__":
    main()
