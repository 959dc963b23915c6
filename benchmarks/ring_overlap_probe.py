"""Does the border-ring step (strip GEMMs + fold: a latency-bound launch) hide behind the interior launch of the same data gradient
when the two are issued on different streams?  fp32 two-plane kernels, the c1 shapes.  usage: python benchmarks/ring_overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib, ops  # noqa: E402


def med(fn, n=15, skip=3):
    ts = []
    for it in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if it >= skip:
            ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    for B in (16, 48):
        for name, ci, co, H, k in (("3x3 256>256 @32", 256, 256, 32, 3), ("5x5 256>128 @64", 256, 128, 64, 5)):
            w = torch.randn(co, ci, k, k, device=dev) * 0.05
            g = torch.randn(B, H, H, co, device=dev)
            dx = torch.zeros(B, H, H, ci, device=dev)
            w_dg = ops._prepped(w, "dgrad", co, ci, 1)
            w_dg_t = ops._prepped(w, "dgrad_t", co, ci, 1)
            w_h2 = ops._prepped(w, "h2_dgrad", co, ci, 1)
            ga = ops.amax_of(g)
            nws = lib.dwc_conv2d_bwd_data_same_ws_bytes(B, H, H, ci, co, k, k, k // 2)
            ws_ring = torch.empty(nws, dtype=torch.uint8, device=dev)
            ks_ws, ks_n, ks_t = ops._x3_ksplit(lib, dev, B, H, H, co, ci, k, 1)

            def interior(st):
                _lib.check(lib.dwc_h2_conv2d_same_add_ws(g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), None, None, dx.data_ptr(), None, 0, B, H, H, co, ci,
                                                         ci, k, 0, 0, ks_ws.data_ptr() if ks_ws is not None else None, ks_n, ks_t, st), "interior")

            def ring(st):
                _lib.check(lib.dwc_conv2d_bwd_data_ring(g.data_ptr(), w_dg.data_ptr(), w_dg_t.data_ptr(), dx.data_ptr(), B, H, H, ci, co, k, k, k // 2,
                                                        ws_ring.data_ptr(), nws, st), "ring")

            def serial():
                interior(main_s.cuda_stream)
                ring(main_s.cuda_stream)

            def overlapped():
                ev = torch.cuda.Event()
                ev.record(main_s)
                side.wait_event(ev)
                ring(side.cuda_stream)               # (timing only: the fold inside would have to wait for the interior)
                interior(main_s.cuda_stream)
                ev2 = torch.cuda.Event()
                ev2.record(side)
                main_s.wait_event(ev2)

            ti = med(lambda: interior(main_s.cuda_stream))
            tr = med(lambda: ring(main_s.cuda_stream))
            ts, to = med(serial), med(overlapped)
            print("B%-2d %-16s interior %6.1f us  ring %5.1f us  serial %6.1f us  two streams %6.1f us" % (B, name, ti, tr, ts, to))


if __name__ == "__main__":
    main()
