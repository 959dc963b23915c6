"""The fp32 7x7 weight-gradient kernel (smallk_wgrad_x3_kernel + reduce) through the C ABI at the c1 shapes: stems and heads, batches 16 / 48.
usage: python benchmarks/smallk_x3_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dwc-gan_amd"))
from hipdwc import _lib  # noqa: E402


def med(fn, n=12, skip=3, reps=4):
    ts = []
    for it in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        if it >= skip:
            ts.append(a.elapsed_time(e) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    H = W = 128
    for B in (16, 48):
        img = torch.randn(B, H, W, 4, device=dev)
        t64 = torch.randn(B, H, W, 64, device=dev)
        for heads in (0, 1):
            planes = 4 if heads else 3
            dw = torch.empty((planes, 64, 7, 7) if heads else (64, planes, 7, 7), device=dev)
            ws = torch.empty(lib.dwc_x3_conv7_smallk_wgrad_ws_bytes(B, H, W, heads), dtype=torch.uint8, device=dev)
            t = med(lambda: _lib.check(lib.dwc_x3_conv7_smallk_wgrad(img.data_ptr(), t64.data_ptr(), dw.data_ptr(), B, H, W, planes, heads,
                                                                     ws.data_ptr(), ws.numel(), st), "smallk"))
            gb = (img.numel() + t64.numel()) * 4 / 1e9
            fl = 2.0 * B * H * W * 64 * 49 * planes
            print("B%d %s: %.1f us  (operands %.0f MB -> %.2f TB/s; %.1f algorithmic TFLOP/s)" % (B, "heads" if heads else "stem ", t, gb * 1e3, gb / (t * 1e-6) / 1e3, fl / t * 1e-6))


if __name__ == "__main__":
    main()
