import os, sys, torch
sys.path.insert(0, "dwc-gan_amd")
from hipdwc import _lib, ops
sys.path.insert(0, "benchmarks")
import h2_ablation_bench as hb
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for B in [int(v) for v in (sys.argv[1:] or ("16", "32", "40", "48", "56", "64", "96"))]:
    ci = co = 256; H = 32; k = 3
    x = torch.randn(B, H, H, ci, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.05
    b = torch.zeros(co, device=dev); y = torch.empty(B, H, H, co, device=dev)
    hp = torch.zeros(lib.dwc_h2_weight_prepared_elems(co, ci, k), dtype=torch.float16, device=dev)
    wsl, wep = ops.amax_slot(dev); lib.dwc_absmax(w.data_ptr(), w.numel(), wsl, wep, st)
    lib.dwc_h2_weight_prepare(w.data_ptr(), hp.data_ptr(), co, ci, k, co, 0, wsl, wep, st)
    xsl, xep = ops.amax_slot(dev); lib.dwc_absmax(x.data_ptr(), x.numel(), xsl, xep, st)
    t = hb.med(lambda: lib.dwc_h2_conv2d_same_add_ws(x.data_ptr(), xsl, xep, hp.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, 0, B, H, H, ci, co, co, k, 1, 1, None, 0, None, st))
    tiles = B * 4 * 4
    print("B=%d tiles=%d (%.2f rounds of 512): %.1f us, %.3f us per tile, %.3f of 2.5PF" % (B, tiles, tiles / 512, t, t / tiles, 2.0 * B * H * H * ci * co * 9 * 3 / (t * 1e-6) / 2.5e15))
