"""fp32 convolutions computed on the bf16 matrix cores through exact three-way operand splits (csrc/conv_halo_x3.hip).

The claim under test is that this is an fp32 computation, not a reduced-precision one: against a float64 reference the error
of the split-product kernel is the error of the native fp32 MFMA kernel (both are dominated by the fp32 accumulation), far
below what bf16 or tf32 operands would give (2^-9, 2^-11 relative per product).  Tolerances are those of test_hip_parity.py for
single fp32 ops: 2e-5 of the tensor's own scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hipdwc import _lib, ops          # noqa: E402

DEV = "cuda:0"
ACT = {"none": 0, "relu": 1, "lrelu": 2}


def _prep(lib, w, rows, dgrad):
    Cout, Cin, K, _ = w.shape
    kdim = Cout if dgrad else Cin
    n = lib.dwc_x3_weight_prepared_elems(rows, kdim, K)
    out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.dwc_x3_weight_prepare(w.data_ptr(), out.data_ptr(), Cout, Cin, K, rows, int(dgrad),
                                         torch.cuda.current_stream().cuda_stream), "x3_weight_prepare")
    return out


def _x3(lib, x_nhwc, wp, bias, B, H, W, Cin, N, rows, K, act, reflect):
    y = torch.empty(B, H, W, N, dtype=torch.float32, device=x_nhwc.device)
    _lib.check(lib.dwc_x3_conv2d_same(x_nhwc.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      y.data_ptr(), B, H, W, Cin, N, rows, K, act, reflect,
                                      torch.cuda.current_stream().cuda_stream), "x3_conv2d_same")
    return y


# (B, Cin, Cout, H, W, K, act): every template instantiation (BN 256/128/64 for 3x3, 128/64 for 5x5), several blocks per
# image, one slab and many, Cout not a multiple of the tile
SHAPES = [
    (2, 256, 256, 32, 32, 3, "none"),
    (40, 256, 256, 32, 32, 3, "relu"),      # >= 256 blocks: the 256-channel tile
    (3, 64, 128, 16, 48, 3, "lrelu"),
    (1, 16, 64, 16, 16, 3, "none"),         # one slab, one block
    (2, 32, 96, 32, 16, 3, "none"),         # Cout 96: masked rows of the last tile
    (2, 256, 128, 32, 32, 5, "none"),
    (1, 128, 64, 48, 32, 5, "relu"),
    (9, 128, 64, 64, 64, 5, "none"),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_x3_forward_matches_float64(shape):
    B, Cin, Cout, H, W, K, act = shape
    lib = _lib.load()
    assert lib.dwc_x3_conv2d_same_ok(B, H, W, Cin, Cout, K)
    g = torch.Generator().manual_seed(sum(shape[:6]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    pad = K // 2
    ref = F.conv2d(F.pad(x.double(), (pad,) * 4, mode="reflect"), w.double(), b.double())
    ref = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[act](ref)
    xd = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    wd, bd = w.to(DEV), b.to(DEV)
    wp = _prep(lib, wd, Cout, False)
    y = _x3(lib, xd, wp, bd, B, H, W, Cin, Cout, Cout, K, ACT[act], 1)
    torch.cuda.synchronize()
    got = y.permute(0, 3, 1, 2).double().cpu()
    scale = ref.abs().max().item()
    err_x3 = (got - ref).abs().max().item() / scale
    # yardsticks of "fp32 accuracy": the same product in plain fp32 on the CPU, and the native fp32 MFMA path of ops.conv2d
    ref32 = F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, b)
    ref32 = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[act](ref32).double()
    err_32 = (ref32 - ref).abs().max().item() / scale
    old = ops.X3
    ops.X3 = 0
    old_mode = _lib.load().dwc_x3_gemm_mode(0)           # the yardstick is the NATIVE fp32 MFMA product
    try:
        with torch.no_grad():
            yn = ops.conv2d(xd.permute(0, 3, 1, 2), wd, bd, 1, pad, act)
    finally:
        ops.X3 = old
        _lib.load().dwc_x3_gemm_mode(old_mode)
    err_native = (yn[:, :Cout].double().cpu() - ref).abs().max().item() / scale
    print("%s max err / scale vs float64: split-bf16 %.2e | native fp32 MFMA path %.2e | fp32 CPU conv %.2e" % (
        "x".join(str(v) for v in shape), err_x3, err_native, err_32))
    assert err_x3 <= 5e-6, err_x3                     # test_hip_parity's fp32 tolerance is 2e-5; bf16 operands give ~4e-3
    assert err_x3 <= 2 * max(err_native, err_32) + 2e-7


@pytest.mark.parametrize("shape", [(2, 64, 128, 64, 32, "relu"), (3, 128, 256, 64, 64, "lrelu"), (1, 16, 64, 32, 32, "none"),
                                   (2, 256, 256, 32, 32, "none"), (24, 64, 128, 128, 128, "relu")], ids=lambda s: "x".join(str(v) for v in s))
def test_x3_stride2_forward_matches_float64(shape):
    """The 4x4 stride-2 reflect-pad-1 layers on the split-product kernel (2x2 taps per input-pixel parity, space-to-depth in the patch
    gather; dwc_x3_conv2d_s2) against a float64 convolution: fp32 accuracy, as for the stride-1 layers."""
    B, Cin, Cout, H, W, act = shape
    lib = _lib.load()
    assert lib.dwc_x3_conv2d_s2_ok(B, H, W, Cin, Cout)
    g = torch.Generator().manual_seed(sum(shape[:5]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * (1.0 / (Cin * 16) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    fn = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[act]
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = fn(F.conv2d(F.pad(xd.double(), (1,) * 4, mode="reflect"), wd.double(), bd.double(), stride=2))
    ref32 = fn(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b, stride=2)).double() if B <= 3 else None
    xn = xd.permute(0, 2, 3, 1).contiguous()
    wp = _prep(lib, wd, Cout, False)
    y = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.float32, device=DEV)
    _lib.check(lib.dwc_x3_conv2d_s2(xn.data_ptr(), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, Cout, ACT[act],
                                    torch.cuda.current_stream().cuda_stream), "x3_conv2d_s2")
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err_x3 = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / scale
    old = ops.X3_S2
    ops.X3_S2 = 0
    old_mode = _lib.load().dwc_x3_gemm_mode(0)
    try:
        with torch.no_grad():
            yn = ops.conv2d(xd, wd, bd, 2, 1, act)
    finally:
        ops.X3_S2 = old
        _lib.load().dwc_x3_gemm_mode(old_mode)
    err_native = (yn[:, :Cout].double() - ref).abs().max().item() / scale
    err_32 = (ref32 - ref.cpu()).abs().max().item() / scale if ref32 is not None else 0.0
    print("s2 %s max err / scale vs float64: split-bf16 %.2e | native fp32 MFMA path %.2e | fp32 CPU conv %.2e" % (
        "x".join(str(v) for v in shape), err_x3, err_native, err_32))
    assert err_x3 <= 5e-6, err_x3
    assert err_x3 <= 2 * max(err_native, err_32) + 2e-7


def test_x3_dgrad_interior_zero_rule():
    """reflect == 0 with dgrad-prepared weights == conv_transpose of dy (zero padding): the data-gradient interior."""
    lib = _lib.load()
    B, Cin, Cout, H, W, K = 2, 128, 64, 32, 32, 5
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * 0.05
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=K // 2)          # [B, Cin, H, W]
    wd = w.to(DEV)
    wp = _prep(lib, wd, Cin, True)
    dyd = dy.to(DEV).permute(0, 2, 3, 1).contiguous()
    dx = _x3(lib, dyd, wp, None, B, H, W, Cout, Cin, Cin, K, 0, 0)
    torch.cuda.synchronize()
    got = dx.permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, err


@pytest.mark.parametrize("shape", [(2, 256, 128, 32, 32, 5), (3, 128, 64, 16, 48, 5), (2, 64, 128, 32, 16, 3), (5, 128, 64, 24, 32, 3),
                                   (20, 64, 64, 64, 64, 5)], ids=lambda s: "x".join(str(v) for v in s))
def test_x3_weight_gradient_matches_float64(shape):
    """dW of the reflect-padded convolution from fp32 x and dy, split products on both (activation) operands."""
    B, Cin, Cout, H, W, K = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    pad = K // 2
    xp = F.pad(x.double(), (pad,) * 4, mode="reflect")
    ref = torch.nn.grad.conv2d_weight(xp, (Cout, Cin, K, K), dy.double())
    ref32 = torch.nn.grad.conv2d_weight(F.pad(x, (pad,) * 4, mode="reflect"), (Cout, Cin, K, K), dy).double()
    xd = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    dyd = dy.to(DEV).permute(0, 2, 3, 1).contiguous()
    nws = lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, W, Cin, Cout, K)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    dw = torch.empty(Cout, Cin, K, K, dtype=torch.float32, device=DEV)
    _lib.check(lib.dwc_x3_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, K, Cin, Cout, ws.data_ptr(), nws,
                                       torch.cuda.current_stream().cuda_stream), "x3_conv2d_wgrad")
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (dw.double().cpu() - ref).abs().max().item() / scale
    err32 = (ref32 - ref).abs().max().item() / scale
    print("%s dW max err / scale vs float64: split-bf16 %.2e | fp32 CPU %.2e" % ("x".join(str(v) for v in shape), err, err32))
    assert err <= 2e-6 and err <= 2 * err32 + 2e-7


@pytest.mark.parametrize("shape", [(2, 64, 128, 64, 32), (3, 128, 256, 32, 64), (2, 64, 64, 16, 32), (16, 64, 128, 128, 128)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_x3_stride2_weight_gradient_matches_float64(shape):
    """dW of the 4x4 stride-2 reflect-pad-1 convolution on the split-product weight-gradient kernel (K = 4 form: the patch of a filter
    column is a stride-2 column set of the input, filter row kh of output row ks reads patch row 2 ks + kh)."""
    B, Cin, Cout, H, W = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g).to(DEV)
    ref = torch.nn.grad.conv2d_weight(F.pad(x.double(), (1,) * 4, mode="reflect"), (Cout, Cin, 4, 4), dy.double(), stride=2)
    ref32 = torch.nn.grad.conv2d_weight(F.pad(x.cpu(), (1,) * 4, mode="reflect"), (Cout, Cin, 4, 4), dy.cpu(), stride=2).double()
    xd = x.permute(0, 2, 3, 1).contiguous()
    dyd = dy.permute(0, 2, 3, 1).contiguous()
    nws = lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, W, Cin, Cout, 4)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    dw = torch.empty(Cout, Cin, 4, 4, dtype=torch.float32, device=DEV)
    _lib.check(lib.dwc_x3_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, 4, Cin, Cout, ws.data_ptr(), nws,
                                       torch.cuda.current_stream().cuda_stream), "x3_conv2d_wgrad")
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (dw.double() - ref).abs().max().item() / scale
    err32 = (ref32 - ref.cpu()).abs().max().item() / scale
    print("s2 %s dW max err / scale vs float64: split-bf16 %.2e | fp32 CPU %.2e" % ("x".join(str(v) for v in shape), err, err32))
    assert err <= 2e-6 and err <= 2 * err32 + 2e-7


def test_x3_split_is_exact():
    """The three bf16 planes written by dwc_x3_weight_prepare sum to the fp32 weight exactly (no rounding anywhere)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    w = torch.randn(32, 16, 3, 3, generator=g) * torch.logspace(-6, 3, 32).view(32, 1, 1, 1)
    wd = w.to(DEV)
    wp = _prep(lib, wd, 32, False).view(9, 1, 3, 32, 2, 8).float().cpu()      # [tap][slab][plane][row][half][8]
    swap = ((torch.arange(32) >> 3) & 1).bool()                                # rows 8-15, 24-31 store their halves swapped
    wp[:, :, :, swap] = wp[:, :, :, swap].flip(4)
    wp = wp.reshape(9, 1, 3, 32, 16)
    total = wp[:, 0, 0].double() + wp[:, 0, 1].double() + wp[:, 0, 2].double()              # [tap][co][ci]
    want = w.permute(2, 3, 0, 1).reshape(9, 32, 16).double()
    assert torch.equal(total, want)


GEMM_SHAPES = [(6, 256, 512, 16, 16, 4, 2, 1, "lrelu"),     # discriminator tail: too small for the halo form, split-K
               (3, 4, 64, 32, 32, 7, 1, 3, "relu"),          # 7x7 stem on an NHWC4 image (K = 196, image data gradient)
               (4, 128, 128, 8, 8, 3, 1, 1, "none"),         # 3x3 below the halo kernel's 16x16 block (ring strips in the data gradient)
               (5, 256, 256, 1, 1, 1, 1, 0, "relu")]         # nn.Linear as a 1x1 convolution


@pytest.mark.parametrize("shape", GEMM_SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_x3_im2col_gemm_and_wgrad_match_float64(shape, monkeypatch):
    """r04: the im2col kernels (conv_gemm_body / conv_wgrad_kernel of csrc/conv_igemm.hip) take
    their inner products as split products too (dwc_x3_gemm_mode, default 3).  Forward, data gradient and weight gradient through
    ops.conv2d against the float64 convolution, with the native fp32 MFMA result (mode 0) of the SAME kernels and the fp32 CPU
    convolution as yardsticks: error <= 5e-6 of the output scale and <= 2x the larger yardstick error (+2e-7)."""
    B, Cin, Cout, H, W, K, stride, pad, act = shape
    monkeypatch.setattr(ops, "X3", 0)                  # keep the layer on the im2col kernels
    monkeypatch.setattr(ops, "X3_S2", 0)
    monkeypatch.setattr(ops, "S2DGRAD", 0)
    g = torch.Generator().manual_seed(B + Cin + Cout + K)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    gy_shape = (B, Cout, (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1)
    gy = torch.randn(gy_shape, generator=g)
    fn = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[act]

    def cpu(dtype):
        xr, wr, br = (t.detach().clone().to(dtype).requires_grad_(True) for t in (x, w, b))    # (.to(float32) alone would alias x)
        xp = F.pad(xr, (pad,) * 4, mode="reflect") if pad else xr
        y = fn(F.conv2d(xp, wr, br, stride=stride))
        (y * gy.to(dtype)).sum().backward()
        return [t.detach().double() for t in (y, xr.grad, wr.grad)]

    ref, ref32 = cpu(torch.float64), cpu(torch.float32)
    lib = _lib.load()

    def gpu(mode):
        old = lib.dwc_x3_gemm_mode(mode)
        try:
            xd = x.detach().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
            y = ops.conv2d(xd, wd, bd, stride, pad, act)
            (y * gy.cuda()).sum().backward()
            torch.cuda.synchronize()
            return [t.detach().double().cpu() for t in (y, xd.grad, wd.grad)]
        finally:
            lib.dwc_x3_gemm_mode(old)

    got, native = gpu(3), gpu(0)
    for name, a, n_, c32, r in zip(("y", "dx", "dw"), got, native, ref32, ref):
        if name == "dx" and Cin < 8:
            r, a, n_, c32 = r[:, :3], a[:, :3], n_[:, :3], c32[:, :3]
        scale = r.abs().max().item()
        e, en, e32 = ((t - r).abs().max().item() / scale for t in (a, n_, c32))
        print("%s %s: split %.2e native %.2e cpu-fp32 %.2e" % ("x".join(str(v) for v in shape), name, e, en, e32))
        assert e <= 5e-6 and e <= 2 * max(en, e32) + 2e-7, (name, e, en, e32)


@pytest.mark.parametrize("shape", [(16, 256, 256, 32, 32, 3, 1, "relu", True), (4, 128, 256, 32, 32, 3, 1, "none", False),
                                   (2, 256, 128, 32, 32, 5, 1, "lrelu", True), (16, 128, 256, 64, 64, 4, 2, "lrelu", False),
                                   (4, 64, 128, 32, 64, 4, 2, "none", False),
                                   (36, 128, 256, 32, 32, 3, 1, "lrelu", True),        # 576 tiles: mixed mode = 512 whole + a split tail of 64
                                   (48, 128, 256, 64, 64, 4, 2, "relu", False)],       # stride 2, 768 tiles: 512 whole + 256 split
                         ids=lambda s: "x".join(str(v) for v in s))
def test_x3_contraction_split_of_small_launches(shape):
    """Launches of at most 256 tiles -- and the tail of launches whose last round of 512 resident workgroups would be at most half
    full -- run two workgroups per tile, each half of the channel slabs; whichever arrives second adds the other's half sum (conv_halo_x3_kernel, KSP == 2; dwc_x3_conv2d_same_add_ws / dwc_x3_conv2d_s2_ws).  Against float64 at fp32
    accuracy like the unsplit launch, bit-identical from run to run (the sum of two halves does not depend on who arrives first),
    tickets back at zero, and the second arriver really waited for the first: repeated 20 times on a busy device."""
    B, Cin, Cout, H, W, K, stride, act, with_add = shape
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    need = lib.dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cin, Cout, K, stride)
    if B >= 36 and need == 0:
        pytest.skip("tail split of larger launches is opt-in (DWC_X3_KSPLIT=2)")
    assert need > 0, "shape is meant to be split"
    g = torch.Generator().manual_seed(sum(shape[:7]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = H // stride, W // stride
    add = torch.randn(B, Cout, Ho, Wo, generator=g) if with_add else None
    fn = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[act]
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    pad = 1 if stride == 2 else K // 2
    ref = fn(F.conv2d(F.pad(xd.double(), (pad,) * 4, mode="reflect"), wd.double(), bd.double(), stride=stride))
    if add is not None:
        ref = ref + add.to(DEV).double()
    xn = xd.permute(0, 2, 3, 1).contiguous()
    addn = add.to(DEV).permute(0, 2, 3, 1).contiguous() if add is not None else None
    wp = _prep(lib, wd, Cout, False)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    tickets = torch.zeros(lib.dwc_x3_conv2d_ksplit_ticket_words(), dtype=torch.int32, device=DEV)

    def run(ws_t, tk):
        y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float32, device=DEV)
        wsp, wsn, tkp = (ws_t.data_ptr(), ws_t.numel(), tk.data_ptr()) if ws_t is not None else (None, 0, None)
        if stride == 1:
            _lib.check(lib.dwc_x3_conv2d_same_add_ws(xn.data_ptr(), wp.data_ptr(), bd.data_ptr(), addn.data_ptr() if addn is not None else None,
                                                     y.data_ptr(), B, H, W, Cin, Cout, Cout, K, ACT[act], 1, wsp, wsn, tkp, st), "same_add_ws")
        else:
            _lib.check(lib.dwc_x3_conv2d_s2_ws(xn.data_ptr(), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, Cout, ACT[act],
                                               wsp, wsn, tkp, st), "s2_ws")
        return y

    plain = run(None, None)
    first = run(ws, tickets)
    torch.cuda.synchronize()
    assert int(tickets.abs().sum()) == 0
    scale = ref.abs().max().item()
    err_split = (first.permute(0, 3, 1, 2).double() - ref).abs().max().item() / scale
    err_plain = (plain.permute(0, 3, 1, 2).double() - ref).abs().max().item() / scale
    print("%s max err / scale vs float64: split launch %.2e | plain launch %.2e" % ("x".join(str(v) for v in shape), err_split, err_plain))
    assert err_split <= 5e-6 and err_split <= 2 * err_plain + 2e-7
    filler = torch.randn(4096, 4096, device=DEV)
    for it in range(20):
        if it % 2:
            filler = filler @ filler * 1e-4        # other kernels in flight on the device while the pairs meet
        again = run(ws, tickets)
        assert torch.equal(again, first), "run %d differs" % it
    torch.cuda.synchronize()
    assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("heads", [0, 1])
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 21, 24), (3, 16, 40), (2, 9, 7)])
def test_x3_smallk_weight_gradient_matches_float64(B, H, W, heads):
    """r05: the 7x7 weight gradients of the stems (4-plane image -> 64 channels) and of the image heads (64 channels -> 4 planes) on
    smallk_wgrad_x3_kernel (csrc/conv_narrow_x3.hip) through the C ABI, against the float64 gradient of the reflect-padded
    convolution: error <= 5e-6 of the gradient's scale and <= 2x the fp32 CPU gradient's (+2e-7).  Sizes that are not multiples of the
    8x16-pixel unit included; plane 3 of the image is the attention plane for the heads and zero padding for the stems."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(100 * B + H + W + heads)
    planes = 4 if heads else 3
    img = torch.randn(B, planes, H, W, generator=g)
    t64 = torch.randn(B, 64, H, W, generator=g)

    def cpu(dtype):
        if heads:      # y = conv(pad(x64), w[4,64,7,7]); dY = img
            w = torch.zeros(planes, 64, 7, 7, dtype=dtype, requires_grad=True)
            y = F.conv2d(F.pad(t64.to(dtype), (3,) * 4, mode="reflect"), w)
            (y * img.to(dtype)).sum().backward()
        else:          # y = conv(pad(img), w[64,3,7,7]); dY = t64
            w = torch.zeros(64, planes, 7, 7, dtype=dtype, requires_grad=True)
            y = F.conv2d(F.pad(img.to(dtype), (3,) * 4, mode="reflect"), w)
            (y * t64.to(dtype)).sum().backward()
        return w.grad.double()

    ref, ref32 = cpu(torch.float64), cpu(torch.float32)
    img4 = torch.zeros(B, H, W, 4)
    img4[..., :planes] = img.permute(0, 2, 3, 1)
    img4 = img4.to(DEV).contiguous()
    t = t64.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.full(tuple(ref.shape), float("nan"), dtype=torch.float32, device=DEV)
    ws = torch.empty(lib.dwc_x3_conv7_smallk_wgrad_ws_bytes(B, H, W, heads), dtype=torch.uint8, device=DEV)
    _lib.check(lib.dwc_x3_conv7_smallk_wgrad(img4.data_ptr(), t.data_ptr(), dw.data_ptr(), B, H, W, planes, heads, ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream().cuda_stream), "x3_conv7_smallk_wgrad")
    torch.cuda.synchronize()
    got = dw.double().cpu()
    scale = ref.abs().max().item()
    e, e32 = (got - ref).abs().max().item() / scale, (ref32 - ref).abs().max().item() / scale
    print("smallk x3 heads=%d B%d %dx%d: split %.2e cpu-fp32 %.2e" % (heads, B, H, W, e, e32))
    assert e <= 5e-6 and e <= 2 * e32 + 2e-7, (e, e32)
