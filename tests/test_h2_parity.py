"""fp32 convolutions on the f16 matrix cores through TWO-plane operand splits (csrc/conv_halo_x3.hip, NPL == 2; r05).

s*a = hi + lo (hi = f16(s*a), lo = f16(s*a - hi), round to nearest; s a per-tensor power of two from the tensor's
absmax slot), three MFMAs per fp32 MFMA-equivalent instead of the six of the three-plane bf16 split (tests/test_x3_parity.py).  The
claim under test is the same: this is an fp32 computation.  Gates are those of test_x3_parity.py -- error <= 5e-6 of the output
scale and <= 2x the larger of the native-fp32-MFMA / fp32-CPU errors (+2e-7) against float64 -- on the same shapes, PLUS operands
scaled to 1e-6 and 1e+4 (f16 alone would underflow / overflow there), a lognormal operand spanning 2^40, non-finite values, and the
slot protocol (a stale epoch must poison the result).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hipdwc import _lib, ops          # noqa: E402

DEV = "cuda:0"
ACT = {"none": 0, "relu": 1, "lrelu": 2}
FN = {"none": lambda v: v, "relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}


def _st():
    return torch.cuda.current_stream().cuda_stream


def _amax(t):
    lib = _lib.load()
    slot, ep = ops.amax_slot(t.device)
    _lib.check(lib.dwc_absmax(t.data_ptr(), t.numel(), slot, ep, _st()), "absmax")
    return slot, ep


def _prep(w, rows, dgrad):
    lib = _lib.load()
    Cout, Cin, K, _ = w.shape
    kdim = Cout if dgrad else Cin
    out = torch.empty(lib.dwc_h2_weight_prepared_elems(rows, kdim, K), dtype=torch.float16, device=w.device)
    slot, ep = _amax(w)
    _lib.check(lib.dwc_h2_weight_prepare(w.data_ptr(), out.data_ptr(), Cout, Cin, K, rows, int(dgrad), slot, ep, _st()), "h2_weight_prepare")
    return out


def _conv(xn, wp, bias, add, B, H, W, Cin, N, K, act, reflect, ws=None, tickets=None, amax=None):
    lib = _lib.load()
    y = torch.full((B, H, W, N), float("nan"), dtype=torch.float32, device=DEV)
    slot, ep = amax if amax is not None else _amax(xn)
    _lib.check(lib.dwc_h2_conv2d_same_add_ws(
        xn.data_ptr(), slot, ep, wp.data_ptr(), bias.data_ptr() if bias is not None else None, add.data_ptr() if add is not None else None,
        y.data_ptr(), None, 0, B, H, W, Cin, N, N, K, act, reflect, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
        tickets.data_ptr() if tickets is not None else None, _st()), "h2_conv2d_same_add_ws")
    return y


def _native(xd, wd, bd, stride, pad, act):
    """the same layer on the native fp32 MFMA kernels (the yardstick of 'fp32 accuracy' beside the fp32 CPU convolution)"""
    old = (ops.X3, ops.X3_S2)
    ops.X3, ops.X3_S2 = 0, 0
    old_mode = _lib.load().dwc_x3_gemm_mode(0)
    try:
        with torch.no_grad():
            return ops.conv2d(xd, wd, bd, stride, pad, act)
    finally:
        ops.X3, ops.X3_S2 = old
        _lib.load().dwc_x3_gemm_mode(old_mode)


# (B, Cin, Cout, H, W, K, act, x scale, w scale)
SHAPES = [
    (2, 256, 256, 32, 32, 3, "none", 1.0, 1.0),
    (40, 256, 256, 32, 32, 3, "relu", 1.0, 1.0),
    (3, 64, 128, 16, 48, 3, "lrelu", 1.0, 1.0),
    (1, 16, 64, 16, 16, 3, "none", 1.0, 1.0),
    (2, 32, 96, 32, 16, 3, "none", 1.0, 1.0),          # Cout 96: masked rows of the last tile
    (2, 256, 128, 32, 32, 5, "none", 1.0, 1.0),
    (1, 128, 64, 48, 32, 5, "relu", 1.0, 1.0),
    (9, 128, 64, 64, 64, 5, "none", 1.0, 1.0),
    (2, 256, 256, 32, 32, 3, "none", 1e-6, 1.0),       # gradient-sized activations: plain f16 planes would be subnormal
    (2, 256, 128, 32, 32, 5, "lrelu", 1e4, 1.0),       # plain f16 would overflow in the products' neighbourhood
    (2, 256, 128, 32, 32, 5, "none", 1e-6, 1e4),
    (2, 128, 64, 32, 32, 3, "none", 3e-20, 1e-8),      # far outside the f16 exponent range on both operands
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_h2_forward_matches_float64(shape):
    B, Cin, Cout, H, W, K, act, sx, sw = shape
    g = torch.Generator().manual_seed(sum(shape[:6]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3 * sx
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K) ** 0.5) * sw
    b = torch.randn(Cout, generator=g) * 0.1 * sx * sw
    pad = K // 2
    ref = FN[act](F.conv2d(F.pad(x.double(), (pad,) * 4, mode="reflect"), w.double(), b.double()))
    ref32 = FN[act](F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, b)).double()
    xd = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    wd, bd = w.to(DEV), b.to(DEV)
    y = _conv(xd, _prep(wd, Cout, False), bd, None, B, H, W, Cin, Cout, K, ACT[act], 1)
    yn = _native(xd.permute(0, 3, 1, 2), wd, bd, 1, pad, act)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (y.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() / scale
    err_native = (yn[:, :Cout].double().cpu() - ref).abs().max().item() / scale
    err_32 = (ref32 - ref).abs().max().item() / scale
    print("%s max err / scale vs float64: two-plane f16 %.2e | native fp32 MFMA path %.2e | fp32 CPU conv %.2e" % (
        "x".join(str(v) for v in shape), err, err_native, err_32))
    assert err <= 5e-6, err
    assert err <= 2 * max(err_native, err_32) + 2e-7


def test_h2_wide_dynamic_range_operand():
    """A lognormal operand (sigma 4: magnitudes over ~2^40).  Values below 2^-16 of the tensor's largest magnitude lose bits (f16
    subnormals) -- by at most 2^-38 of that largest magnitude each: invisible at the output scale."""
    B, Cin, Cout, H, W, K = 2, 128, 128, 32, 32, 3
    g = torch.Generator().manual_seed(77)
    x = torch.exp(4.0 * torch.randn(B, Cin, H, W, generator=g)) * torch.sign(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, K, K, generator=g) * 0.03
    ref = F.conv2d(F.pad(x.double(), (1,) * 4, mode="reflect"), w.double())
    ref32 = F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w).double()
    xd, wd = x.to(DEV).permute(0, 2, 3, 1).contiguous(), w.to(DEV)
    y = _conv(xd, _prep(wd, Cout, False), None, None, B, H, W, Cin, Cout, K, 0, 1)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (y.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() / scale
    err_32 = (ref32 - ref).abs().max().item() / scale
    print("lognormal operand: two-plane f16 %.2e | fp32 CPU conv %.2e" % (err, err_32))
    assert err <= 2e-6 and err <= 2 * err_32 + 2e-7


# (the five shapes of test_x3_parity.py::test_x3_stride2_forward_matches_float64, same data, + two with scaled activations)
@pytest.mark.parametrize("shape", [(2, 64, 128, 64, 32, "relu", 1.0), (3, 128, 256, 64, 64, "lrelu", 1.0), (1, 16, 64, 32, 32, "none", 1.0),
                                   (2, 256, 256, 32, 32, "none", 1.0), (24, 64, 128, 128, 128, "relu", 1.0),
                                   (2, 128, 256, 32, 32, "none", 1e-5), (2, 64, 128, 64, 32, "relu", 3e4)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_h2_stride2_forward_matches_float64(shape):
    B, Cin, Cout, H, W, act, sx = shape
    lib = _lib.load()
    assert lib.dwc_x3_conv2d_s2_ok(B, H, W, Cin, Cout)
    g = torch.Generator().manual_seed(sum(shape[:5]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3 * sx
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * (1.0 / (Cin * 16) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1 * sx
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = FN[act](F.conv2d(F.pad(xd.double(), (1,) * 4, mode="reflect"), wd.double(), bd.double(), stride=2))
    ref32 = FN[act](F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b, stride=2)).double() if B <= 3 else None
    xn = xd.permute(0, 2, 3, 1).contiguous()
    wp = _prep(wd, Cout, False)
    y = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.float32, device=DEV)
    slot, ep = _amax(xn)
    _lib.check(lib.dwc_h2_conv2d_s2_ws(xn.data_ptr(), slot, ep, wp.data_ptr(), bd.data_ptr(), y.data_ptr(), None, 0, B, H, W, Cin, Cout, Cout, ACT[act],
                                       None, 0, None, _st()), "h2_conv2d_s2")
    yn = _native(xd, wd, bd, 2, 1, act)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / scale
    err_native = (yn[:, :Cout].double() - ref).abs().max().item() / scale
    err_32 = (ref32 - ref.cpu()).abs().max().item() / scale if ref32 is not None else 0.0
    print("s2 %s max err / scale vs float64: two-plane f16 %.2e | native fp32 MFMA path %.2e | fp32 CPU conv %.2e" % (
        "x".join(str(v) for v in shape), err, err_native, err_32))
    assert err <= 5e-6, err
    assert err <= 2 * max(err_native, err_32) + 2e-7


@pytest.mark.parametrize("scale_dy", [1.0, 1e-7])
def test_h2_dgrad_interior_zero_rule(scale_dy):
    """reflect == 0 with dgrad-prepared weights == conv_transpose of dy (zero padding): the data-gradient interior."""
    B, Cin, Cout, H, W, K = 2, 128, 64, 32, 32, 5
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, H, W, generator=g) * scale_dy
    w = torch.randn(Cout, Cin, K, K, generator=g) * 0.05
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=K // 2)
    wd = w.to(DEV)
    dyd = dy.to(DEV).permute(0, 2, 3, 1).contiguous()
    dx = _conv(dyd, _prep(wd, Cin, True), None, None, B, H, W, Cout, Cin, K, 0, 0)
    torch.cuda.synchronize()
    err = (dx.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, err


@pytest.mark.parametrize("shape", [(2, 256, 128, 32, 32, 5, 1.0, 1.0), (3, 128, 64, 16, 48, 5, 1.0, 1.0), (2, 64, 128, 32, 16, 3, 1.0, 1.0),
                                   (5, 128, 64, 24, 32, 3, 1.0, 1e-6), (20, 64, 64, 64, 64, 5, 1.0, 1.0), (2, 64, 128, 64, 32, 4, 1.0, 1.0),
                                   (3, 128, 256, 32, 64, 4, 30.0, 1e-8), (16, 64, 128, 128, 128, 4, 1.0, 1.0)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_h2_weight_gradient_matches_float64(shape):
    """dW from fp32 x and dy, both split on the fly and scaled by their absmax slots (K = 4: the stride-2 form)."""
    B, Cin, Cout, H, W, K, sx, sdy = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(shape[:6]))
    stride, pad = (2, 1) if K == 4 else (1, K // 2)
    x = (torch.randn(B, Cin, H, W, generator=g) * sx).to(DEV)
    dy = (torch.randn(B, Cout, H // stride, W // stride, generator=g) * sdy).to(DEV)
    ref = torch.nn.grad.conv2d_weight(F.pad(x.double(), (pad,) * 4, mode="reflect"), (Cout, Cin, K, K), dy.double(), stride=stride)
    ref32 = torch.nn.grad.conv2d_weight(F.pad(x.cpu(), (pad,) * 4, mode="reflect"), (Cout, Cin, K, K), dy.cpu(), stride=stride).double()
    xd = x.permute(0, 2, 3, 1).contiguous()
    dyd = dy.permute(0, 2, 3, 1).contiguous()
    nws = lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, W, Cin, Cout, K)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    dw = torch.empty(Cout, Cin, K, K, dtype=torch.float32, device=DEV)
    (xs, xe), (ds, de) = _amax(xd), _amax(dyd)
    _lib.check(lib.dwc_h2_conv2d_wgrad(xd.data_ptr(), xs, xe, dyd.data_ptr(), ds, de, dw.data_ptr(), B, H, W, Cin, Cout, K, Cin, Cout,
                                       ws.data_ptr(), nws, _st()), "h2_conv2d_wgrad")
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (dw.double() - ref).abs().max().item() / scale
    err32 = (ref32 - ref.cpu()).abs().max().item() / scale
    print("%s dW max err / scale vs float64: two-plane f16 %.2e | fp32 CPU %.2e" % ("x".join(str(v) for v in shape), err, err32))
    assert err <= 2e-6 and err <= 2 * err32 + 2e-7


def test_h2_planes_reconstruct_the_weight():
    """(hi + lo) / s_w of the prepared planes against the fp32 weight, over ten decades of magnitude inside one tensor: to 2^-21.9
    relative for every element down to 2^-16 of the tensor's largest magnitude (the split keeps 22-24 significand bits while the
    residual is a normal f16), and to 2^-37.9 of that largest magnitude below (f16 subnormals)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    w = torch.randn(32, 16, 3, 3, generator=g) * torch.logspace(-6, 1, 32).view(32, 1, 1, 1)
    wd = w.to(DEV)
    out = _prep(wd, 32, False)
    torch.cuda.synchronize()
    n = lib.dwc_h2_weight_prepared_elems(32, 16, 3) - 8
    tail = out[n:n + 4].view(torch.float32).cpu()
    s, inv = float(tail[0]), float(tail[1])
    wmax = w.abs().max().item()
    assert s * inv == 1.0 and 2 ** 13 <= s * wmax < 2 ** 14
    pl = out[:n].view(9, 1, 2, 32, 2, 8).double().cpu()              # [tap][slab][plane][row][half][8]
    swap = ((torch.arange(32) >> 3) & 1).bool()
    pl[:, :, :, swap] = pl[:, :, :, swap].flip(4)
    pl = pl.reshape(9, 1, 2, 32, 16)
    total = (pl[:, 0, 0] + pl[:, 0, 1]) * inv
    want = w.permute(2, 3, 0, 1).reshape(9, 32, 16).double()
    err = (total - want).abs()
    lim = torch.maximum(want.abs() * 2.0 ** -21.9, torch.full_like(want, wmax * 2.0 ** -37.9))
    assert (err <= lim).all(), float((err / lim).max())
    assert (want.abs() < wmax * 2.0 ** -16).any() and (want.abs() > wmax * 2.0 ** -3).any()      # both regimes are exercised


def test_h2_stale_slot_poisons_and_nonfinite_propagates():
    B, Cin, Cout, H, W, K = 1, 64, 64, 16, 16, 3
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * 0.05).to(DEV)
    wp = _prep(w, Cout, False)
    slot, ep = _amax(x)
    good = _conv(x, wp, None, None, B, H, W, Cin, Cout, K, 0, 1, amax=(slot, ep))
    stale = _conv(x, wp, None, None, B, H, W, Cin, Cout, K, 0, 1, amax=(slot, ep + 1))
    torch.cuda.synchronize()
    assert torch.isfinite(good).all()
    assert torch.isnan(stale).all(), "a slot of another epoch must not pass for a measurement"
    # an inf and a NaN in the input reach (at least) every output whose 3x3 window holds them, nothing finite is invented there
    x2 = x.clone()
    x2[0, 5, 5, 3] = float("inf")
    x2[0, 12, 9, 7] = float("nan")
    y = _conv(x2, wp, None, None, B, H, W, Cin, Cout, K, 0, 1)
    torch.cuda.synchronize()
    assert not torch.isfinite(y[0, 4:7, 4:7]).any()
    assert not torch.isfinite(y[0, 11:14, 8:11]).any()


@pytest.mark.parametrize("shape", [(16, 256, 256, 32, 32, 3, 1, "relu", True), (4, 128, 256, 32, 32, 3, 1, "none", False),
                                   (2, 256, 128, 32, 32, 5, 1, "lrelu", True), (16, 128, 256, 64, 64, 4, 2, "lrelu", False)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_h2_contraction_split_of_small_launches(shape):
    """The contraction split of launches of at most 256 tiles (conv_halo_x3_kernel, KSP == 2) in the two-plane form: fp32 accuracy,
    bit-identical from run to run, tickets back at zero."""
    B, Cin, Cout, H, W, K, stride, act, with_add = shape
    lib = _lib.load()
    need = lib.dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cin, Cout, K, stride)
    assert need > 0, "shape is meant to be split"
    g = torch.Generator().manual_seed(sum(shape[:7]))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.rand(1, Cin, 1, 1, generator=g) * 3
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = H // stride, W // stride
    add = torch.randn(B, Cout, Ho, Wo, generator=g) if with_add else None
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    pad = 1 if stride == 2 else K // 2
    ref = FN[act](F.conv2d(F.pad(xd.double(), (pad,) * 4, mode="reflect"), wd.double(), bd.double(), stride=stride))
    if add is not None:
        ref = ref + add.to(DEV).double()
    xn = xd.permute(0, 2, 3, 1).contiguous()
    addn = add.to(DEV).permute(0, 2, 3, 1).contiguous() if add is not None else None
    wp = _prep(wd, Cout, False)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    tickets = torch.zeros(lib.dwc_x3_conv2d_ksplit_ticket_words(), dtype=torch.int32, device=DEV)
    amax = _amax(xn)

    def run(ws_t, tk):
        if stride == 1:
            return _conv(xn, wp, bd, addn, B, H, W, Cin, Cout, K, ACT[act], 1, ws_t, tk, amax)
        y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float32, device=DEV)
        _lib.check(lib.dwc_h2_conv2d_s2_ws(xn.data_ptr(), amax[0], amax[1], wp.data_ptr(), bd.data_ptr(), y.data_ptr(), None, 0, B, H, W, Cin, Cout, Cout,
                                           ACT[act], ws_t.data_ptr() if ws_t is not None else None, ws_t.numel() if ws_t is not None else 0,
                                           tk.data_ptr() if tk is not None else None, _st()), "h2_s2_ws")
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
    for it in range(10):
        if it % 2:
            filler = filler @ filler * 1e-4
        again = run(ws, tickets)
        assert torch.equal(again, first), "run %d differs" % it
    torch.cuda.synchronize()
    assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("B,ci,co,H,W", [(3, 64, 128, 32, 32), (2, 128, 256, 64, 32)])
def test_h2_stride2_data_gradient(B, ci, co, H, W):
    """Interior of the stride-2 4x4 data gradient (four output-parity classes of 2x2-tap zero-padded convolutions over dY) + the
    fp32 ring through ops.conv2d's backward, against float64 autograd."""
    g = torch.Generator().manual_seed(B + ci + co)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 4, 4, generator=g) * (1.0 / (ci * 16) ** 0.5)
    gy = torch.randn(B, co, H // 2, W // 2, generator=g) * 1e-4
    xr = x.double().requires_grad_(True)
    (F.conv2d(F.pad(xr, (1,) * 4, mode="reflect"), w.double(), stride=2) * gy.double()).sum().backward()
    assert ops.X3_PLANES == 2
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.conv2d(xd, w.to(DEV), None, 2, 1, "none")
    (y * gy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    err = (xd.grad.double().cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    assert err <= 2e-6, err


def test_h2_cached_slot_is_not_trusted_past_a_lap_of_the_pool():
    """A (slot, epoch) pair cached on a long-lived tensor must be measured again once the pool has handed the slot's neighbourhood
    out anew -- not meet the slot raised to another epoch by its next user (which would poison a correct tensor with NaN)."""
    x = torch.randn(1, 16, 16, 64, device=DEV)
    a0 = ops.amax_of(x)
    assert ops.amax_live(x) == a0 and ops.amax_of(x) == a0                # cache hit
    pool = ops._AMAX[torch.cuda.current_device()]
    pool[1] += ops.AMAX_SLOTS - 100              # ~a lap of allocations later (moving the counter FORWARD is always safe; never back)
    assert ops.amax_live(x) is None
    a1 = ops.amax_of(x)
    assert a1 != a0
    w = (torch.randn(64, 64, 3, 3, device=DEV) * 0.05)
    y = _conv(x, _prep(w, 64, False), None, None, 1, 16, 16, 64, 64, 3, 0, 1, amax=a1)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()


def test_contraction_split_dirty_ticket_is_reported_and_healed():
    """a ticket left dirty by an aborted launch makes both halves of its tile wait for a partner that never publishes.
    The wait is bounded and the tile is overwritten with NaN; the sticky status word behind the ticket row must be set, so that
    ops.ksplit_status_poll raises (instead of a run that silently turns NaN) and re-zeroes the row: the next launch is clean."""
    lib = _lib.load()
    B, H, W, Cin, N, K = 16, 32, 32, 128, 128, 3                       # 128 tiles: split whole
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(N, Cin, K, K, generator=g) * 0.03).to(DEV)
    wp = _prep(w, N, False)
    need = lib.dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cin, N, K, 1)
    assert need > 0
    ws, n, row_ptr = ops._x3_ksplit(lib, torch.device(DEV), B, H, W, Cin, N, K, 1)
    row = ops._X3_TICKETS[(0, _st())]
    assert row.numel() == lib.dwc_x3_conv2d_ksplit_ticket_words() and int(row[-1]) == 0
    ops.ksplit_status_poll(wait=True)                                  # drain a poll an earlier test may have started
    xa = _amax(x)

    def run():
        y = torch.empty(B, H, W, N, device=DEV)
        _lib.check(lib.dwc_h2_conv2d_same_add_ws(x.data_ptr(), xa[0], xa[1], wp.data_ptr(), None, None, y.data_ptr(), None, 0, B, H, W, Cin, N, N,
                                                 K, 0, 1, ws.data_ptr(), n, row_ptr, _st()), "h2_conv2d_same_add_ws")
        torch.cuda.synchronize()
        return y

    clean = run()
    assert torch.isfinite(clean).all() and int(row[-1]) == 0
    row[5] = 2                                                         # what a late first arriver of an aborted launch leaves behind
    bad = run()
    assert torch.isnan(bad).any() and int(row[-1]) == 1
    ops.ksplit_status_poll()                                           # starts the copy of the status word ...
    with pytest.raises(_lib.HipKernelError):
        ops.ksplit_status_poll(wait=True)                              # ... and finds it set
    assert int(row.abs().sum()) == 0
    again = run()
    assert torch.equal(again, clean)
