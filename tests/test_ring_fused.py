"""The border ring of a reflect-padded data gradient INSIDE the halo launch (r06; csrc/conv_halo16_bf16.inc RING, reference
networks.py:579-585 through autograd).

dx of ``conv(reflect_pad(x), w)`` on the bf16 path, at launch sizes that take the fused form (>= 512 four-wave workgroups), against
(a) torch's fp32 autograd of the same padded convolution on the same bf16-rounded operands and (b) the two-call form of rounds 2-5
(halo interior + strip GEMM + fold launch).  The error is reported per REGION of the image -- interior, the rows / columns the ring
folds onto, the corners where both meet -- so that a wrong ring term is seen where it lands; the fused form must be at least as close
to fp32 as the two-call form (its ring sums stay in the fp32 accumulators, the fold added them to the rounded interior)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hipdwc import _lib, ops          # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def bf16_mode():
    ops.set_precision("bf16")
    keep = ops.RING_FUSED
    yield
    ops.RING_FUSED = keep
    ops.set_precision("fp32")


def _regions(H, W, p):
    rows = torch.zeros(H, W, dtype=torch.bool)
    cols = torch.zeros(H, W, dtype=torch.bool)
    rows[1:p + 1] = True
    rows[H - 1 - p:H - 1] = True
    cols[:, 1:p + 1] = True
    cols[:, W - 1 - p:W - 1] = True
    return {"interior": ~(rows | cols), "ring rows": rows & ~cols, "ring columns": cols & ~rows, "corners": rows & cols}


# (B, channels of x / dx, channels of y / dY, H, W, K, residual add)
CASES = [
    (32, 256, 256, 32, 32, 3, False),     # ResBlock 3x3 at 32 x 32: every tile a corner tile
    (32, 256, 256, 32, 32, 3, True),      # ... with the identity-branch gradient riding on it
    (16, 256, 256, 32, 64, 3, False),     # left / right tiles with interior columns between them
    (16, 256, 256, 64, 32, 3, False),     # top / bottom tiles with interior rows
    (8, 256, 128, 64, 64, 5, False),      # first 5x5 (x: 256 channels at 64 x 64): corner, edge and interior tiles
    (4, 128, 64, 128, 128, 5, False),     # second 5x5 (x: 128 channels at 128 x 128)
    (8, 128, 64, 64, 128, 5, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_ring_matches_fp32_and_the_two_call_form(case):
    B, Cx, Cy, H, W, K, res = case
    p = (K - 1) // 2
    lib = _lib.load()
    assert lib.dwc_bf16_conv2d_bwd_data_same_fused_ok(B, H, W, Cx, Cy, K) == 1
    g = torch.Generator().manual_seed(B + Cx + Cy + H + W + K)
    x = torch.randn(B, Cx, H, W, generator=g).to(BF).to(DEV)
    w = (torch.randn(Cy, Cx, K, K, generator=g) / (Cx * K * K) ** 0.5).to(DEV)
    gy = torch.randn(B, Cy, H, W, generator=g).to(BF).to(DEV)

    # fp32 autograd of the padded convolution, bf16-rounded weights (what the kernels multiply with)
    xr = x.float().requires_grad_(True)
    F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), w.to(BF).float()).backward(gy.float())
    want = xr.grad
    r = None
    if res:         # the identity-branch gradient of a residual block, added by the kernel's epilogue (ops.ResGradToken)
        r = torch.randn(B, Cx, H, W, generator=g).to(BF).to(DEV).contiguous(memory_format=torch.channels_last)
        want = want + r.float()

    def run(fused):
        ops.RING_FUSED = fused
        xd = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.clone().requires_grad_(True)
        tok = ops.ResGradToken() if res else None
        y = ops.conv2d(xd, wd, None, 1, p, "none", token=tok)
        if res:
            tok.g = r
        y.backward(gy.contiguous(memory_format=torch.channels_last))
        return xd.grad.float(), wd.grad

    dx_f, dw_f = run(1)
    dx_u, dw_u = run(0)
    scale = want.abs().max().item()
    worst = {}
    for name, mask in _regions(H, W, p).items():
        m = mask.to(DEV)
        ef = ((dx_f - want).abs() * m).max().item() / scale
        eu = ((dx_u - want).abs() * m).max().item() / scale
        worst[name] = (ef, eu)
        print("%-13s fused %.3e  two-call %.3e  (of max |dx|)" % (name, ef, eu))
    for name, (ef, eu) in worst.items():
        assert ef <= 6e-3, (name, ef, eu)
        assert ef <= 1.25 * eu + 5e-4, (name, ef, eu)
    assert torch.equal(dw_f, dw_u)


# fp32 path (two-plane split products, csrc/conv_halo_x3.hip RING): (B, channels of x / dx, channels of y / dY, H, W, K, residual add)
CASES_F32 = [
    (16, 256, 256, 32, 32, 3, False),     # c1's ResBlock launch: 256 tiles, contraction split (two workgroups per tile)
    (16, 256, 256, 32, 32, 3, True),
    (48, 256, 256, 32, 32, 3, False),     # the 3B passes: whole tiles
    (8, 256, 256, 32, 64, 3, False),
    (8, 256, 256, 64, 32, 3, True),
    (16, 256, 128, 64, 64, 5, False),     # first 5x5 (x: 256 channels at 64 x 64)
    (16, 128, 64, 128, 128, 5, False),    # second 5x5 (x: 128 channels at 128 x 128)
    (2, 64, 64, 32, 48, 5, True),         # a small launch, three tiles across
]


@pytest.mark.parametrize("case", CASES_F32, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_ring_fp32_matches_float64_and_the_two_call_form(case):
    B, Cx, Cy, H, W, K, res = case
    p = (K - 1) // 2
    ops.set_precision("fp32")
    g = torch.Generator().manual_seed(B + Cx + Cy + H + W + K + 1)
    x = torch.randn(B, Cx, H, W, generator=g)
    w = torch.randn(Cy, Cx, K, K, generator=g) / (Cx * K * K) ** 0.5
    gy = torch.randn(B, Cy, H, W, generator=g)
    r = torch.randn(B, Cx, H, W, generator=g) if res else None
    # float64 on the host for a sample of the images (samples are independent through a convolution)
    idx = sorted({0, B // 2, B - 1})
    xr = x[idx].double().requires_grad_(True)
    F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), w.double()).backward(gy[idx].double())
    want = xr.grad + (r[idx].double() if res else 0.0)

    def run(fused):
        ops.RING_FUSED = fused
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.to(DEV).requires_grad_(True)
        tok = ops.ResGradToken() if res else None
        y = ops.conv2d(xd, wd, None, 1, p, "none", token=tok)
        if res:
            tok.g = r.to(DEV).contiguous(memory_format=torch.channels_last)
        y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
        return xd.grad.detach().cpu(), wd.grad.detach().cpu()

    dx_f, dw_f = run(1)
    dx_u, dw_u = run(0)
    scale = want.abs().max().item()
    for name, mask in _regions(H, W, p).items():
        ef = ((dx_f[idx].double() - want).abs() * mask).max().item() / scale
        eu = ((dx_u[idx].double() - want).abs() * mask).max().item() / scale
        d = ((dx_f - dx_u).abs() * mask).max().item() / scale
        print("%-13s fused %.2e  two-call %.2e  fused vs two-call (all images) %.2e  (of max |dx|)" % (name, ef, eu, d))
        assert ef <= 5e-6 and ef <= 2.0 * eu + 3e-7, (name, ef, eu)
        assert d <= 3e-6, (name, d)
    assert torch.equal(dw_f, dw_u)


# ---- the 4x4 stride-2 reflect-pad-1 layers (reference networks.py:90,94,437, networks_v2.py:107-111): padded row 0 folds onto dx row 1,
# row H + 1 onto H - 2, columns alike.  (B, channels of x / dx, channels of y / dY, H = W of x)
S2_CASES = [(16, 64, 128, 128), (16, 128, 256, 64), (48, 256, 256, 32), (16, 64, 128, 64), (128, 64, 128, 32)]


def _s2_regions(H):
    rows = torch.zeros(H, H, dtype=torch.bool)
    cols = torch.zeros(H, H, dtype=torch.bool)
    rows[1] = rows[H - 2] = True
    cols[:, 1] = cols[:, H - 2] = True
    return {"interior": ~(rows | cols), "ring rows": rows & ~cols, "ring columns": cols & ~rows, "corners": rows & cols}


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_ring_stride2(case, precision):
    B, Cx, Cy, H = case
    half = precision == "bf16"
    ops.set_precision(precision)
    g = torch.Generator().manual_seed(B + Cx + Cy + H + 5)
    x = torch.randn(B, Cx, H, H, generator=g)
    w = torch.randn(Cy, Cx, 4, 4, generator=g) / (Cx * 16) ** 0.5
    gy = torch.randn(B, Cy, H // 2, H // 2, generator=g)
    if half:
        x, gy = x.to(BF).float(), gy.to(BF).float()
    idx = sorted({0, B // 2, B - 1})
    xr = x[idx].double().requires_grad_(True)
    wr = (w.to(BF).float() if half else w).double()
    F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, stride=2).backward(gy[idx].double())
    want = xr.grad

    def run(fused):
        ops.RING_FUSED = fused
        xd = x.to(DEV).to(BF if half else torch.float32).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.to(DEV).requires_grad_(True)
        y = ops.conv2d(xd, wd, None, 2, 1, "none")
        y.backward(gy.to(DEV).to(y.dtype).contiguous(memory_format=torch.channels_last))
        return xd.grad.detach().float().cpu(), wd.grad.detach().cpu()

    dx_f, dw_f = run(1)
    dx_u, dw_u = run(0)
    scale = want.abs().max().item()
    for name, mask in _s2_regions(H).items():
        ef = ((dx_f[idx].double() - want).abs() * mask).max().item() / scale
        eu = ((dx_u[idx].double() - want).abs() * mask).max().item() / scale
        d = ((dx_f - dx_u).abs() * mask).max().item() / scale
        print("%-13s fused %.2e  two-call %.2e  fused vs two-call (all images) %.2e  (of max |dx|)" % (name, ef, eu, d))
        if half:
            assert ef <= 6e-3 and ef <= 1.25 * eu + 5e-4, (name, ef, eu)
        else:
            assert ef <= 5e-6 and ef <= 2.0 * eu + 3e-7 and d <= 3e-6, (name, ef, eu, d)
    assert torch.equal(dw_f, dw_u)

