"""Small operators of the launch diet (r06): nn.Linear on few rows (csrc/linear_small.hip), the GMM style-space KL term in one launch
each way (dwc_gmm_kl_sp_*), style-code L1 terms on the L1 kernels, and parameter concatenations that hang off ONE autograd node when a
module is used several times in a graph (hipdwc.ops.cat_params).  Each against torch's own fp32 / float64 result of the reference's
expression (reference networks.py:587-634, gmm.py:13-22, solver.py:113-114, networks_v2.py:116-121)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hipdwc import _lib, ops          # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def fp32_mode():
    ops.set_precision("fp32")
    yield
    ops.set_precision("fp32")


# ---- nn.Linear on few rows (csrc/linear_small.hip, r06): the AdaIN-parameter MLP and the style mapping (reference networks.py:491-503,
# 587-634, networks_v2.py:116-121) on the exact fp32 matrix instruction, against float64
@pytest.mark.parametrize("M,K,N,relu,bias", [(16, 64, 256, True, True), (16, 256, 256, True, True), (16, 256, 4096, False, True),
                                             (48, 256, 128, False, True), (3, 64, 32, True, False), (384, 256, 4096, False, True),
                                             (130, 48, 80, True, True), (32, 256, 256, False, False)])
def test_linear_small_matches_float64(M, K, N, relu, bias):
    ops.set_precision("fp32")
    lib = _lib.load()
    assert lib.dwc_linear_small_ok(M, N, K) == 1
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1 if bias else None
    gy = torch.randn(M, N, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.linear(xr, wr, br)
    yr = torch.relu(yr) if relu else yr
    (yr * gy.double()).sum().backward()
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    calls = []
    real = lib.dwc_linear_small_fwd
    try:
        lib.dwc_linear_small_fwd = lambda *a: (calls.append(1), real(*a))[1]
        yd = ops.linear(xd, wd, bd, "relu" if relu else "none")
    finally:
        lib.dwc_linear_small_fwd = real
    assert calls, "the small-linear kernel was not taken"
    (yd * gy.to(DEV)).sum().backward()
    for name, got, want in (("y", yd, yr), ("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad)) + ((("db", bd.grad, br.grad),) if bias else ()):
        err = (got.detach().cpu().double() - want.detach()).abs().max().item() / max(want.detach().abs().max().item(), 1e-30)
        print("%-3s max err %.2e of the largest magnitude" % (name, err))
        assert err <= 2e-6, (name, err)


def _kl_reference(pred_mus, pred_lv, mus, sigma):
    """The reference's loop over attributes (gmm.py:13-22), in the tensors' own precision."""
    kl = 0.0
    for i, mu in enumerate(pred_mus):
        var = pred_lv[i].exp()
        kl = kl + (0.5 * (torch.log(sigma / var) + (var + (mu - mus[:, i:i + 1]) ** 2) / sigma - 1.0)).sum(dim=1).mean()
    return kl


@pytest.mark.parametrize("B,K,D,extra", [(16, 8, 8, 0), (3, 8, 8, 2), (128, 8, 8, 0), (5, 3, 16, 1)])
def test_gmm_kl_one_launch_matches_reference_expression(B, K, D, extra):
    import gmm
    g = torch.Generator().manual_seed(B * 7 + K)
    mu = torch.randn(B, K * D, generator=g)
    lv = torch.randn(B, K * D, generator=g) * 0.7
    lab = (torch.rand(B, K + extra, generator=g) > 0.5).float() * 2 - 1
    sigma = torch.tensor(0.5 ** 2)
    mu64, lv64 = mu.double().requires_grad_(True), lv.double().requires_grad_(True)
    want = _kl_reference(list(mu64.split(D, 1)), list(lv64.split(D, 1)), lab.double(), sigma.double())
    want.backward()
    mud, lvd = mu.to(DEV).requires_grad_(True), lv.to(DEV).requires_grad_(True)
    calls = []
    lib = _lib.load()
    real = lib.dwc_gmm_kl_sp_fwd
    try:
        lib.dwc_gmm_kl_sp_fwd = lambda *a: (calls.append(1), real(*a))[1]
        got = gmm.gmm_kl_distance_sp(list(mud.split(D, 1)), list(lvd.split(D, 1)), lab.to(DEV), sigma.to(DEV))
    finally:
        lib.dwc_gmm_kl_sp_fwd = real
    assert calls, "the fused kernel was not taken"
    (got * 1.5).backward()
    assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item()), (got.item(), want.item())
    for name, a, b in (("dmu", mud.grad, mu64.grad * 1.5), ("dlv", lvd.grad, lv64.grad * 1.5)):
        err = (a.cpu().double() - b).abs().max().item() / b.abs().max().item()
        print("%s max err %.2e of the largest magnitude" % (name, err))
        assert err <= 2e-6, (name, err)
    # and the unfused expression (DWC_GMM_FUSED=0) agrees
    keep = ops.GMM_FUSED
    try:
        ops.GMM_FUSED = 0
        plain = gmm.gmm_kl_distance_sp(list(mud.detach().split(D, 1)), list(lvd.detach().split(D, 1)), lab.to(DEV), sigma.to(DEV))
    finally:
        ops.GMM_FUSED = keep
    assert abs(plain.item() - got.item()) <= 2e-6 * abs(got.item())


def test_style_l1_on_the_l1_kernels():
    from solver import Solver  # noqa: F401  (criterion_l1 is a method; exercised through the op it dispatches to)
    g = torch.Generator().manual_seed(3)
    a, z = torch.randn(16, 128, generator=g), torch.randn(16, 64, generator=g)
    ad, zd = a.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
    av = ad[:, :64]                                   # a non-contiguous slice, as the heads of one [B, 128] product are
    got = ops.l1_mean(av, zd)
    (got * 2.0).backward()
    a64, z64 = a.double().requires_grad_(True), z.double().requires_grad_(True)
    want = F.l1_loss(a64[:, :64], z64)
    (want * 2.0).backward()
    assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item())
    assert torch.equal(ad.grad.cpu().double(), a64.grad.float().double())
    assert torch.equal(zd.grad.cpu().double(), z64.grad.float().double())


def test_cat_params_one_node_for_several_uses():
    g = torch.Generator().manual_seed(11)
    ps = [torch.nn.Parameter(torch.randn(8, 32, generator=g).to(DEV)) for _ in range(4)]
    x1, x2 = torch.randn(5, 32, generator=g).to(DEV), torch.randn(7, 32, generator=g).to(DEV)
    w1 = ops.cat_params(ps)
    w2 = ops.cat_params(ps)
    assert w1 is w2 and w1.grad_fn is not None, "two uses in one graph must share the node"
    loss = (x1 @ w1.t()).square().sum() + (x2 @ w2.t()).sum()
    loss.backward()
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    wr = torch.cat(ref, 0)
    ((x1 @ wr.t()).square().sum() + (x2 @ wr.t()).sum()).backward()
    for p, r in zip(ps, ref):
        assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-5)
    # a second backward through the cached node (no optimiser step in between): gradients accumulate as usual
    (x1 @ ops.cat_params(ps).t()).sum().backward()
    (x1 @ torch.cat(ref, 0).t()).sum().backward()
    for p, r in zip(ps, ref):
        assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-5)
    # the parameters change: new values, new node
    with torch.no_grad():
        ps[2].add_(1.0)
    w3 = ops.cat_params(ps)
    assert w3 is not w1 and torch.equal(w3.detach(), torch.cat([p.detach() for p in ps], 0))
    with torch.no_grad():
        w4 = ops.cat_params(ps)
    assert w4.grad_fn is None and not w4.requires_grad
    # stack form
    bs = [torch.nn.Parameter(torch.randn(6, generator=g).to(DEV)) for _ in range(2)]
    s1 = ops.cat_params(bs, stack=True)
    assert s1.shape == (2, 6) and s1 is ops.cat_params(bs, stack=True)
    (s1 * torch.arange(12, device=DEV).view(2, 6)).sum().backward()
    assert torch.equal(bs[1].grad, torch.arange(6, 12, device=DEV).float())
