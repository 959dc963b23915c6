"""BASELINE configs[2] (bf16 activations, batch 128 per GPU) under a checker AT ITS OWN BATCH.

Tile and kernel selection in the bf16 path depends on the grid (halo tile width by Cout, narrow / stem kernels at 3B = 384
images, split planners of the weight gradients, XCD remap of >= 16 blocks), so the small-batch cases of
tests/test_bf16_parity.py do not exercise the instantiations and grids the benchmark launches.  Here:

  * every hot convolution shape of the step at B = 128 and B = 384 (the 2B / 3B batched passes): forward, data gradient,
    weight and bias gradient.  Checker (a): the CPU oracle (oracle.conv_block, fp32) on a SAMPLE of the batch's images --
    first, middle, last: samples are independent through a convolution, so y and dx of those images must match whatever the
    batch around them is.  Checker (b), ALL images, through the size-independent properties of a convolution: y and dx of
    the big launch must equal those of the same images pushed through in chunks of 4 (the launch size the oracle checks in
    tests/test_bf16_parity.py; grids, tile choices and split plans differ), and the weight / bias gradients, which sum over
    the batch, must equal the sum of the chunks' gradients (linearity).  Tolerances are those of tests/test_bf16_parity.py.
  * forward passes of the generator (encode, decode) and the discriminator at B = 128 against the fp32 CPU oracle run in
    chunks of 16 (outside the text encoder samples are independent), per sample.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hipdwc import host, ops, synth          # noqa: E402
from oracle import dwcgan_oracle as orc      # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def rb(t):
    return t.to(BF).float()


def relerr(a, b):
    a, b = a.detach().float(), b.detach().float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def outliers(a, b, rel):
    """fraction of elements with |a-b| > rel * max|b|"""
    a, b = a.detach().float(), b.detach().float()
    return ((a - b).abs() > rel * b.abs().max()).float().mean().item()


@pytest.fixture(autouse=True)
def bf16_mode():
    ops.set_precision("bf16")
    yield
    ops.set_precision("fp32")


# (B, Cin, Cout, H, k, stride, pad, act): the conv calls of one c2 step with their real batch (B = 128 per-GPU batch; the
# decoder / re-encode / D passes are batched 2B and 3B)
HOT = [
    (128, 256, 256, 32, 3, 1, 1, "none"),     # ResBlock 3x3, halo BN = 256 (content encoder on x_real)
    (384, 256, 256, 32, 3, 1, 1, "none"),     # ... in the 3B decode / re-encode passes
    (128, 256, 128, 64, 5, 1, 2, "none"),     # first upsampling 5x5, halo BN = 128
    (384, 256, 128, 64, 5, 1, 2, "none"),
    (128, 128, 64, 128, 5, 1, 2, "none"),     # second upsampling 5x5, halo BN = 64
    (384, 128, 64, 128, 5, 1, 2, "none"),
    (384, 3, 64, 128, 7, 1, 3, "relu"),       # 7x7 stem on NHWC8 images (stem kernel, image data gradient on the narrow kernel)
    (384, 64, 128, 128, 4, 2, 1, "relu"),     # stride-2 4x4 (im2col GEMM, parity-class data gradient + fold)
    (384, 128, 256, 64, 4, 2, 1, "relu"),
    (384, 3, 64, 128, 4, 2, 1, "lrelu"),      # D stem on images, 3B
    (384, 256, 512, 16, 4, 2, 1, "lrelu"),    # D tail
    (384, 512, 512, 8, 4, 2, 1, "lrelu"),
]


def _gpu_randn(shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(shape, generator=g, device=DEV)


def _chunked(fn, xs, gys, chunk=4, stride=None):
    """Run fn(x_chunk, gy_chunk) -> (y, dx, dw, db) over chunks of the batch; returns y, dx concatenated over the visited
    chunks and dw, db summed.  stride: visit every stride-th chunk only (None: all)."""
    ys, dxs, dw, db, idx = [], [], None, None, []
    B = xs.shape[0]
    for c0 in range(0, B, chunk * (stride or 1)):
        y, dx, w, b = fn(xs[c0:c0 + chunk], gys[c0:c0 + chunk])
        ys.append(y)
        dxs.append(dx)
        dw = w if dw is None else dw + w
        db = b if db is None else db + b
        idx += list(range(c0, min(B, c0 + chunk)))
    return torch.cat(ys), torch.cat(dxs), dw, db, idx


@pytest.mark.parametrize("shape", HOT, ids=lambda s: "x".join(str(v) for v in s))
def test_bf16_hot_shapes_at_bench_batch(shape):
    B, ci, co, H, k, s, p, act = shape
    seed = sum(v for v in shape if isinstance(v, int)) + 3
    g = torch.Generator().manual_seed(seed)
    x = _gpu_randn((B, ci, H, H), seed).to(BF).float()                  # bf16-representable, generated on the device
    w = torch.randn(co, ci, k, k, generator=g) * (1.0 / (ci * k * k) ** 0.5)
    b = torch.randn(co, generator=g) * 0.1
    Ho = (H + 2 * p - k) // s + 1
    gy = _gpu_randn((B, co, Ho, Ho), seed + 1).to(BF).float()
    plain = act == "none"
    image = ci == 3
    wd0, bd0 = w.to(DEV), b.to(DEV)

    def hip(xc, gyc):
        """forward + backward of the op on a batch slice; returns detached y, dx (fp32, logical NCHW), dw, db"""
        if image:
            x0 = xc.clone().requires_grad_(True)
            xd = ops.pack_image(x0)
        else:
            x0 = xd = xc.to(BF).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = wd0.clone().requires_grad_(True), bd0.clone().requires_grad_(True)
        yd = ops.conv2d(xd, wd, bd, s, p, act)
        assert yd.dtype == BF and yd.shape == (xc.shape[0], co, Ho, Ho)
        (yd.float() * gyc).sum().backward()
        return yd.detach().float(), x0.grad.detach().float(), wd.grad.detach(), bd.grad.detach()

    y_big, dx_big, dw_big, db_big = hip(x, gy)
    # --- checker (a): the CPU oracle on sampled images ---
    idx = [0, B // 2, B - 1]
    xs = x[idx].cpu().requires_grad_(True)
    ys = orc.conv_block(xs, rb(w), b, s, p, act=act)
    (ys * gy[idx].cpu()).sum().backward()
    # behind an activation the tolerance of a MAXIMUM over 10^8 elements is wider than over the 10^5 of the small-batch tests
    tol_dx = 6e-3 if plain else 2.5e-2
    assert relerr(y_big[idx].cpu(), ys) <= 6e-3, ("y vs oracle", relerr(y_big[idx].cpu(), ys))
    assert relerr(dx_big[idx].cpu(), xs.grad) <= tol_dx, ("dx vs oracle", relerr(dx_big[idx].cpu(), xs.grad))
    # --- checker (b): every image, against the same op launched on chunks of 4 images ---
    y_c, dx_c, dw_c, db_c, seen = _chunked(hip, x, gy, chunk=4)
    assert len(seen) == B
    # identical operands and products, different grids / tile shapes / split plans: the stored bf16 values may differ by one
    # rounding where the fp32 sums were associated differently
    assert relerr(y_big, y_c) <= 4e-3, ("y big vs chunks", relerr(y_big, y_c))
    if plain:
        # (r06: the big launch of a stride-1 layer folds the border ring inside the halo kernel -- one rounding of interior + ring --, the
        # chunks of 4 take the strip GEMM + fold launch, which adds the ring to the ROUNDED interior: two roundings apart on the rows /
        # columns the ring folds onto, one elsewhere)
        ring = k in (3, 5) and s == 1
        assert relerr(dx_big, dx_c) <= (8e-3 if ring else 4e-3), ("dx big vs chunks", relerr(dx_big, dx_c))
        if ring:
            inner = (slice(None), slice(None), slice(p + 1, H - 1 - p), slice(p + 1, H - 1 - p))
            assert relerr(dx_big[inner], dx_c[inner]) <= 4e-3, ("dx big vs chunks, off the ring", relerr(dx_big[inner], dx_c[inner]))
    else:
        # behind ReLU / LeakyReLU the derivative is taken from the stored output's sign: where the fp32 sum is within rounding
        # noise of zero the two launches (different split-K / tile plans) can land on opposite signs, which changes ONE term
        # (0.9-1.0 * dy * w) of the dx sums around that pixel -- a handful of elements in 10^7, up to a few per cent of max|dx|
        # each.  Bound the population instead of the maximum: all but 5e-4 of the elements agree to one bf16 rounding
        # (measured r03: <= 7e-5 on the discriminator tails, 0 elsewhere) and no image is off as a whole (per_img below).
        assert outliers(dx_big, dx_c, 8e-3) <= 5e-4, ("dx big vs chunks: outliers", outliers(dx_big, dx_c, 8e-3))
        assert relerr(dx_big, dx_c) <= 1e-1, ("dx big vs chunks", relerr(dx_big, dx_c))
    per_img = (y_big - y_c).abs().flatten(1).mean(1) / y_c.abs().mean()
    assert per_img.max().item() <= 1e-3, ("an image of the big launch is off as a whole", per_img.argmax().item(), per_img.max().item())
    per_img = (dx_big - dx_c).abs().flatten(1).mean(1) / dx_c.abs().mean()
    # (a mis-addressed block or image would show as ~1; a few flipped derivative signs on an 8x8 map reach ~1e-2)
    assert per_img.max().item() <= (2e-3 if plain else 5e-2), ("dx of an image is off as a whole", per_img.argmax().item(), per_img.max().item())
    # (behind an activation: the same few flipped derivative signs, each worth one 0.9 * dy * x term of a sum whose maximum is ~250 terms' worth)
    assert relerr(dw_big, dw_c) <= (2e-4 if plain else 3e-2), ("dw vs sum over chunks", relerr(dw_big, dw_c))
    assert relerr(db_big, db_c) <= (2e-4 if plain else 3e-2), ("db vs sum over chunks", relerr(db_big, db_c))


@pytest.mark.parametrize("B", [128, 384])
def test_bf16_image_heads_at_bench_batch(B):
    """The fused tanh x3 + sigmoid heads (64 -> 8 planes, 7x7) at 128x128 on the narrow kernel, stem-form data gradient and
    small-channel weight gradient: CPU oracle on three images, chunk consistency / linearity on all."""
    C, H = 64, 128
    g = torch.Generator().manual_seed(B)
    x = _gpu_randn((B, C, H, H), B).to(BF).float()
    w = torch.randn(4, C, 7, 7, generator=g) * (1.0 / (C * 49) ** 0.5)
    b = torch.randn(4, generator=g) * 0.1
    gy = _gpu_randn((B, 4, H, H), B + 1).to(BF).float()
    wd0, bd0 = w.to(DEV), b.to(DEV)

    def hip(xc, gyc):
        xd = xc.to(BF).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = wd0.clone().requires_grad_(True), bd0.clone().requires_grad_(True)
        yd = ops.conv2d_heads(xd, torch.cat([wd, wd.new_zeros(4, C, 7, 7)], 0), torch.cat([bd, bd.new_zeros(4)], 0))
        assert yd.shape == (xc.shape[0], 8, H, H) and float(yd[:, 4:].abs().max()) == 0.0
        gy8 = torch.cat([gyc, torch.zeros_like(gyc)], 1)
        (yd.float() * gy8).sum().backward()
        return yd.detach().float()[:, :4], xd.grad.detach().float(), wd.grad.detach(), bd.grad.detach()

    def heads(pre):
        return torch.cat([torch.tanh(pre[:, :3]), torch.sigmoid(pre[:, 3:4])], 1)
    y_big, dx_big, dw_big, db_big = hip(x, gy)
    idx = [0, B // 2, B - 1]
    xs = x[idx].cpu().requires_grad_(True)
    ys = heads(orc.conv_block(xs, rb(w), b, 1, 3))
    (ys * gy[idx].cpu()).sum().backward()
    assert relerr(y_big[idx].cpu(), ys) <= 6e-3
    assert relerr(dx_big[idx].cpu(), xs.grad) <= 2.5e-2
    y_c, dx_c, dw_c, db_c, seen = _chunked(hip, x, gy, chunk=4)
    assert relerr(y_big, y_c) <= 4e-3 and outliers(dx_big, dx_c, 8e-3) <= 5e-4 and relerr(dx_big, dx_c) <= 1e-1
    assert relerr(dw_big, dw_c) <= 3e-2 and relerr(db_big, db_c) <= 3e-2


def test_bf16_forward_passes_b128_vs_oracle_chunks():
    """encode (content + style heads), decode (with attention head) and the two-scale discriminator at the c2 batch, forward
    only, against the fp32 CPU oracle evaluated in chunks of 16 samples.  bf16 activations through 10-20 layers: every output
    within 4e-2 of the tensor's scale at its worst element, within 5e-3 on average and within 1e-2 on average for every single
    sample (measured r03: worst 1.7e-2 (image), mean <= 2.1e-3, worst per-sample mean 4.4e-3 (D second scale))."""
    from solver import Solver
    B, S, CH = 128, 128, 16
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = synth.make_config(image_size=S, lstm_dropout=0.0)
    gcfg, dcfg = cfg["gen"], cfg["dis"]
    host.set_noise(host.HostNoise())
    try:
        torch.manual_seed(1234)
        s = Solver(cfg, torch.device(DEV), None).to(DEV)
        s.eval()
        G = {k: v.detach().cpu() for k, v in s.gen.state_dict().items()}
        D = {k: v.detach().cpu() for k, v in s.dis.state_dict().items()}
        batch = synth.make_batch(B, S, seed=21)
        x = batch["x_real"]
        g = torch.Generator().manual_seed(22)
        style = torch.randn(B, gcfg["num_cls"] * cfg["c_dim"], generator=g) * 0.5
        noise = orc.GlobalCpuNoise()
        with torch.no_grad():
            x4 = ops.pack_image(x.to(DEV))
            content, mus, lvs = s.gen.encode(x4)
            img, att = s.gen.decode(content, style.to(DEV))
            outs = s.dis(x4)
            ref = {"content": [], "mu": [], "img": [], "att": [], "src0": [], "cls0": [], "src1": [], "cls1": []}
            for c0 in range(0, B, CH):
                xc = x[c0:c0 + CH]
                rc, rmu, _ = orc.gen_encode(G, xc, gcfg, noise, training=False)
                # the decoder is compared on the HIP path's own content code (its input), so that this leg measures the decoder
                ri, ra = orc.gen_decode(G, content[c0:c0 + CH].float().cpu(), style[c0:c0 + CH], gcfg)
                ro = orc.dis_forward(D, xc, dcfg)
                ref["content"].append(rc)
                ref["mu"].append(torch.cat(rmu, 1))
                ref["img"].append(ri)
                ref["att"].append(ra)
                for sc in range(2):
                    ref["src%d" % sc].append(ro[sc][0])
                    ref["cls%d" % sc].append(ro[sc][1])
        got = {"content": content, "mu": torch.cat(mus, 1), "img": img, "att": att, "src0": outs[0][0], "cls0": outs[0][1],
               "src1": outs[1][0], "cls1": outs[1][1]}
        report = {}
        for k, parts in ref.items():
            r = torch.cat(parts).float()
            h = got[k].detach().float().cpu().reshape(r.shape)
            scale = r.abs().max().item()
            worst, mean = (h - r).abs().max().item() / scale, (h - r).abs().mean().item() / scale
            # per sample: no sample may be off as a whole (a mis-addressed block in a large grid would show here)
            per = ((h - r).abs().flatten(1).mean(1) / scale)
            report[k] = (round(worst, 5), round(mean, 6), round(per.max().item(), 6))
            assert worst <= 4e-2 and mean <= 5e-3 and per.max().item() <= 1e-2, (k, report[k])
        print("B=128 forward vs fp32 oracle (worst, mean, worst per-sample mean; relative to max|ref|):", report)
    finally:
        host.set_noise(host.DeviceNoise())


def test_bf16_narrow_persistent_form_equals_block_form(monkeypatch):
    """conv_narrow_persist_kernel (r06: one workgroup per CU walks block columns, filter taps in registers, next patch by LDS-DMA behind
    the tap loop) against the block-per-workgroup kernel it replaces at B >= 64, on the same operands, bit for bit: the image gradient of a
    7x7 stem and the tanh / sigmoid heads (same summation order, same activation code); the heads' activation against torch's."""
    B, C, H = 128, 64, 128
    g = torch.Generator().manual_seed(77)
    x = _gpu_randn((B, C, H, H), 5).to(BF).contiguous(memory_format=torch.channels_last)
    w8 = torch.cat([torch.randn(4, C, 7, 7, generator=g) * (1.0 / (C * 49) ** 0.5), torch.zeros(4, C, 7, 7)], 0).to(DEV)
    b8 = torch.cat([torch.randn(4, generator=g) * 0.1, torch.zeros(4)]).to(DEV)
    img = _gpu_randn((B, 8, H, H), 6)
    img[:, 3:] = 0
    img = img.to(BF).contiguous(memory_format=torch.channels_last)
    ws = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).to(DEV)
    gz = _gpu_randn((B, 64, H, H), 7).to(BF).contiguous(memory_format=torch.channels_last)
    res = {}
    for form in ("0", "1"):
        monkeypatch.setenv("DWC_NARROW_PERSIST", form)
        with torch.no_grad():
            y = ops.conv2d_heads(x, w8, b8).float()
        xi = img.clone().requires_grad_(True)
        ops.conv2d(xi, ws, None, 1, 3, "none").backward(gz)
        res[form] = (y, xi.grad.float())
    y0, dx0 = res["0"]
    y1, dx1 = res["1"]
    assert torch.equal(dx0, dx1), "image gradient: persistent vs block form"
    assert float(y1[:, 4:].abs().max()) == 0.0
    assert torch.equal(y0, y1), "heads: persistent vs block form (same summation order, same activation code)"
    # ... and the hardware exp2 / rcp activation of both against torch's tanh / sigmoid of the fp32 pre-activation: one bf16 step
    pre = F.conv2d(F.pad(x[:4].float(), (3, 3, 3, 3), mode="reflect"), w8[:4].to(BF).float(), b8[:4])
    want = torch.cat([torch.tanh(pre[:, :3]), torch.sigmoid(pre[:, 3:4])], 1)
    d = (y1[:4, :4] - want).abs()
    step = want.abs().clamp_min(2.0 ** -20) * 2.0 ** -7 + 2e-6   # one bf16 step at the value's magnitude (+ the fp32 sums' own rounding)
    assert bool((d <= step).all()), ("heads vs torch: more than one bf16 step", float((d / step).max()))
