"""CPU ORACLE for the DWC-GAN training hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in this repo's own words and as explicit formulas over plain fp32
CPU tensors, what the reference computes on the path SURVEY.md section 8(a) lists:
generator encode/decode, the AdaIN style injection, the multi-scale discriminator, the
loss terms of ``dis_update``/``gen_update`` and the Adam/StepLR/EMA bookkeeping.  It is
the checker the HIP path is compared against; nothing in the product imports it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

PARITY PIN.  The reference ships no tests or golden vectors for this path
(SURVEY.md section 4), and its arithmetic lives in PyTorch itself (reference
environment.yaml:9 pins pytorch=0.4.1; the oracle available here is torch 2.10 CPU).  The
pin is therefore: outputs of the *imported, unmodified reference* run in the build
container, committed as tests/golden/*.npz|json by tests/golden/make_golden.py, and
checked by tests/test_oracle_golden.py (per-op forward/backward vectors, a tiny-config
three-iteration solver run including gradients / post-Adam weights / EMA copies, and
100-step loss trajectories at S=64 B=4 and S=128 B=16 that the oracle reproduces from
the same seed because it consumes the CPU random stream in exactly the reference's order).

Conventions: parameters live in flat dicts keyed by the reference's ``state_dict`` names
(e.g. ``enc_content.model.3.model.0.model.1.conv.weight``), tensors are NCHW fp32.
Each function cites the reference file:line it follows.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------


def reflect_indices(n, pad):
    """Source index for every position of a reflect-padded axis of length n+2*pad.

    torch.nn.ReflectionPad2d semantics (reference networks.py:530-531): the border
    element is not repeated, i.e. position -k maps to +k and n-1+k maps to n-1-k.
    """
    idx = torch.arange(-pad, n + pad)
    idx = idx.abs()
    idx = torch.where(idx >= n, 2 * (n - 1) - idx, idx)
    return idx


def pad_reflect(x, pad):
    if pad == 0:
        return x
    ih = reflect_indices(x.shape[2], pad)
    iw = reflect_indices(x.shape[3], pad)
    return x.index_select(2, ih).index_select(3, iw)


def activation(x, kind):
    """reference networks.py:556-571 (LeakyReLU slope is 0.1 for conv blocks, :559)."""
    if kind == "relu":
        return torch.clamp_min(x, 0.0)
    if kind == "lrelu":
        return torch.where(x > 0, x, 0.1 * x)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "sigmoid":
        return torch.sigmoid(x)
    assert kind in ("none", None), kind
    return x


def instance_norm(x, eps=1e-5):
    """Per-(n,c) mean / BIASED variance normalisation (nn.InstanceNorm2d, reference networks.py:545)."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def adain(x, weight, bias, eps=1e-5):
    """reference networks.py:706-719: batch_norm over a (1, B*C, H, W) view in training mode,
    i.e. instance norm followed by a per-(n,c) scale (weight = 'std') and shift (bias = 'mean')."""
    b, c = x.shape[0], x.shape[1]
    return instance_norm(x, eps) * weight.view(b, c, 1, 1) + bias.view(b, c, 1, 1)


def layer_norm_munit(x, gamma, beta, eps=1e-5):
    """reference networks.py:736-752: per-sample mean and UNBIASED std over C*H*W;
    eps is added to the std (not the variance); per-channel affine.  The B==1 branch
    of the reference (:739-742) evaluates the same formula."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    n = flat.shape[1]
    mu = flat.mean(dim=1)
    std = torch.sqrt(((flat - mu[:, None]) ** 2).sum(dim=1) / (n - 1))
    y = (x - mu.view(b, 1, 1, 1)) / (std.view(b, 1, 1, 1) + eps)
    return y * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def _bilinear_taps(n_in):
    """align_corners=False x2 taps: out[2i] = .25*in[i-1] + .75*in[i], out[2i+1] = .75*in[i] + .25*in[i+1],
    with the neighbour index clamped to the edge."""
    o = torch.arange(2 * n_in)
    src = (o.float() + 0.5) / 2.0 - 0.5
    src = torch.clamp_min(src, 0.0)
    i0 = src.floor().long()
    i1 = torch.clamp_max(i0 + 1, n_in - 1)
    w1 = src - i0.float()
    return i0, i1, w1


def upsample_bilinear2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear') (reference networks_v2.py:154)."""
    h0, h1, wh = _bilinear_taps(x.shape[2])
    w0, w1, ww = _bilinear_taps(x.shape[3])
    wh = wh.view(1, 1, -1, 1)
    ww = ww.view(1, 1, 1, -1)
    rows = x.index_select(2, h0) * (1 - wh) + x.index_select(2, h1) * wh
    return rows.index_select(3, w0) * (1 - ww) + rows.index_select(3, w1) * ww


def downsample_half(x):
    """F.interpolate(x, 0.5, 'bilinear') (reference networks.py:113) == 2x2 mean for even sizes."""
    b, c, h, w = x.shape
    return x.reshape(b, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


def conv_block(x, w, b, stride, pad, norm="none", act="none", gamma=None, beta=None,
               adain_w=None, adain_b=None):
    """act(norm(conv(reflect_pad(x)))) — reference networks.py:579-585."""
    y = F.conv2d(pad_reflect(x, pad), w, b, stride=stride)
    if norm == "in":
        y = instance_norm(y)
    elif norm == "adain":
        y = adain(y, adain_w, adain_b)
    elif norm == "ln":
        y = layer_norm_munit(y, gamma, beta)
    else:
        assert norm == "none"
    return activation(y, act)


def linear(x, w, b):
    return x @ w.t() + b


# --------------------------------------------------------------------------------------
# noise: the places where the reference consumes the (CPU) random stream, in its order
# --------------------------------------------------------------------------------------
class GlobalCpuNoise:
    """Draws from torch's global CPU generator with the same calls / shapes / order as the
    reference running on CPU, so a seeded oracle run reproduces a seeded reference run."""

    def dropout(self, x, p, training=True):
        return F.dropout(x, p=p, training=training)

    def style_sample(self, mu, c_dim, stddev):
        """reference tools.py:65-70.  Normal(mu, s).sample((1, c_dim)) has shape
        (1, c_dim, B, A); the two transposes give (1, B, A, c_dim) -> view (B, A*c_dim),
        i.e. attribute-major rows [a0 x c_dim, a1 x c_dim, ...]."""
        shape = (1, c_dim) + tuple(mu.shape)
        draw = torch.normal(mu.expand(shape), (torch.ones_like(mu) * stddev).expand(shape))
        return draw.permute(0, 2, 3, 1).reshape(mu.shape[0], -1)

    def rand(self, shape):
        """torch.rand(...) of the gradient penalty's interpolation weights (reference solver.py:339)."""
        return torch.rand(tuple(shape))


# --------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------
def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def style_encoder(P, x, cfg, noise, training=True):
    """reference networks_v2.py:98-141.  P keys are relative to 'enc_style.'."""
    n_down = cfg["style_downsample"]
    h = conv_block(x, P["model.0.conv.weight"], P["model.0.conv.bias"], 1, 3, act=cfg["activ"])
    for i in range(1, n_down + 1):
        h = conv_block(h, P["model.%d.conv.weight" % i], P["model.%d.conv.bias" % i], 2, 1, act=cfg["activ"])
    f = h.mean(dim=(2, 3))                                      # AdaptiveAvgPool2d(1), :113
    if cfg["use_map"]:                                          # :116-121
        f = torch.clamp_min(linear(f, P["mapping.0.weight"], P["mapping.0.bias"]), 0)
        f = noise.dropout(f, 0.1, training)
        f = torch.clamp_min(linear(f, P["mapping.3.weight"], P["mapping.3.bias"]), 0)
    mus = [linear(f, P["fcs.%d.weight" % i], P["fcs.%d.bias" % i]) for i in range(cfg["num_cls"])]
    lvs = [linear(f, P["fcvars.%d.weight" % i], P["fcvars.%d.bias" % i]) for i in range(cfg["num_cls"])]
    return mus, lvs


def res_blocks(P, x, n_res, norm, act, adain_params=None):
    """reference networks.py:480-522.  P keys relative to '<...>.model.' of the ResBlocks."""
    for r in range(n_res):
        kw = [{}, {}]
        if norm == "adain":
            kw = [dict(adain_w=adain_params[2 * r][0], adain_b=adain_params[2 * r][1]),
                  dict(adain_w=adain_params[2 * r + 1][0], adain_b=adain_params[2 * r + 1][1])]
        pre = "%d.model." % r
        h = conv_block(x, P[pre + "0.conv.weight"], P[pre + "0.conv.bias"], 1, 1, norm=norm, act=act, **kw[0])
        h = conv_block(h, P[pre + "1.conv.weight"], P[pre + "1.conv.bias"], 1, 1, norm=norm, act="none", **kw[1])
        x = h + x
    return x


def content_encoder(P, x, cfg):
    """reference networks.py:428-446.  P keys relative to 'enc_content.'."""
    act = cfg["activ"]
    h = conv_block(x, P["model.0.conv.weight"], P["model.0.conv.bias"], 1, 3, norm="in", act=act)
    n_down = cfg["content_downsample"]
    for i in range(1, n_down + 1):
        h = conv_block(h, P["model.%d.conv.weight" % i], P["model.%d.conv.bias" % i], 2, 1, norm="in", act=act)
    return res_blocks(_sub(P, "model.%d.model." % (n_down + 1)), h, cfg["n_res"], "in", act)


def mlp(P, style):
    """reference networks.py:491-503 with n_blk=3 (networks_v2.py:52): Linear-ReLU, Linear-ReLU, Linear."""
    h = torch.clamp_min(linear(style, P["model.0.fc.weight"], P["model.0.fc.bias"]), 0)
    h = torch.clamp_min(linear(h, P["model.1.fc.weight"], P["model.1.fc.bias"]), 0)
    return linear(h, P["model.2.fc.weight"], P["model.2.fc.bias"])


def split_adain_params(flat, n_layers, channels):
    """reference networks_v2.py:78-87: per AdaIN layer (in module order) the first C columns are
    the shift ('mean' -> bias) and the next C the scale ('std' -> weight), flattened to B*C."""
    out = []
    for j in range(n_layers):
        base = 2 * channels * j
        shift = flat[:, base:base + channels].contiguous().view(-1)
        scale = flat[:, base + channels:base + 2 * channels].contiguous().view(-1)
        out.append((scale, shift))
    return out


def decoder(P, content, adain_flat, cfg):
    """reference networks_v2.py:144-169.  P keys relative to 'dec.'.  Returns (image, attention)."""
    act = cfg["activ"]
    n_res = cfg["n_res"]
    ch = content.shape[1]
    ap = split_adain_params(adain_flat, 2 * n_res, ch)
    h = res_blocks(_sub(P, "model.0.model."), content, n_res, "adain", act, ap)
    for u in range(cfg["content_downsample"]):
        idx = 2 + 2 * u                                      # Sequential: [ResBlocks, Up, Conv, Up, Conv]
        h = upsample_bilinear2x(h)
        h = conv_block(h, P["model.%d.conv.weight" % idx], P["model.%d.conv.bias" % idx], 1, 2, norm="ln",
                       act=act, gamma=P["model.%d.norm.gamma" % idx], beta=P["model.%d.norm.beta" % idx])
    img = conv_block(h, P["image_content.conv.weight"], P["image_content.conv.bias"], 1, 3, act="tanh")
    att = None
    if cfg["use_attention"]:                                 # the decoder's own flag (networks_v2.py:167)
        att = conv_block(h, P["image_attention.conv.weight"], P["image_attention.conv.bias"], 1, 3, act="sigmoid")
    return img, att


def txt_encoder(P, style, tokens, lens, cfg, noise, training=True):
    """reference networks_v2.py:213-254, including the cat(dim=1).view(B,-1) step (:249) that
    mixes samples of the local batch.  P keys relative to 'enc_txt.'."""
    hidden, layers = cfg["hidden_size"], cfg["num_layers"]
    tok = tokens.t()
    seq_len, bsz = tok.shape
    lens_sorted, order = torch.sort(lens, descending=True)
    tok = tok.index_select(1, order)
    sty = style.index_select(0, order)
    emb = F.embedding(tok, P["embed_tokens.weight"], padding_idx=0)
    emb = noise.dropout(emb, cfg["dropout_in"], training)
    inp = torch.cat([emb, sty.expand(seq_len, -1, -1)], dim=-1)
    packed = torch.nn.utils.rnn.pack_padded_sequence(inp, lens_sorted.tolist())
    rng = torch.get_rng_state()       # the throw-away container's own init must not eat the stream
    lstm = torch.nn.LSTM(inp.shape[-1], hidden, layers, dropout=cfg["dropout_out"] if layers > 1 else 0.0,
                         bidirectional=True)
    torch.set_rng_state(rng)
    lstm.train(training)
    lp = {k[len("lstm."):]: v for k, v in P.items() if k.startswith("lstm.")}
    zeros = inp.new_zeros(2 * layers, bsz, hidden)
    outs, (h_n, c_n) = torch.func.functional_call(lstm, lp, (packed, (zeros, zeros)))
    mem, _ = torch.nn.utils.rnn.pad_packed_sequence(outs)
    noise.dropout(mem, cfg["dropout_out"], training)          # result unused by the reference, but it draws (:236)

    def merge(t):  # (layers*2, B, H) -> (layers, B, 2H)
        return t.view(layers, 2, bsz, -1).transpose(1, 2).contiguous().view(layers, bsz, -1)
    h_n, c_n = merge(h_n), merge(c_n)
    inverse = torch.sort(order)[1]
    h_n, c_n = h_n.index_select(1, inverse), c_n.index_select(1, inverse)
    feat = torch.cat([h_n, c_n], dim=1).view(bsz, -1)
    mus = [linear(feat, P["fcs.%d.weight" % i], P["fcs.%d.bias" % i]) for i in range(cfg["num_cls"])]
    lvs = [linear(feat, P["fcvars.%d.weight" % i], P["fcvars.%d.bias" % i]) for i in range(cfg["num_cls"])]
    return mus, lvs


def gen_encode(G, x, cfg, noise, training=True):
    """reference networks_v2.py:61-65 (style encoder first, then content encoder)."""
    mus, lvs = style_encoder(_sub(G, "enc_style."), x, cfg, noise, training)
    content = content_encoder(_sub(G, "enc_content."), x, cfg)
    return content, mus, lvs


def gen_decode(G, content, style, cfg):
    """reference networks_v2.py:71-76."""
    flat = mlp(_sub(G, "mlp."), style.view(style.shape[0], -1))
    return decoder(_sub(G, "dec."), content, flat, cfg)


def gen_encode_txt(G, style, tokens, lens, cfg, noise, training=True):
    return txt_encoder(_sub(G, "enc_txt."), style, tokens, lens, cfg, noise, training)


# --------------------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------------------
def dis_forward(D, x, cfg):
    """reference networks.py:102-114 — list over scales of [src map, cls logits]."""
    outs = []
    for s in range(cfg["num_scales"]):
        h = x
        for l in range(cfg["n_layer"]):
            pre = "cnns_feat.%d.%d.conv." % (s, l)
            h = conv_block(h, D[pre + "weight"], D[pre + "bias"], 2, 1, act=cfg["activ"])
        src = F.conv2d(h, D["cnns_src.%d.weight" % s], D["cnns_src.%d.bias" % s])
        cls = F.conv2d(h, D["cnns_cls.%d.weight" % s]).view(h.shape[0], -1)
        outs.append((src, cls))
        x = downsample_half(x)
    return outs


def bce_with_logits_mean(z, t):
    """F.binary_cross_entropy_with_logits(..., 'mean') (reference networks.py:83):
    max(z,0) - z*t + log(1 + exp(-|z|))."""
    return (torch.clamp_min(z, 0) - z * t + torch.log1p(torch.exp(-z.abs()))).mean()


def calc_dis_loss(D, fake, real, real_cls, w_gan, w_cls, cfg):
    """reference networks.py:116-146, lsgan branch."""
    loss = 0.0
    for (fs, _), (rs, rc) in zip(dis_forward(D, fake, cfg), dis_forward(D, real, cfg)):
        loss = loss + ((fs ** 2).mean() + ((rs - 1) ** 2).mean()) * w_gan
        loss = loss + bce_with_logits_mean(rc, real_cls) * w_cls
    return loss


def calc_gen_loss(D, fake, target_cls, w_gan, w_cls, cfg):
    """reference networks.py:148-170, lsgan branch."""
    loss = 0.0
    for fs, fc in dis_forward(D, fake, cfg):
        loss = loss + ((fs - 1) ** 2).mean() * w_gan
        loss = loss + bce_with_logits_mean(fc, target_cls) * w_cls
    return loss


def gmm_kl_sp(mus, logvars, c, sigma2):
    """reference gmm.py:13-22: sum over attributes of mean_n sum_d KL(N(mu, e^lv) || N(c_i, sigma2))."""
    total = 0.0
    for i, (m, lv) in enumerate(zip(mus, logvars)):
        v = lv.exp()
        total = total + (0.5 * (torch.log(sigma2 / v) + (v + (m - c[:, i:i + 1]) ** 2) / sigma2 - 1.0)).sum(1).mean()
    return total


def gmm_em_sp(mus, c):
    """reference gmm.py:33-41."""
    total = 0.0
    for i, m in enumerate(mus):
        total = total + (m - c[:, i:i + 1]).abs().sum(1).mean()
    return total


def l1_mean(a, b):
    return (a - b).abs().mean()


# --------------------------------------------------------------------------------------
# optimiser bookkeeping restated
# --------------------------------------------------------------------------------------
# ---- VGG16 perceptual loss (reference networks.py:639-688, solver.py:242-247, utils.py:207-217) -----------------------
VGG_CFG = ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3))       # (block, number of 3x3 convs); 2x2 max pool after blocks 1-3


def vgg_preprocess(batch):
    """RGB -> BGR, [-1, 1] -> [0, 255], subtract the ImageNet BGR means (reference utils.py:207-217)."""
    r, g, b = torch.chunk(batch, 3, dim=1)
    out = (torch.cat((b, g, r), dim=1) + 1) * 255 * 0.5
    return out - torch.tensor([103.939, 116.779, 123.680], dtype=out.dtype).view(1, 3, 1, 1)


def vgg16_relu5_3(sd, x):
    """sd: state_dict with conv{b}_{i}.weight/.bias (reference networks.py:642-659); zero padding 1, ReLU after every
    conv, max pooling after blocks 1-3 (reference networks.py:661-686)."""
    h = x
    for blk, n in VGG_CFG:
        for i in range(n):
            h = F.relu(F.conv2d(h, sd["conv%d_%d.weight" % (blk, i + 1)], sd["conv%d_%d.bias" % (blk, i + 1)], padding=1))
        if blk <= 3:
            h = F.max_pool2d(h, kernel_size=2, stride=2)
    return h


def vgg_loss(sd, img, target):
    """reference solver.py:242-247 with nn.InstanceNorm2d(512, affine=False) (solver.py:56)."""
    a = F.instance_norm(vgg16_relu5_3(sd, vgg_preprocess(img)), eps=1e-5)
    b = F.instance_norm(vgg16_relu5_3(sd, vgg_preprocess(target)), eps=1e-5)
    return torch.mean((a - b) ** 2)


class AdamState:
    """torch.optim.Adam as the reference configures it (reference solver.py:62-68): coupled L2
    weight decay added to the gradient, bias-corrected moments, eps outside the sqrt.
    Parameters whose gradient is None are skipped entirely (SURVEY.md section 7 quirk viii)."""

    def __init__(self, params, lr, beta1, beta2, weight_decay, eps=1e-8):
        self.lr, self.b1, self.b2, self.wd, self.eps = lr, beta1, beta2, weight_decay, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    @torch.no_grad()
    def step(self, params, grads):
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            self.t[k] += 1
            t = self.t[k]
            g = g + self.wd * p
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


class OracleSolver:
    """Functional restatement of reference solver.py:22-413 restricted to the training step.

    ``gen``/``dis`` are OrderedDicts of fp32 CPU tensors keyed like the reference's
    state_dicts (buffers such as the AdaIN running stats are accepted and ignored).
    """

    def __init__(self, cfg, gen_params, dis_params, noise=None, as_written=False, vgg_params=None):
        self.cfg = cfg
        self.vgg = vgg_params          # frozen Vgg16 state_dict; needed when cfg["vgg_w"] > 0 (reference solver.py:79-83)
        self.noise = noise or GlobalCpuNoise()
        self.as_written = as_written   # True: back-propagate everything like .backward() does
        keep = lambda k: "running_" not in k
        self.gen = OrderedDict((k, v.detach().clone().float().requires_grad_(True))
                               for k, v in gen_params.items() if keep(k))
        self.dis = OrderedDict((k, v.detach().clone().float().requires_grad_(True))
                               for k, v in dis_params.items() if keep(k))
        self.gcfg = dict(cfg["gen"])
        self.dcfg = dict(cfg["dis"])
        self.use_attention = cfg["gen"]["use_attention"]      # reference solver.py:43
        self.att_status = self.use_attention
        self.init_ds_w = cfg["ds_w"]
        self.c_dim, self.stddev = cfg["c_dim"], cfg["stddev"]
        self.sigma2 = torch.tensor(cfg["stddev"] ** 2)        # reference solver.py:53
        self.base_lr, self.sched_steps = cfg["lr"], 0
        mk = lambda P: AdamState(P, cfg["lr"], cfg["beta1"], cfg["beta2"], cfg["weight_decay"])
        self.gen_opt, self.dis_opt = mk(self.gen), mk(self.dis)
        self.gen_copy = self.dis_copy = None
        self.losses = {}

    # ---- helpers -----------------------------------------------------------------
    def copy_nets(self):                                       # reference solver.py:92-94
        self.gen_copy = OrderedDict((k, v.detach().clone()) for k, v in self.gen.items())
        self.dis_copy = OrderedDict((k, v.detach().clone()) for k, v in self.dis.items())

    def _blend(self, img, att, x_real):                        # reference solver.py:160-161 etc.
        if self.use_attention:
            return img * att + x_real * (1 - att)
        return img

    def _sample_style(self, c):
        return self.noise.style_sample(c, self.c_dim, self.stddev)

    def _grads(self, loss, params):
        names = list(params.keys())
        if self.as_written:
            allp = list(self.gen.values()) + list(self.dis.values())
            for p in allp:
                p.grad = None
            loss.backward()
            return {k: params[k].grad for k in names}
        gs = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
        return dict(zip(names, gs))

    # ---- D step ------------------------------------------------------------------
    def dis_update(self, x_real, c_src, c_trg, txt, txt_lens, label_src, label_trg, cfg=None, iters=0):
        """reference solver.py:317-353."""
        cfg = cfg or self.cfg
        G, D, g, d = self.gen, self.dis, self.gcfg, self.dcfg
        # as_written=False evaluates the generator without a tape: same numbers, it only skips
        # the generator backward whose result the reference throws away (solver.py:153)
        with torch.set_grad_enabled(self.as_written):
            content, mus, _ = gen_encode(G, x_real, g, self.noise)
            style_real = torch.cat(mus, 1)
            style1 = self._sample_style(c_trg)
            tmu, _ = gen_encode_txt(G, style_real, txt, txt_lens, g, self.noise)
            style_txt = torch.cat(tmu, 1)
            x_fake = self._blend(*gen_decode(G, content, style_txt, g), x_real)
            x_fake1 = self._blend(*gen_decode(G, content, style1, g), x_real)
        loss = calc_dis_loss(D, x_fake, x_real, label_src, cfg["gan_w"], cfg["cls_w"], d) + \
            calc_dis_loss(D, x_fake1, x_real, label_src, cfg["gan_w"], cfg["cls_w"], d)
        # gradient / R1 penalties on the first scale's src map (reference solver.py:337-350, :291-315): a gradient w.r.t. the INPUT,
        # differentiated again w.r.t. D's weights by the backward below.  Both are added in place to the tensor that loss_dis and
        # loss_dis_all name in the reference, so both scalars read the penalised value.
        one_scale = dict(d, num_scales=1)
        if cfg.get("gp_w", 0.0) > 0.0:
            alpha = self.noise.rand((x_real.shape[0], 1, 1, 1))
            x_hat = (alpha * x_real.detach() + (1 - alpha) * x_fake.detach()).requires_grad_(True)
            y = dis_forward(D, x_hat, one_scale)[0][0]
            dydx = torch.autograd.grad(y, x_hat, grad_outputs=torch.ones_like(y), retain_graph=True, create_graph=True)[0]
            norm = torch.sqrt(torch.sum(dydx.reshape(dydx.shape[0], -1) ** 2, dim=1))
            loss_gp = torch.mean((norm - 1) ** 2) * cfg["gp_w"]
            self.losses["loss_gp"] = float(loss_gp.detach())
            loss = loss + loss_gp
        if cfg.get("use_r1", False) and (iters + 1) % 16 == 0:                       # d_reg_every = 16 (reference solver.py:54)
            x_r = x_real.detach().clone().requires_grad_(True)
            y = dis_forward(D, x_r, one_scale)[0][0]
            dydx = torch.autograd.grad(y, x_r, grad_outputs=torch.ones_like(y), create_graph=True)[0]
            sq = torch.sum(dydx.reshape(dydx.shape[0], -1) ** 2, dim=1)
            loss_r1 = torch.mean(sq ** 2) * 10. / 2                                  # (the square of the squared norm: as written, :313-314)
            self.losses["loss_r1"] = float(loss_r1.detach())
            loss = loss + loss_r1
        self.losses["loss_dis"] = self.losses["loss_dis_all"] = float(loss.detach())
        grads = self._grads(loss, D)
        self.last_dis_grads = grads
        self.dis_opt.lr = self._lr()
        self.dis_opt.step(D, grads)

    # ---- G step ------------------------------------------------------------------
    def gen_update(self, x_real, c_src, c_trg, txt, txt_lens, label_src, label_trg, cfg=None, iters=0):
        """reference solver.py:151-240."""
        cfg = cfg or self.cfg
        G, D, g, d = self.gen, self.dis, self.gcfg, self.dcfg
        L = {}
        content_real, s_real, lv_real = gen_encode(G, x_real, g, self.noise)
        s_real_cat = torch.cat(s_real, 1)
        x_rec = self._blend(*gen_decode(G, content_real, s_real_cat, g), x_real)
        content_rec, s_rec, _ = gen_encode(G, x_rec, g, self.noise)
        s_txt, lv_txt = gen_encode_txt(G, s_real_cat, txt, txt_lens, g, self.noise)
        s_txt_cat = torch.cat(s_txt, 1)
        x_fake = self._blend(*gen_decode(G, content_real, s_txt_cat, g), x_real)
        style1 = self._sample_style(c_trg)
        img1, att1 = gen_decode(G, content_real, style1, g)
        style2 = self._sample_style(c_trg)
        img2, att2 = gen_decode(G, content_real, style2, g)
        x_fake1 = self._blend(img1, att1, x_real)
        x_fake2 = self._blend(img2, att2, x_real)
        L["loss_ds"] = l1_mean(x_fake1, x_fake2.detach())
        content_rand, s_rand, _ = gen_encode(G, x_fake1, g, self.noise)
        self.init_ds_w = max(self.init_ds_w - 1 / 1e5, 0.0)
        content_fake, s_fake, _ = gen_encode(G, x_fake, g, self.noise)
        if cfg["recon_x_cyc_w"] > 0:
            x_cyc = self._blend(*gen_decode(G, content_fake, s_real_cat, g), x_real)
        L["loss_gen_recon_x"] = l1_mean(x_rec, x_real)
        L["loss_gen_recon_c_real"] = l1_mean(content_rec, content_real)
        L["loss_gen_recon_c_fake"] = l1_mean(content_fake, content_real)
        L["loss_gen_recon_c_rand"] = l1_mean(content_rand, content_real)
        L["loss_gen_recon_s_real"] = l1_mean(torch.cat(s_rec, 1), s_real_cat)
        L["loss_gen_recon_s_fake"] = l1_mean(torch.cat(s_fake, 1), s_txt_cat)
        L["loss_gen_recon_s_rand"] = l1_mean(torch.cat(s_rand, 1), style1)
        L["loss_gen_cycrecon_x"] = l1_mean(x_cyc, x_real) if cfg["recon_x_cyc_w"] > 0 else 0.0
        L["loss_gen_adv"] = calc_gen_loss(D, x_fake, label_trg, cfg["gan_w"], cfg["cls_w"], d) + \
            calc_gen_loss(D, x_fake1, label_trg, cfg["gan_w"], cfg["cls_w"], d)
        if cfg["dist_mode"] == "kls":
            L["loss_kl_x"] = gmm_kl_sp(s_real, lv_real, c_src, self.sigma2)
            L["loss_kl_trg"] = gmm_kl_sp(s_txt, lv_txt, c_trg, self.sigma2)
        else:
            L["loss_kl_x"] = gmm_em_sp(s_real, c_src)
            L["loss_kl_trg"] = gmm_em_sp(s_txt, c_trg)
        L["loss_gen_vgg"] = 0.0
        if cfg["recon_x_cyc_w"] > 0 and cfg["vgg_w"] > 0:       # reference solver.py:221-223
            L["loss_gen_vgg"] = vgg_loss(self.vgg, x_real, x_cyc)
        total = L["loss_gen_adv"] + \
            cfg["recon_x_w"] * L["loss_gen_recon_x"] + \
            cfg["recon_c_w"] * L["loss_gen_recon_c_real"] + \
            cfg["recon_c_w"] * L["loss_gen_recon_c_fake"] + \
            cfg["recon_c_w"] * L["loss_gen_recon_c_rand"] + \
            cfg["recon_s_w"] * L["loss_gen_recon_s_real"] + \
            cfg["recon_s_w"] * L["loss_gen_recon_s_fake"] + \
            cfg["recon_s_w"] * L["loss_gen_recon_s_rand"] + \
            cfg["recon_x_cyc_w"] * L["loss_gen_cycrecon_x"] + \
            cfg["kl_w"] * L["loss_kl_x"] + \
            cfg["kl_w"] * L["loss_kl_trg"] + \
            cfg["vgg_w"] * L["loss_gen_vgg"] - \
            self.init_ds_w * L["loss_ds"]
        L["loss_gen_total"] = total
        for k, v in L.items():
            self.losses[k] = float(v.detach()) if torch.is_tensor(v) else float(v)
        grads = self._grads(total, G)
        self.last_gen_grads = grads
        self.gen_opt.lr = self._lr()
        self.gen_opt.step(G, grads)

    # ---- per-iteration bookkeeping -------------------------------------------------
    def _lr(self):
        """StepLR stepped once per iteration (reference solver.py:104-107, utils.py:220-224)."""
        return self.base_lr * (self.cfg["gamma"] ** (self.sched_steps // self.cfg["step_size"]))

    @torch.no_grad()
    def smooth_moving(self, beta=0.999):
        """reference utils.py:52-54: copy <- lerp(param, copy, beta) over parameters only."""
        for P, C in ((self.gen, self.gen_copy), (self.dis, self.dis_copy)):
            for k in P:
                C[k] = torch.lerp(P[k].detach(), C[k], beta)

    def update_learning_rate(self):
        self.sched_steps += 1

    def update_attention_status(self, iters):
        """reference solver.py:109-111."""
        if self.att_status:
            self.use_attention = iters >= 10000

    def iteration(self, batch, it):
        """One pass of the reference training loop body (reference train.py:102-111)."""
        a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"],
             batch["label_src"], batch["label_trg"], self.cfg, it)
        self.dis_update(*a)
        self.gen_update(*a)
        self.smooth_moving()
        self.update_learning_rate()
        self.update_attention_status(it)
