"""MI355X-native building blocks and discriminator behind the reference's ``networks.networks`` API.

Same class names, constructor signatures, method names and ``state_dict`` keys as
reference networks/networks.py (so ``from networks.networks import MsImageDis`` and reference
checkpoints keep working), but every forward/backward runs on the hand-written gfx950
kernels of libdwcgan_hip.so (``hipdwc.ops``).  ``nn.Conv2d`` / ``nn.Linear`` objects are kept
only as parameter containers: that preserves the key names *and* makes construction consume
the random stream exactly like the reference, so a seeded build has identical initial weights.

Internally activations are channels-last (NHWC) and the 3-channel images travel as NHWC4.
Reflect padding is never materialised (it is an index rule inside the conv kernels), the
activation is fused into the conv epilogue or the norm's apply pass, and ResBlock's
residual add is fused into its second norm.
"""
import torch
import torch.nn.functional as F
from torch import nn

from hipdwc import ops

_CONV_ACTS = ("relu", "lrelu", "tanh", "sigmoid", "none")


# --------------------------------------------------------------------------------------
# normalisation layers
# --------------------------------------------------------------------------------------
class AdaptiveInstanceNorm2d(nn.Module):
    """Instance norm whose per-sample scale (``weight``) and shift (``bias``) are assigned from
    outside before each call (reference networks.py:693-722).  The running buffers exist only
    for checkpoint compatibility; like the reference's, they never influence the output."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = None
        self.bias = None
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x, relu=False, residual=None, token=None):
        if self.weight is None or self.bias is None:
            raise AssertionError("Please assign weight and bias before calling AdaIN!")
        return ops.instance_norm(x, self.weight, self.bias, residual=residual, relu=relu, eps=self.eps, token=token)

    def __repr__(self):
        return "%s(%d)" % (type(self).__name__, self.num_features)


class LayerNorm(nn.Module):
    """MUNIT's LayerNorm: per-sample statistics over C*H*W with the unbiased std and eps added
    to the std; gamma ~ U(0,1), beta = 0 (reference networks.py:725-752)."""

    def __init__(self, num_features, eps=1e-5, affine=True):
        super().__init__()
        self.num_features, self.affine, self.eps = num_features, affine, eps
        if affine:
            self.gamma = nn.Parameter(torch.Tensor(num_features).uniform_())
            self.beta = nn.Parameter(torch.zeros(num_features))

    def forward(self, x, relu=False):
        if not self.affine:
            one = torch.ones(self.num_features, device=x.device)
            return ops.layer_norm_munit(x, one, torch.zeros_like(one), relu=relu, eps=self.eps)
        return ops.layer_norm_munit(x, self.gamma, self.beta, relu=relu, eps=self.eps)


class _PlainInstanceNorm(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(affine=False) as a marker/parameter-less container (reference networks.py:545)."""

    def forward(self, x, relu=False, residual=None, token=None):
        return ops.instance_norm(x, None, None, residual=residual, relu=relu, eps=self.eps, token=token)


# --------------------------------------------------------------------------------------
# basic blocks
# --------------------------------------------------------------------------------------
class Conv2dBlock(nn.Module):
    """pad -> conv -> norm -> activation (reference networks.py:524-585) as at most two fused passes.

    What the shipped configurations use -- reflect padding, norm in {none, in, ln, adain}, activation in {relu, lrelu, tanh, sigmoid,
    none} -- is all HIP: the padding rule inside the convolution's gather, bias / activation in its epilogue, ReLU and the residual
    add in the norm's apply pass.  The rest of the reference's signature is reachable too (r05), through stock PyTorch-ROCm DEVICE
    ops around the HIP convolution: ``pad_type`` zero / replicate at stride 1 (``F.pad``, then the convolution without padding), ``norm='bn'``
    (``nn.BatchNorm2d``), ``activation`` prelu / selu (``nn.PReLU`` / ``F.selu``), and any norm followed by an activation other than
    ReLU (the norm unfused, then the activation).  ``norm='sn'`` (the reference's SpectralNorm wrapper) is not built."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="relu",
                 pad_type="zero"):
        super().__init__()
        if pad_type not in ("reflect", "replicate", "zero"):
            raise AssertionError("Unsupported padding type: {}".format(pad_type))
        if pad_type != "reflect" and padding > 0 and stride != 1:
            raise NotImplementedError("pad_type=%r with stride %d: the strided data gradient of the HIP kernels is built for the reflect rule "
                                      "(4x4, stride 2, pad 1); zero / replicate padding is reachable at stride 1" % (pad_type, stride))
        if activation not in _CONV_ACTS and activation not in ("prelu", "selu"):
            raise AssertionError("Unsupported activation: {}".format(activation))
        self.use_bias = True
        self.stride, self.padding = stride, padding
        self.pad_type = pad_type
        self.norm_kind, self.act_kind = norm, activation
        # norm (then activation) is created before the conv, as in the reference: LayerNorm draws its gamma then
        if norm == "in":
            self.norm = _PlainInstanceNorm(output_dim)
        elif norm == "ln":
            self.norm = LayerNorm(output_dim)
        elif norm == "adain":
            self.norm = AdaptiveInstanceNorm2d(output_dim)
        elif norm == "bn":
            self.norm = nn.BatchNorm2d(output_dim)
        elif norm == "none":
            self.norm = None
        elif norm == "sn":
            raise NotImplementedError("norm='sn' (the reference's SpectralNorm wrapper, networks.py:755-800) is not built: no shipped "
                                      "configuration uses it")
        else:
            raise AssertionError("Unsupported normalization: {}".format(norm))
        if activation == "prelu":
            self.activation = nn.PReLU()
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, bias=self.use_bias)  # parameter container

    def _torch_act(self, y):
        if self.act_kind == "prelu":
            return self.activation(y)
        if self.act_kind == "selu":
            return torch.nn.functional.selu(y)
        return {"lrelu": lambda t: torch.nn.functional.leaky_relu(t, 0.1), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[self.act_kind](y)

    def forward(self, x, residual=None, conv_token=None, res_token=None):
        """``conv_token`` / ``res_token`` (hipdwc.ops.ResGradToken, both optional): this block's convolution opens / this block's
        norm closes a residual block whose identity-branch gradient is added in the convolution's data-gradient epilogue."""
        if x.shape[1] < 4:
            x = ops.pack_image(x)
        pad = self.padding
        if self.pad_type != "reflect" and pad > 0:        # zero / replicate: a device pad in front of the unpadded HIP convolution
            x = torch.nn.functional.pad(x, (pad,) * 4, mode="constant" if self.pad_type == "zero" else "replicate")
            pad = 0
        fused_act = self.act_kind in _CONV_ACTS
        if self.norm is None:
            y = ops.conv2d(x, self.conv.weight, self.conv.bias, self.stride, pad, self.act_kind if fused_act else "none", token=conv_token)
            if not fused_act:
                y = self._torch_act(y)
            return y if residual is None else y + residual
        if self.norm_kind == "bn" or self.act_kind not in ("relu", "none"):
            # off the shipped configurations: the norm on its own (HIP for in / ln / adain, torch for bn), then the activation
            y = ops.conv2d(x, self.conv.weight, self.conv.bias, self.stride, pad, "none", token=conv_token)
            y = self.norm(y) if self.norm_kind in ("bn", "ln") else self.norm(y, relu=False)
            y = torch.relu(y) if self.act_kind == "relu" else (y if self.act_kind == "none" else self._torch_act(y))
            return y if residual is None else y + residual
        y = ops.conv2d(x, self.conv.weight, self.conv.bias, self.stride, pad, "none",
                       bias_grad=self.norm_kind == "ln", token=conv_token)   # IN / AdaIN subtract the per-(n,c) mean: d/d bias == 0
        relu = self.act_kind == "relu"
        if self.norm_kind == "ln":
            y = self.norm(y, relu=relu)
            return y if residual is None else y + residual
        return self.norm(y, relu=relu, residual=residual, token=res_token)


class ResBlock(nn.Module):
    """x + conv-norm(conv-norm-act(x)) (reference networks.py:509-522); the add rides on the second norm."""

    def __init__(self, dim, norm="in", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(
            Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation=activation, pad_type=pad_type),
            Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation="none", pad_type=pad_type))

    def forward(self, x, groups=1):
        # the gradient of the identity branch rides on the data gradient of the first convolution (ops.ResGradToken); only when
        # the closing norm takes the residual itself (IN / AdaIN) -- otherwise autograd sums the two gradients as usual
        fused = self.model[1].norm_kind in ("in", "adain")
        if groups > 1:
            return self._forward_groups(x, groups)
        token = ops.res_token(x) if fused else None
        return self.model[1](self.model[0](x, conv_token=token), residual=x, res_token=token)

    def _forward_groups(self, x, groups):
        """The block applied to ``groups`` copies of the batch x that differ only in their AdaIN parameters (the solver decodes one
        content code with several styles in one pass: reference solver.py:171-190 runs gen.decode three times on c_real): the first
        convolution sees the same input in every group, so it runs ONCE at batch B and its output is repeated in front of the
        AdaIN -- a third of its forward, data-gradient and weight-gradient work at three groups.  Same value as the block on
        torch.cat([x] * groups) up to fp32 summation order (the three groups' gradients meet before the convolution instead of
        inside its batch)."""
        c0, c1 = self.model[0], self.model[1]
        if c0.norm_kind != "adain" or c0.act_kind != "relu" or c0.pad_type != "reflect" or c1.norm_kind != "adain":
            return self.forward(torch.cat([x] * groups), 1)
        y = ops.conv2d(x, c0.conv.weight, c0.conv.bias, c0.stride, c0.padding, "none", bias_grad=False)
        h = c0.norm(torch.cat([y] * groups), relu=True)
        return c1(h, residual=torch.cat([x] * groups))


class ResBlocks(nn.Module):
    def __init__(self, num_blocks, dim, norm="in", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(*[ResBlock(dim, norm=norm, activation=activation, pad_type=pad_type)
                                     for _ in range(num_blocks)])

    def forward(self, x, groups=1):
        if groups > 1 and len(self.model) > 0:        # (see ResBlock._forward_groups: x is ONE copy of the batch)
            x = self.model[0](x, groups=groups)
            for blk in list(self.model)[1:]:
                x = blk(x)
            return x
        return self.model(x)


class LinearBlock(nn.Module):
    """Linear (+ReLU) (reference networks.py:587-634) on the 1x1-conv kernel."""

    def __init__(self, input_dim, output_dim, norm="none", activation="relu"):
        super().__init__()
        if norm != "none":
            raise NotImplementedError("LinearBlock norm %r is not on the HIP path" % norm)
        if activation not in ("relu", "none"):
            raise NotImplementedError("LinearBlock activation %r is not on the HIP path" % activation)
        self.act_kind = activation
        self.norm = None
        self.fc = nn.Linear(input_dim, output_dim, bias=True)

    def forward(self, x):
        return ops.linear(x, self.fc.weight, self.fc.bias, self.act_kind)


class MLP(nn.Module):
    """style -> AdaIN parameters (reference networks.py:491-503)."""

    def __init__(self, input_dim, output_dim, dim, n_blk, norm="none", activ="relu"):
        super().__init__()
        dims = [input_dim] + [dim] * (n_blk - 1) + [output_dim]
        blocks = [LinearBlock(dims[i], dims[i + 1], norm=norm if i < n_blk - 1 else "none",
                              activation=activ if i < n_blk - 1 else "none") for i in range(n_blk)]
        self.model = nn.Sequential(*blocks)

    def forward(self, x):
        return self.model(x.reshape(x.size(0), -1))


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear') stand-in without parameters."""

    def forward(self, x):
        return ops.upsample2x(x)


# --------------------------------------------------------------------------------------
# encoders / decoder shared with networks_v2
# --------------------------------------------------------------------------------------
class ContentEncoder(nn.Module):
    """7x7 stem, n_downsample stride-2 convs (channel cap 256), n_res IN ResBlocks
    (reference networks.py:428-446)."""

    def __init__(self, n_downsample, n_res, input_dim, dim, norm, activ, pad_type):
        super().__init__()
        layers = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type=pad_type)]
        for _ in range(n_downsample):
            nxt = min(dim * 2, 256)
            layers.append(Conv2dBlock(dim, nxt, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type))
            dim = nxt
        layers.append(ResBlocks(n_res, dim, norm=norm, activation=activ, pad_type=pad_type))
        self.model = nn.Sequential(*layers)
        self.output_dim = dim

    def forward(self, x):
        return self.model(ops.pack_image(x) if x.shape[1] < 4 else x)


class Decoder(nn.Module):
    """AdaIN ResBlocks -> n_upsample x [bilinear x2, 5x5 conv, LN, act] -> tanh image head and
    sigmoid attention head (reference networks.py:449-475 / networks_v2.py:144-169).

    The two 7x7 heads read the same feature map, so they run as ONE 4-channel convolution
    (planes 0-2 tanh, plane 3 sigmoid); ``forward`` hands back the usual (image, attention)
    pair as channel views of that NHWC4 buffer, ``forward_nhwc4`` the buffer itself."""

    def __init__(self, n_upsample, n_res, dim, output_dim, res_norm="adain", activ="relu", pad_type="zero",
                 use_attention=False):
        super().__init__()
        self.use_attention = use_attention
        self.output_dim = output_dim
        layers = [ResBlocks(n_res, dim, res_norm, activ, pad_type=pad_type)]
        for _ in range(n_upsample):
            layers += [Upsample2x(), Conv2dBlock(dim, dim // 2, 5, 1, 2, norm="ln", activation=activ, pad_type=pad_type)]
            dim //= 2
        self.model = nn.Sequential(*layers)
        self.image_content = Conv2dBlock(dim, output_dim, 7, 1, 3, norm="none", activation="tanh", pad_type=pad_type)
        self.image_attention = Conv2dBlock(dim, 1, 7, 1, 3, norm="none", activation="sigmoid", pad_type=pad_type)

    def forward_nhwc4(self, x, attention_used=True, groups=1):
        """``attention_used=False``: the caller will not read plane 3, so the attention head is taken
        off the tape — its parameters then get NO gradient (not a zero one), exactly like the
        reference, whose optimiser skips gradient-less parameters (no weight decay / momentum).
        ``groups`` > 1: x is ONE copy of a batch that is decoded with ``groups`` sets of AdaIN parameters (assigned for
        groups * B samples); the result has groups * B samples (ResBlock._forward_groups)."""
        if self.output_dim != 3:
            raise NotImplementedError("fused heads assume a 3-channel image")
        wc, bc = self.image_content.conv.weight, self.image_content.conv.bias
        wa, ba = self.image_attention.conv.weight, self.image_attention.conv.bias
        owner = (wc, wa)                 # the prepared-weight cache lives on the two parameters, not on the concatenation
        if not attention_used:
            wa, ba = wa.detach(), ba.detach()
        with ops.scope("decode"):        # label for the profiler's decode-stack roofline figure
            if groups > 1 and isinstance(self.model[0], ResBlocks):
                feat = self.model[0](x, groups=groups)
                for layer in list(self.model)[1:]:
                    feat = layer(feat)
            else:
                feat = self.model(x if groups == 1 else torch.cat([x] * groups))
            ws, bs = [wc, wa], [bc, ba]
            extra = ops.image_planes(feat.dtype) - 4         # bf16 images are NHWC8: four zero planes
            if extra:
                ws.append(wa.new_zeros((extra,) + tuple(wa.shape[1:])))
                bs.append(ba.new_zeros(extra))
            return ops.conv2d_heads(feat, torch.cat(ws, 0), torch.cat(bs, 0), owner=owner)

    def forward(self, x):
        heads = self.forward_nhwc4(x)
        return heads[:, :3], (heads[:, 3:4] if self.use_attention else None)


# --------------------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------------------
class MsImageDis(nn.Module):
    """Multi-scale discriminator (reference networks.py:43-170): per scale n_layer x
    [reflect-pad 1, 4x4 stride-2 conv, LeakyReLU(0.1)], a 1x1 'src' head and a full-extent
    'cls' head; the next scale sees the 2x2-mean image."""

    def __init__(self, input_dim, params, device=None):
        super().__init__()
        self.n_layer, self.gan_type, self.dim = params["n_layer"], params["gan_type"], params["dim"]
        self.norm, self.activ, self.num_scales = params["norm"], params["activ"], params["num_scales"]
        self.pad_type, self.num_cls, self.input_dim = params["pad_type"], params["num_cls"], input_dim
        self.image_size, self.dataset = params["image_size"], params["dataset"]
        self.device = device if device is not None else torch.device("cpu")
        self.cnns_feat, self.cnns_src, self.cnns_cls = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for s in range(self.num_scales):
            feat, src, cls = self._make_net(self.image_size // (2 ** s))
            self.cnns_feat.append(feat)
            self.cnns_src.append(src)
            self.cnns_cls.append(cls)

    def _make_net(self, im_size):
        dim, chain, prev = self.dim, [], self.input_dim
        for l in range(self.n_layer):
            chain.append(Conv2dBlock(prev, dim, 4, 2, 1, norm="none" if l == 0 else self.norm, activation=self.activ,
                                     pad_type=self.pad_type))
            prev, dim = dim, min(dim * 2, 512)
        src = nn.Conv2d(prev, 1, 1, 1, 0)
        cls = nn.Conv2d(prev, self.num_cls, kernel_size=im_size // (2 ** self.n_layer), stride=1, padding=0, bias=False)
        return nn.Sequential(*chain), src, cls

    def forward(self, x, use_multiscales=True):
        x = ops.pack_image(x)
        outputs = []
        for s in range(self.num_scales):
            h = self.cnns_feat[s](x)
            src = ops.conv2d(h, self.cnns_src[s].weight, self.cnns_src[s].bias, 1, 0).float()   # losses are fp32 reductions
            cls = ops.conv2d(h, self.cnns_cls[s].weight, None, 1, 0).float()
            outputs.append([src, cls.reshape(cls.size(0), -1)])
            if not use_multiscales:
                break
            if s + 1 < self.num_scales:
                x = ops.downsample_half(x)
        return outputs

    def forward_src_scale0_torch(self, x):
        """The first scale's 'src' map -- ``self(x, False)[0][0]`` -- on stock torch device ops (F.pad / F.conv2d / activation), for the
        two callers that differentiate it TWICE: Solver.gradient_penalty / r1_penalty (reference solver.py:291-315,338-350; off in the
        shipped configuration).  The HIP autograd Functions are once-differentiable; these penalties are an O(B) side branch of the D
        step, so they take torch's own double backward on the same parameters instead.  x: [B, 3, H, W] fp32."""
        if self.norm != "none" or self.pad_type not in ("reflect", "zero", "replicate"):
            raise NotImplementedError("gradient penalties: discriminator norm %r / pad %r" % (self.norm, self.pad_type))
        mode = {"reflect": "reflect", "zero": "constant", "replicate": "replicate"}[self.pad_type]
        h = x.float()
        for blk in self.cnns_feat[0]:
            p = blk.padding
            h = F.conv2d(F.pad(h, (p, p, p, p), mode=mode) if p else h, blk.conv.weight, blk.conv.bias, stride=blk.stride)
            if blk.act_kind == "relu":
                h = torch.relu(h)
            elif blk.act_kind == "lrelu":
                h = F.leaky_relu(h, 0.1)
            elif blk.act_kind != "none":
                h = blk._torch_act(h)
        return F.conv2d(h, self.cnns_src[0].weight, self.cnns_src[0].bias)

    def _classification_loss(self, logit, target, dataset="CelebA"):
        if dataset in ("CelebA", "CUB200"):
            return F.binary_cross_entropy_with_logits(logit, target, reduction="mean")
        return F.cross_entropy(logit, target)

    def _gan_term(self, out, target_is_real):
        if self.gan_type == "lsgan":
            return torch.mean((out - (1.0 if target_is_real else 0.0)) ** 2)
        if self.gan_type == "nsgan":
            tgt = torch.ones_like(out) if target_is_real else torch.zeros_like(out)
            return F.binary_cross_entropy(torch.sigmoid(out), tgt)
        if self.gan_type == "wgan":
            return -torch.mean(out) if target_is_real else torch.mean(out)
        raise AssertionError("Unsupported GAN type: {}".format(self.gan_type))

    @staticmethod
    def split_outputs(outputs, sizes):
        """Outputs of ONE forward over a concatenated batch -> per-segment output lists (the
        discriminator has no cross-sample op, so this equals separate forwards)."""
        parts = [[] for _ in sizes]
        for src, cls in outputs:
            for p, s, c in zip(parts, torch.split(src, sizes), torch.split(cls, sizes)):
                p.append([s, c])
        return parts

    def adv_loss(self, outputs, B, labels, targets, w_src, w_cls):
        """Adversarial objective of a BATCHED pass (segments of B samples: [x_fake | x_fake1 | x_real] in the D step,
        [x_fake | x_fake1] in the G step), one tail launch per scale (hipdwc.ops.adv_tail) instead of the reference's term-by-term
        LSGAN / BCE algebra (networks.py:116-170):  sum over scales and segments s of
        w_src[s] * mean((src_s - targets[s])^2) + w_cls[s] * BCEwithLogits(cls_s, labels).  LSGAN + CelebA/CUB200 only (the
        shipped configuration); anything else takes the term-by-term methods below."""
        if self.gan_type != "lsgan" or self.dataset not in ("CelebA", "CUB200"):
            raise NotImplementedError("adv_loss covers gan_type lsgan with attribute BCE; use dis_loss_terms / gen_loss_terms")
        loss = None
        for src, cls in outputs:
            term = ops.adv_tail(src, cls, labels, B, targets, w_src, w_cls)
            loss = term if loss is None else loss + term
        return loss

    def dis_loss_terms(self, outs_fake, outs_real, real_cls, weight_gan=1.0, weight_cls=1.0):
        loss = 0.0
        for (src_f, _), (src_r, cls_r) in zip(outs_fake, outs_real):
            loss = loss + (self._gan_term(src_f, False) + self._gan_term(src_r, True)) * weight_gan
            loss = loss + self._classification_loss(cls_r, real_cls, self.dataset) * weight_cls
        return loss

    def gen_loss_terms(self, outs_fake, target_cls, weight_gan=1.0, weight_cls=1.0):
        loss = 0
        for src_f, cls_f in outs_fake:
            loss = loss + self._gan_term(src_f, True) * weight_gan
            loss = loss + self._classification_loss(cls_f, target_cls, self.dataset) * weight_cls
        return loss

    def calc_dis_loss(self, input_fake, input_real, fake_cls, real_cls, weight_gan=1.0, weight_cls=1.0):
        """D objective (reference networks.py:116-146)."""
        return self.dis_loss_terms(self.forward(input_fake), self.forward(input_real), real_cls, weight_gan, weight_cls)

    def calc_gen_loss(self, input_fake, target_cls, weight_gan=1.0, weight_cls=1.0):
        """G-side adversarial objective (reference networks.py:148-170)."""
        return self.gen_loss_terms(self.forward(input_fake), target_cls, weight_gan, weight_cls)


class Vgg16(nn.Module):
    """The frozen VGG16 trunk of the perceptual loss (reference networks.py:639-688): thirteen zero-padded 3x3
    convolutions + ReLU, 2x2 max pooling after blocks 1-3, output relu5_3.  Same attribute names and state_dict keys
    (conv1_1.weight ... conv5_3.bias); the nn.Conv2d objects are parameter containers."""

    CFG = ((1, 3, 64, 2), (2, 64, 128, 2), (3, 128, 256, 3), (4, 256, 512, 3), (5, 512, 512, 3))

    def __init__(self):
        super().__init__()
        for blk, cin, cout, n in self.CFG:
            for i in range(n):
                setattr(self, "conv%d_%d" % (blk, i + 1), nn.Conv2d(cin if i == 0 else cout, cout, kernel_size=3, stride=1, padding=1))

    def forward(self, X):
        # Under bf16 activations conv1_1 still runs in fp32: its input is the PREPROCESSED image (values in [-124, 152]), which
        # bf16 would quantise to steps of 0.5-1.0 -- coarser than the bf16 image it came from.  Its 64-channel output is cast to
        # bf16 and the other twelve convolutions, the poolings and the loss's instance norms run on the bf16 kernels.
        half = ops.PRECISION == "bf16"
        if X.shape[1] < 4:
            h = ops._Pack4.apply(X.float(), False) if half else ops.pack_image(X)
        else:
            h = X
        first = True
        for blk, _, _, n in self.CFG:
            for i in range(n):
                conv = getattr(self, "conv%d_%d" % (blk, i + 1))
                h = ops.conv2d_zeropad(h, conv.weight, conv.bias, 1, "relu")
                if first and half:
                    h = h.to(ops.BF16)
                first = False
            if blk <= 3:
                h = ops.max_pool2(h)
        return h
