"""Host-side glue of the training step: weight init, LR schedule, EMA, random draws.

These follow reference utils.py:52-54 (moving_average), :220-231 (get_scheduler),
:234-254 (weights_init) and the four places where the reference draws random numbers
(tools.py:65-70, networks_v2.py:119, :222, :201/:236).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.optim import lr_scheduler


# --------------------------------------------------------------------------------------
# random draws
# --------------------------------------------------------------------------------------
_STYLE_NORMAL_CHECKED = int(__import__("os").environ.get("DWC_STYLE_NORMAL", "0"))


class DeviceNoise:
    """Default: draw on the tensor's own device, like the reference does on its GPU."""
    align_stream = False     # draws whose result is unused may be skipped

    def __init__(self):
        self._ones = {}

    def dropout(self, x, p, training=True):
        return F.dropout(x, p=p, training=training)

    def dropout_mask(self, shape, p, device):
        """Scaled keep-mask (0 or 1/(1-p)) drawn now, to be multiplied in later."""
        key = (tuple(shape), str(device))
        ones = self._ones.get(key)
        if ones is None:               # (a constant input of the draw: filled once per shape, not once per call)
            if len(self._ones) > 64:
                self._ones.clear()
            ones = self._ones[key] = torch.ones(tuple(shape), dtype=torch.float32, device=device)
        return F.dropout(ones, p=p, training=True)

    def rand(self, shape, device):
        """U[0, 1) draws (the interpolation weights of the gradient penalty, reference solver.py:339)."""
        return torch.rand(tuple(shape), device=device)

    def style_sample(self, mu, c_dim, stddev):
        shape = (1, c_dim) + tuple(mu.shape)
        # (r06: torch.normal(mean tensor, std TENSOR) checks `std.min() >= 0` on the host -- a device synchronisation in the middle of
        # the step, three per iteration, each of which drains the launch queue (profiles/r05_torch_kernel_sites_c1.txt: 6
        # _local_scalar_dense).  Same distribution, same generator, no check: standard normal draws scaled by the scalar stddev.)
        if _STYLE_NORMAL_CHECKED:       # (A/B switch DWC_STYLE_NORMAL=1: the synchronising form of rounds 1-5)
            draw = torch.normal(mu.expand(shape), torch.full_like(mu, stddev).expand(shape))
            return draw.permute(0, 2, 3, 1).reshape(mu.shape[0], -1)
        # drawn in the layout it is used in ([B, A, c_dim]: the entries are i.i.d., so WHICH draw lands where is immaterial on the
        # device generator): randn + one fused multiply-add, no transposing copy
        eps = torch.randn(tuple(mu.shape) + (c_dim,), dtype=mu.dtype, device=mu.device)
        return torch.add(mu.unsqueeze(-1), eps, alpha=float(stddev)).reshape(mu.shape[0], -1)


class HostNoise:
    """Parity mode: every draw is made on torch's global CPU generator with the calls, shapes
    and order the reference makes when it runs on CPU, then moved to the device.  With the
    same seed the HIP run then sees the same dropout masks and style samples as a CPU run
    (reference or oracle)."""
    align_stream = True      # reproduce even the draws whose result the reference throws away

    def dropout(self, x, p, training=True):
        if not training or p == 0:
            return x
        return x * self.dropout_mask(x.shape, p, x.device)

    def dropout_mask(self, shape, p, device):
        """Same stream consumption as F.dropout on a CPU tensor of that shape (one bernoulli_ call)."""
        mask = F.dropout(torch.ones(tuple(shape), dtype=torch.float32), p=p, training=True)
        return mask.to(device, non_blocking=True)

    def style_sample(self, mu, c_dim, stddev):
        m = mu.detach().to("cpu", torch.float32)
        shape = (1, c_dim) + tuple(m.shape)
        draw = torch.normal(m.expand(shape), torch.full_like(m, stddev).expand(shape))
        return draw.permute(0, 2, 3, 1).reshape(m.shape[0], -1).to(mu.device)

    def rand(self, shape, device):
        return torch.rand(tuple(shape)).to(device)


_NOISE = DeviceNoise()


def noise():
    return _NOISE


def set_noise(source):
    global _NOISE
    _NOISE = source
    return source


# --------------------------------------------------------------------------------------
# init / schedule / EMA
# --------------------------------------------------------------------------------------
def weights_init(init_type="gaussian"):
    """Initialiser applied with Module.apply: touches modules whose class name starts with
    Conv/Linear and that own a ``weight`` (i.e. nn.Conv2d / nn.Linear), zeroes their bias."""
    fillers = {
        "gaussian": lambda w: nn.init.normal_(w, 0.0, 0.02),
        "xavier": lambda w: nn.init.xavier_normal_(w, gain=math.sqrt(2)),
        "kaiming": lambda w: nn.init.kaiming_normal_(w, a=0, mode="fan_in"),
        "orthogonal": lambda w: nn.init.orthogonal_(w, gain=math.sqrt(2)),
        "default": lambda w: w,
    }
    if init_type not in fillers:
        raise ValueError("Unsupported initialization: %s" % init_type)
    fill = fillers[init_type]

    def visit(m):
        name = type(m).__name__
        if (name.startswith("Conv") or name.startswith("Linear")) and hasattr(m, "weight"):
            fill(m.weight.data)
            touched = [m.weight]
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
                touched.append(m.bias)
            torch.autograd.graph.increment_version(touched)     # .data writes: keep version-keyed caches honest
    return visit


def get_scheduler(optimizer, hp, iterations=-1):
    policy = hp.get("lr_policy", "const")
    if policy == "const":
        return None
    if policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=hp["step_size"], gamma=hp["gamma"], last_epoch=iterations)
    if policy == "cosa":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=hp["step_size"], eta_min=hp["eta_min"],
                                              last_epoch=iterations)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % policy)


@torch.no_grad()
def moving_average(model, model_copy, beta=0.999):
    """copy <- lerp(param, copy, beta) over parameters (buffers are not averaged)."""
    src = [p.data for p in model.parameters()]
    dst = [p.data for p in model_copy.parameters()]
    if src and src[0].is_cuda:
        # dst + (1-beta)*(src-dst) in torch.lerp's high-weight form, one fused multi-tensor call
        torch._foreach_lerp_(dst, src, 1.0 - beta)
        torch.autograd.graph.increment_version(list(model_copy.parameters()))   # .data writes do not bump it
    else:
        for s, d in zip(src, dst):
            d.copy_(torch.lerp(s, d, beta))


def load_vgg16(model_dir):
    """Reference utils.py:180-193 without the download: loads <model_dir>/vgg16.weight (the state_dict the reference
    caches there after converting the Torch7 file).  There is no network on the training boxes this targets, so a
    missing file is an error, not a fetch."""
    import os
    from networks.networks import Vgg16
    path = os.path.join(model_dir, "vgg16.weight")
    if not os.path.exists(path):
        raise FileNotFoundError("%s not found: place the reference's converted VGG16 state_dict there "
                                "(reference utils.py:186-190 produces it), or run with vgg_w: 0" % path)
    vgg = Vgg16()
    vgg.load_state_dict(torch.load(path, map_location="cpu"))
    return vgg


def vgg_preprocess(batch):
    """Reference utils.py:207-217: RGB -> BGR, [-1, 1] -> [0, 255], subtract the ImageNet BGR means."""
    r, g, b = torch.chunk(batch[:, :3].float(), 3, dim=1)       # fp32 from here (internal images may be bf16 NHWC8)
    out = (torch.cat((b, g, r), dim=1) + 1) * 255 * 0.5
    mean = torch.tensor([103.939, 116.779, 123.680], dtype=out.dtype, device=out.device).view(1, 3, 1, 1)
    return out - mean
