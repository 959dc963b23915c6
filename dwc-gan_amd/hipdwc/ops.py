"""torch.autograd wrappers around the HIP kernels (C ABI in include/dwcgan_hip.h).

Tensors keep the reference's logical NCHW shapes but live channels-last in memory
(``torch.channels_last``), which is exactly the NHWC layout the kernels stream.  Every
function here requires device tensors; there is no CPU path (use the oracle for that,
from tests only).
"""
import os
import weakref

import numpy as np
import torch

from . import _lib

ACT = {"none": 0, None: 0, "relu": 1, "lrelu": 2, "tanh": 3, "sigmoid": 4, "heads": 5, "heads8": 6}

# Activation precision of the conv/norm stack.  "fp32" (default; BASELINE configs[1], the parity configuration): fp32 NHWC
# activations, images as NHWC4.  "bf16" (BASELINE configs[2]): `pack_image` produces bf16 NHWC8 images and every op
# downstream keeps bf16 activations / activation gradients on the bf16 MFMA kernels (dwc_bf16_* entry points); master
# weights, weight gradients, biases, norm statistics, AdaIN/LN parameters and all loss reductions stay fp32.
# Ops dispatch on the dtype of the tensor they are given, so fp32 side branches (the style MLP, the text encoder) coexist.
PRECISION = "fp32"
BF16 = torch.bfloat16

# Switches of earlier rounds that no longer select anything: a script or profile recipe that still sets one would
# otherwise run another configuration than it claims, silently.
_REMOVED_SWITCHES = ("DWC_WINOGRAD", "DWC_TXT_STREAM", "DWC_NORM_FUSED_FINAL", "DWC_GEMM_TILE", "DWC_STRIP_BM", "DWC_X3_DUO", "DWC_X3_BN",
                     "DWC_X3_BN32", "DWC_X3_WIDE", "DWC_X3_STAGGER", "DWC_HALO_STAGGER", "DWC_X3_WDUO", "DWC_HALO16", "DWC_HALO_DUO",
                     "DWC_HALO_BN128", "DWC_BF16_TILE", "DWC_BF16_STAGES3", "DWC_NARROW_ROWS16")
_set = [k for k in _REMOVED_SWITCHES if k in os.environ]
if _set:
    import warnings
    warnings.warn("hipdwc: %s no longer select%s anything (removed in round 5, DESIGN.md 10.3) and %s ignored" % (
        ", ".join(_set), "s" if len(_set) == 1 else "", "is" if len(_set) == 1 else "are"), stacklevel=2)
del _set


def set_precision(name):
    """Select "fp32" or "bf16" activations for images packed from now on."""
    global PRECISION
    if name not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    PRECISION = name


def act_dtype():
    return BF16 if PRECISION == "bf16" else torch.float32


def image_planes(dtype=None):
    """Channel planes of an internal image buffer: NHWC4 (fp32) or NHWC8 (bf16; one 16-byte chunk per pixel either way)."""
    return 8 if (act_dtype() if dtype is None else dtype) == BF16 else 4


def _fn(lib, name, t):
    """C-ABI entry point `dwc_<name>` for fp32 tensors, `dwc_bf16_<name>` for bf16 ones."""
    return getattr(lib, ("dwc_bf16_" if t.dtype == BF16 else "dwc_") + name)


# --------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------
def _require_device(t):
    if not t.is_cuda:
        raise RuntimeError("dwc-gan_amd HIP op called with a CPU tensor: the product path has no CPU "
                           "fallback (the CPU oracle lives under oracle/ and is for tests only)")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The raw hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object
    through three layers of device-index helpers: 8.7 us per call, 440 calls per c1 iteration in the forward alone = 3.8 ms of the
    host's 35 ms -- benchmarks/host_profile.py, r06.  The private accessor is the one call underneath it.)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_WS = {}


def workspace(nbytes, device):
    """Scratch arena handed to the kernels (grown on demand, never shrunk): one per (device, stream) -- kernels on one
    stream run in order and may share it, concurrent streams must not."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def cl(x):
    """channels-last contiguous view/copy of a 4-D fp32 or bf16 tensor."""
    if x.dtype not in (torch.float32, BF16):
        raise TypeError("fp32 or bf16 only")
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW-shaped tensor")
    return x.contiguous(memory_format=torch.channels_last)


def empty_cl(b, c, h, w, device, dtype=torch.float32):
    return torch.empty((b, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def _p(t):
    return None if t is None else t.data_ptr()


class KernelTimer:
    """Optional per-launch timing of the conv GEMM kernels with HIP events recorded on the
    launch stream (torch's current stream IS the stream the C ABI launches on).  bench.py
    switches it on for one timed step to obtain the live roofline figure; off by default."""

    def __init__(self):
        self.spans = []   # (kernel tag, algorithmic flops, start event, end event)
        self.details = []  # problem shape of each span (benchmarks/step_breakdown.py)
        self.executed = []  # multiply-add flops the matrix cores actually issued (< algorithmic for Winograd launches)
        self.hbm = {}       # HBM-bound op family -> [calls, algorithmic bytes] (benchmarks/roofline_table.py: GB/s of those kernels)

    def run(self, tag, flops, fn, detail="", exec_flops=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn()
        b.record()
        self.spans.append((tag, flops, a, b))
        self.details.append(detail)
        self.executed.append(flops if exec_flops is None else exec_flops)
        return rc

    def ledger(self):
        """Per launch KIND (first word of the span's detail + filter size, e.g. "fwd-x3/k5"): launches, milliseconds, algorithmic and
        executed flops, and the problem shapes seen -- what bench.py's `roofline` and benchmarks/roofline_table.py are built from."""
        torch.cuda.synchronize()
        out = {}
        for (tag, flops, a, b), ex, det in zip(self.spans, self.executed, self.details):
            words = det.split()
            kind = words[0] if words else tag
            ksz = next((w for w in words[1:] if w[0] == "k" and w[1:].isdigit()), "")
            ent = out.setdefault(kind + ("/" + ksz if ksz else ""), {"launches": 0, "ms": 0.0, "flops": 0.0, "exec_flops": 0.0,
                                                                      "tag": tag.split("/")[-1], "shapes": {}, "shape_ms": {}})
            ms = a.elapsed_time(b)
            ent["launches"] += 1
            ent["ms"] += ms
            ent["flops"] += flops
            ent["exec_flops"] += ex
            ent["shapes"][det] = ent["shapes"].get(det, 0) + 1
            ent["shape_ms"][det] = round(ent["shape_ms"].get(det, 0.0) + ms, 4)
        return out

    def hbm_ledger(self):
        """HBM-bound op families of the instrumented step: calls and algorithmic bytes (see _hbm)."""
        return {k: {"calls": v[0], "bytes": v[1]} for k, v in self.hbm.items()}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for (tag, flops, a, b), ex in zip(self.spans, self.executed):
            ent = out.setdefault(tag, {"launches": 0, "ms": 0.0, "flops": 0.0, "exec_flops": 0.0})
            ent["launches"] += 1
            ent["ms"] += a.elapsed_time(b)
            ent["flops"] += flops
            ent["exec_flops"] += ex
        return out


TIMER = None   # set to a KernelTimer to collect spans


def _hbm(family, nbytes):
    """Bookkeeping for the roofline table: the ALGORITHMIC bytes an HBM-bound op moves (each operand read once, each result written
    once), counted while bench.py's timer is on; no events, no launches."""
    if TIMER is not None:
        e = TIMER.hbm.setdefault(family, [0, 0])
        e[0] += 1
        e[1] += int(nbytes)
SCOPE = ""     # optional label (e.g. "decode") prefixed to the span tags of launches made inside it


class scope:
    """``with ops.scope("decode"): ...`` labels the kernel spans recorded inside (profiling only)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global SCOPE
        self.prev, SCOPE = SCOPE, self.name

    def __exit__(self, *exc):
        global SCOPE
        SCOPE = self.prev


# launch kinds of the fp32 im2col kernels whose inner products run as split products when dwc_x3_gemm_mode says so (r04)
_G3_GEMM = {"fwd", "dgrad", "dgrad-image", "fwd-heads", "dgrad-heads", "fwd-zeropad", "dgrad-zeropad"}
_G3_WGRAD = {"wgrad", "wgrad-heads"}


def _timed(tag, flops, fn, scope_name=None, detail="", exec_flops=None):
    if TIMER is None:
        return fn()
    s = SCOPE if scope_name is None else scope_name
    words = detail.split()
    if PRECISION == "fp32" and words and (words[0] in _G3_GEMM or words[0] in _G3_WGRAD):
        # bookkeeping only: which matrix pipe the span's multiply-adds ran on (kind suffix -g3, split-product family, 6 bf16
        # multiply-adds per executed fp32 one)
        mode = _lib.load().dwc_x3_gemm_mode(-1)
        try:
            ci = int(words[3].split(">")[0])
            k = int(next(w for w in words[4:] if w[0] == "k")[1:])
        except (IndexError, ValueError, StopIteration):
            ci, k = 0, 0
        on = (mode & 2) if words[0] in _G3_WGRAD else ((mode & 1) and k * k * ci >= 128)
        if on:
            detail = words[0] + "-g3 " + " ".join(words[1:])
            exec_flops = 6.0 * (flops if exec_flops is None else exec_flops)
            tag = "wgrad_x3_kernel+reduce" if words[0] in _G3_WGRAD else "conv_halo_x3_kernel"
    return TIMER.run((s + "/" + tag) if s else tag, flops, fn, detail, exec_flops)


def _pad4(n):
    return (n + 3) // 4 * 4


def _padc(n, dtype):
    """Channel count rounded up to one 16-byte chunk's worth of the next multiple the kernels need (4 fp32 / 8 bf16)."""
    return (n + 7) // 8 * 8 if dtype == BF16 else (n + 3) // 4 * 4


# --------------------------------------------------------------------------------------
# weight layouts, cached per parameter version
# --------------------------------------------------------------------------------------
_WCACHE = {}   # id(tensor) -> (weakref to it, {layout key: (version, prepared tensor)})


_X3_BANK_IDX = {}      # (kind, filter shape, pads, device) -> int32 gather table of a split filter bank (see _prepped)


def _x3_bank_recipe(kind, w, cout_pad, cin_pad):
    """fp32 filter (or its element numbers) -> the UNSPLIT bank of csrc/conv_narrow_x3.hip in the kernels' element order.
    stem_steps*: [13 k-steps][64 channels][(h ^ ((co>>3)&1))*8 + 4t + p] with tap = 4j + 2h + t (stem_steps_x3: w is the stem
    filter [64][P<=4][7][7]; stem_steps_dgrad_x3: the heads filter [P<=4][64][7][7], rotated and transposed).
    heads_narrow_x3 / dgrad_image_narrow_x3: the wide bank [32][64][7][14] (8 pixels x 4 planes, copy p = the filter shifted right
    by p taps) as [slab q of 16 channels][tap, 98 padded to 104 with zeros][lane = half*32 + bank row][8 channels 16q + 8*half ..]."""
    if kind in ("stem_steps_x3", "stem_steps_dgrad_x3"):
        wf = w.float()
        if kind == "stem_steps_dgrad_x3":
            wf = wf.flip(2, 3).permute(1, 0, 2, 3)                      # [64][P][7][7]
        co, pl, kh, kw = wf.shape
        assert co == 64 and pl <= 4 and kh == 7 and kw == 7
        bank = torch.zeros((64, 52, 4), dtype=torch.float32, device=w.device)      # [co][tap (49 + 3 zero)][plane]
        bank[:, :49, :pl] = wf.permute(0, 2, 3, 1).reshape(64, 49, pl)
        steps = bank.view(64, 13, 2, 8).permute(1, 0, 2, 3).contiguous()           # [j][co][h][t*4 + p]
        swap = ((torch.arange(64, device=w.device) >> 3) & 1).bool()
        steps[:, swap] = steps[:, swap].flip(2)
        return steps.reshape(13, 64, 16)
    if kind == "heads_narrow_x3":
        px = 32 // w.shape[0]
        bank = _shifted_bank(w.float(), px).reshape(32, w.shape[1], w.shape[2], w.shape[3] + px - 1)
    else:
        co, ci, kh, kw = w.shape
        px = 32 // cin_pad
        wf = torch.zeros((cin_pad, cout_pad, kh, kw), dtype=torch.float32, device=w.device)
        wf[:ci, :co] = w.float().flip(2, 3).permute(1, 0, 2, 3)
        bank = _shifted_bank(wf, px).reshape(px * cin_pad, cout_pad, kh, kw + px - 1)
    rows_, ch, kh, kww = bank.shape
    assert rows_ == 32 and ch == 64
    v = bank.permute(2, 3, 1, 0).reshape(kh * kww, 4, 2, 8, 32).permute(1, 0, 2, 4, 3)                  # [q][tap][half][row][8]
    ntp = (kh * kww + 7) // 8 * 8
    return torch.nn.functional.pad(v.reshape(4, kh * kww, 512), (0, 0, 0, ntp - kh * kww))


def _prepped(w, kind, cout_pad, cin_pad, stride, owner=None, half=False):
    """Re-laid-out copy of an OIHW weight; recomputed only when the parameter changed
    (optimizer steps bump ``_version``).  ``owner``: the parameter(s) ``w`` was derived from when ``w`` itself is a
    fresh tensor on every call (a reshaped Linear weight, the concatenated image heads): the cache then lives on the
    owner(s) and is keyed on their versions, so the derived tensor does not defeat it."""
    owners = (w,) if owner is None else (tuple(owner) if isinstance(owner, (tuple, list)) else (owner,))
    anchor = owners[0]
    slot = _WCACHE.get(id(anchor))
    if slot is None or slot[0]() is not anchor:
        wid = id(anchor)
        slot = (weakref.ref(anchor, lambda _r, wid=wid: _WCACHE.pop(wid, None)), {})
        _WCACHE[wid] = slot
    ent = slot[1]
    key = (kind, cout_pad, cin_pad, stride, tuple(w.shape), bool(half))
    # validity = (version counter, storage address) of every owner: writers that go through ``.data``
    # (dist.broadcast(t.data), load_state_dict on a rebound tensor, Module.to()) do not always bump the version but most
    # of them move or re-bind the storage; the in-tree ``.data`` writers bump the version explicitly
    # (dp.broadcast_module, host.moving_average, host.weights_init)
    stamp = tuple((o._version, o.data_ptr()) for o in owners)
    if ent.get(key, (None, None))[0] == stamp:
        return ent[key][1]
    lib = _lib.load()
    if kind in ("x3_fwd", "x3_dgrad"):
        # three bf16 planes per (tap, 16-channel slab) [tap][slab][plane][row][16] (dwc_x3_weight_prepare); rows = cout_pad
        # forward / cin_pad for the data gradient, the contraction runs over the other (zero-padded) channel count
        co, ci, kh, kw = w.shape
        wz = w.detach()
        if co != cout_pad or ci != cin_pad:
            wz = torch.zeros((cout_pad, cin_pad, kh, kw), dtype=torch.float32, device=w.device)
            wz[:co, :ci] = w.detach()
        dg = kind == "x3_dgrad"
        rows, kdim = (cin_pad, cout_pad) if dg else (cout_pad, cin_pad)
        out = torch.empty(lib.dwc_x3_weight_prepared_elems(rows, kdim, kh), dtype=BF16, device=w.device)
        _lib.check(lib.dwc_x3_weight_prepare(wz.contiguous().data_ptr(), out.data_ptr(), cout_pad, cin_pad, kh, rows, int(dg),
                                             _stream()), "x3_weight_prepare")
        recipe = None
        if co == cout_pad and ci == cin_pad:
            recipe = _recipe(w, owners, kind=5 if dg else 4, n_items=kh * kh * ((kdim + 15) // 16) * rows * 16, Cout=co, Cin=ci,
                             KH=kh, KW=kw, rows=rows, kdim=kdim)
        ent[key] = (stamp, out, recipe)
        return out
    if kind in ("h2_fwd", "h2_dgrad"):
        # two f16 planes of s_w * w per (tap, 16-channel slab) + {s_w, 1 / s_w} (dwc_h2_weight_prepare), s_w from the filter's largest
        # magnitude (dwc_absmax: padding zeros do not change it)
        co, ci, kh, kw = w.shape
        wz = w.detach()
        if co != cout_pad or ci != cin_pad:
            wz = torch.zeros((cout_pad, cin_pad, kh, kw), dtype=torch.float32, device=w.device)
            wz[:co, :ci] = w.detach()
        wz = wz.contiguous()
        dg = kind == "h2_dgrad"
        rows, kdim = (cin_pad, cout_pad) if dg else (cout_pad, cin_pad)
        # (zeros: the last 8 bytes are the filter's absmax slot of the fused refresh, which must start below every epoch)
        out = torch.zeros(lib.dwc_h2_weight_prepared_elems(rows, kdim, kh), dtype=torch.float16, device=w.device)
        slot, ep = amax_slot(w.device)
        _lib.check(lib.dwc_absmax(wz.data_ptr(), wz.numel(), slot, ep, _stream()), "absmax(weight)")
        _lib.check(lib.dwc_h2_weight_prepare(wz.data_ptr(), out.data_ptr(), cout_pad, cin_pad, kh, rows, int(dg), slot, ep, _stream()),
                   "h2_weight_prepare")
        recipe = None
        if co == cout_pad and ci == cin_pad:
            recipe = _recipe(w, owners, kind=9 if dg else 8, n_items=kh * kh * ((kdim + 15) // 16) * rows * 16, Cout=co, Cin=ci,
                             KH=kh, KW=kw, rows=rows, kdim=kdim)
        ent[key] = (stamp, out, recipe)
        return out
    if kind in ("stem_steps", "stem_steps_dgrad"):
        # csrc/conv_narrow_bf16.hip conv_stem_kernel: [25 k-steps][64 channels][2 taps x 8 planes] bf16, halves of a row swapped
        # by (co>>3)&1.  stem_steps: w is the stem filter [64][P<=8][7][7]; stem_steps_dgrad: w is the heads filter
        # [P<=8][64][7][7], rotated and transposed (the data gradient is a convolution of the 8-plane gradient image)
        wf = w.detach().float()
        if kind == "stem_steps_dgrad":
            wf = wf.flip(2, 3).permute(1, 0, 2, 3)                      # [64][P][7][7]
        co, pl, kh, kw = wf.shape
        assert co == 64 and pl <= 8 and kh == 7 and kw == 7
        bank = torch.zeros((64, 50, 8), dtype=torch.float32, device=w.device)      # [co][tap (49 + 1 zero)][plane]
        bank[:, :49, :pl] = wf.permute(0, 2, 3, 1).reshape(64, 49, pl)
        steps = bank.view(64, 25, 2, 8).permute(1, 0, 2, 3).contiguous()           # [j][co][h][p]
        swap = ((torch.arange(64, device=w.device) >> 3) & 1).bool()
        # (torch.where, not boolean-mask indexing: a mask index is a nonzero() = a host synchronisation, six per c2 iteration -- r06)
        steps = torch.where(swap.view(1, 64, 1, 1), steps.flip(2), steps)
        out = steps.reshape(25, 64, 16).to(BF16).contiguous()
        ent[key] = (stamp, out)
        return out
    if kind in ("stem_steps_x3", "stem_steps_dgrad_x3", "heads_narrow_x3", "dgrad_image_narrow_x3"):
        # Split filter banks of csrc/conv_narrow_x3.hip: a fixed permutation (+ zero padding) of the OIHW filter, split into three
        # exact bf16 planes.  The permutation is built ONCE per layout as an index table (the torch recipe below applied to the
        # element numbers); every later refresh is one launch (dwc_x3_gather_split).
        ikey = (kind, tuple(w.shape), cout_pad, cin_pad, str(w.device))
        idx = _X3_BANK_IDX.get(ikey)
        if idx is None:
            num = (torch.arange(w.numel(), dtype=torch.float32, device=w.device) + 1.0).view(w.shape)      # 0 = "zero element"
            idx = (_x3_bank_recipe(kind, num, cout_pad, cin_pad).reshape(-1).round().to(torch.int32) - 1).contiguous()
            _X3_BANK_IDX[ikey] = idx
        n = idx.numel()
        src = w.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        out = torch.empty(3 * n, dtype=BF16, device=w.device)
        _lib.check(lib.dwc_x3_gather_split(src.data_ptr(), idx.data_ptr(), out.data_ptr(), n, _stream()), "x3_gather_split")
        if kind.startswith("stem_steps"):
            assert out.numel() == lib.dwc_x3_conv2d_stem_weight_elems()             # [plane][13][64][16]
        else:
            # [plane][q][tap][lane][8] -> the kernel's [q][tap][plane][lane][8]
            ntp = n // (4 * 512)
            out = out.view(3, 4, ntp, 512).permute(1, 2, 0, 3).contiguous().view(-1)
            assert out.numel() == lib.dwc_x3_conv2d_narrow_weight_elems(7, 14)
        ent[key] = (stamp, out)
        return out
    if kind in ("heads_narrow", "dgrad_image_narrow"):
        # the wide bank ([32][taps*64] prepared rows) in MFMA-fragment order [tap][q][hi][row][8] for csrc/conv_narrow_bf16.hip
        base = _prepped(w, "heads_wide" if kind == "heads_narrow" else "dgrad_image", cout_pad, cin_pad, stride, owner, True)
        ch = cout_pad if kind == "dgrad_image_narrow" else cin_pad          # contraction channels (64)
        taps = base.numel() // (32 * ch)
        out = base.view(32, taps, ch // 16, 2, 8).permute(1, 2, 3, 0, 4).reshape(taps, -1)
        out = torch.nn.functional.pad(out, (0, 0, 0, (taps + 7) // 8 * 8 - taps)).contiguous()      # zero taps: every wave walks the same count
        ent[key] = (stamp, out)
        return out
    if kind == "heads_wide":
        # the P-plane heads (P = w.shape[0]: 4, or 8 on the bf16 path) as px = 32/P pixels x P planes:
        # bank [p*P + co][ci][KH][KW+px-1], copy p shifted right by p taps
        px = 32 // w.shape[0]
        bank = _shifted_bank(w.detach(), px).reshape(32, w.shape[1], w.shape[2], w.shape[3] + px - 1)
        out = _prepped(bank, "fwd", 32, cin_pad, 1, half=half)
        ent[key] = (stamp, out)
        return out
    if kind == "dgrad_image":
        # bank of px shifted copies of the flipped, transposed filter: [p*P + ci][co][KH][KW+px-1] (dwc_conv2d_bwd_data_image);
        # cout_pad = gathered (dY) channels, cin_pad = P image planes (4 fp32 / 8 bf16), px = 32 / P
        co, ci, kh, kw = w.shape
        px = 32 // cin_pad
        wf = torch.zeros((cin_pad, cout_pad, kh, kw), dtype=torch.float32, device=w.device)
        wf[:ci, :co] = w.detach().flip(2, 3).permute(1, 0, 2, 3)
        bank = _shifted_bank(wf, px).reshape(px * cin_pad, cout_pad, kh, kw + px - 1)
        out = _prepped(bank, "fwd", px * cin_pad, cout_pad, 1, half=half)
        ent[key] = (stamp, out)
        return out
    cout, cin, kh, kw = w.shape
    wc = w.detach().contiguous()
    transposed = kind == "dgrad_t"
    if transposed:                  # dgrad layout of the filter with its two spatial axes swapped (taps enumerated kw-major)
        wc = w.detach().transpose(2, 3).contiguous()
        kind = "dgrad"
    pre = "dwc_bf16_" if half else "dwc_"
    n = getattr(lib, pre + "weight_prepared_elems")(cout, cin, kh, kw, stride, cout_pad, cin_pad, int(kind == "dgrad"))
    out = torch.empty(n, dtype=BF16 if half else torch.float32, device=w.device)
    if kind == "fwd":
        _lib.check(getattr(lib, pre + "weight_prepare_fwd")(wc.data_ptr(), out.data_ptr(), cout, cin, kh, kw, cout_pad, cin_pad,
                                                            _stream()), "weight_prepare_fwd")
        rows, kdim = cout_pad, kh * kw * cin_pad
    else:
        _lib.check(getattr(lib, pre + "weight_prepare_dgrad")(wc.data_ptr(), out.data_ptr(), cout, cin, kh, kw, stride,
                                                              cout_pad, cin_pad, _stream()), "weight_prepare_dgrad")
        rows, kdim = (cin_pad, kh * kw * cout_pad) if stride == 1 else (4 * cin_pad, 4 * cout_pad)
    recipe = None
    if (not transposed or kh == kw) and n % rows == 0:
        # (Kp = the padded row length the single-layout entry point chose: n elements / rows)
        recipe = _recipe(w, owners, kind=(2 if half else 0) + (kind == "dgrad"), n_items=n, Cout=cout, Cin=cin, KH=kh, KW=kw,
                         stride=stride, cout_pad=cout_pad, cin_pad=cin_pad, Kp=n // rows, transpose_hw=int(transposed))
    ent[key] = (stamp, out, recipe)
    return out


_PCAT = {}     # id(first parameter) -> (weakref, {key: (stamp, buffer)})


class _CatParams(torch.autograd.Function):
    """torch.cat / torch.stack of PARAMETERS along dim 0, with the concatenated buffer cached until one of them changes (the
    16 style heads, the 4 x 2 LSTM direction weights: concatenated again on every one of ~20 forward calls per iteration with
    unchanged values).  Backward hands out views of the incoming gradient (no kernel)."""

    @staticmethod
    def forward(ctx, stack, *ps):
        anchor = ps[0]
        slot = _PCAT.get(id(anchor))
        if slot is None or slot[0]() is not anchor:
            aid = id(anchor)
            slot = (weakref.ref(anchor, lambda _r, aid=aid: _PCAT.pop(aid, None)), {})
            _PCAT[aid] = slot
        key = (bool(stack), tuple(id(p) for p in ps))
        stamp = tuple((p._version, p.data_ptr()) for p in ps)
        ent = slot[1].get(key)
        if ent is None or ent[0] != stamp:
            with torch.no_grad():
                buf = torch.stack([p.detach() for p in ps]) if stack else torch.cat([p.detach() for p in ps], 0)
            ent = (stamp, buf)
            slot[1][key] = ent
        ctx.stack = bool(stack)
        ctx.sizes = [p.shape[0] for p in ps]
        return ent[1].view_as(ent[1])

    @staticmethod
    def backward(ctx, g):
        parts = g.unbind(0) if ctx.stack else g.split(ctx.sizes, 0)
        return (None,) + tuple(parts)


_PCAT_LIVE = {}     # (stack, ids of the parameters) -> (stamp, weakrefs, graph-attached result)


def cat_params(params, stack=False):
    """Concatenation (``stack=True``: stack) along dim 0 of parameters of one module, cached across calls.

    Under autograd the RESULT (with its graph node) is cached too while the parameters are unchanged: a module used k times in one graph
    (the style encoder on x_real and on x_fake) then hangs off ONE node, the engine sums the k concatenated gradients (k - 1 adds) and
    every parameter receives one view -- instead of k views per parameter and (k - 1) adds for EACH of them (r06: 32 of the 478 stock
    launches of a c1 iteration).  The node saves no tensors, so it may be walked by several backward calls."""
    params = list(params)
    if not params[0].is_cuda:
        return torch.stack(params) if stack else torch.cat(params, 0)
    if not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
        return _CatParams.apply(stack, *params)
    key = (bool(stack), tuple(id(p) for p in params))
    stamp = tuple((p._version, p.data_ptr(), p.requires_grad) for p in params)
    ent = _PCAT_LIVE.get(key)
    if ent is not None and ent[0] == stamp and all(r() is p for r, p in zip(ent[1], params)):
        return ent[2]
    out = _CatParams.apply(stack, *params)
    if len(_PCAT_LIVE) > 256:          # (ids of dead parameters: entries are small, but do not let them pile up)
        for k in [k for k, e in _PCAT_LIVE.items() if any(r() is None for r in e[1])]:
            del _PCAT_LIVE[k]
    _PCAT_LIVE[key] = (stamp, [weakref.ref(p) for p in params], out)
    return out


def _shifted_bank(w, px):
    """[px][*w.shape[:-1]][KW + px - 1]: copy p holds w shifted right by p taps, zeros elsewhere (the "wide" filter banks of the
    image heads).  One fill + ONE strided copy (the diagonal view's stride along p is the bank's plus one tap) instead of px
    pad + stack launches."""
    kw = w.shape[-1]
    bank = torch.zeros((px,) + tuple(w.shape[:-1]) + (kw + px - 1,), dtype=w.dtype, device=w.device)
    diag = bank.as_strided((px,) + tuple(w.shape), (bank.stride(0) + 1,) + tuple(bank.stride()[1:]))
    diag.copy_(w.unsqueeze(0).expand((px,) + tuple(w.shape)))
    return bank


_REFRESH_DT = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n_items", "<u8"), ("kind", "<i4"), ("Cout", "<i4"), ("Cin", "<i4"),
                        ("KH", "<i4"), ("KW", "<i4"), ("stride", "<i4"), ("cout_pad", "<i4"), ("cin_pad", "<i4"), ("Kp", "<i4"),
                        ("rows", "<i4"), ("kdim", "<i4"), ("transpose_hw", "<i4"), ("reserved", "<i4"), ("tail_pad", "<i4")])   # struct dwc_refresh_desc (80 bytes)
assert _REFRESH_DT.itemsize == 80
REFRESH_CHUNK = 8192           # DWC_OPT_CHUNK of include/dwcgan_hip.h
_REFRESH_TABLES = {}           # (src, dst) pointers of a refresh set -> (device descriptor table, chunk maps)
REFRESH_STATS = {"launches": 0, "layouts": 0}
_REFRESH_EPOCH = 0


def _recipe(w, owners, **fields):
    """Descriptor fields with which dwc_weight_refresh_multi can rebuild this layout from the owning parameter's storage, or
    None when the layout was not built straight from it (derived banks: the fused heads, the NHWC8 stems)."""
    if len(owners) != 1 or w.data_ptr() != owners[0].data_ptr() or not w.detach().is_contiguous() or w.dtype != torch.float32:
        return None
    return fields


def refresh_prepared(params):
    """Rebuild EVERY stale prepared layout of `params` in ONE launch (dwc_weight_refresh_multi; FusedAdam.step calls this right
    behind the update: SURVEY.md section 8(f) rank 1).  Layouts that were never used, or that are derived through torch ops (no
    recipe), stay with the lazy per-layout path of _prepped.  Returns the number of layouts rebuilt."""
    todo = []
    for p in params:
        slot = _WCACHE.get(id(p))
        if slot is None or slot[0]() is not p:
            continue
        stamp = ((p._version, p.data_ptr()),)
        for key, val in slot[1].items():
            if len(val) > 2 and val[2] is not None and val[0] != stamp:
                todo.append((p, slot[1], key, val, stamp))
    if not todo:
        return 0
    lib = _lib.load()
    dev = todo[0][0].device
    # (the recipe is part of the identity: a lazily rebuilt layout may reuse the address of another layout of the same parameter)
    ident = tuple((p.data_ptr(), val[1].data_ptr(), tuple(sorted(val[2].items()))) for p, _, _, val, _ in todo)
    table = _REFRESH_TABLES.get(ident)
    if table is None:
        desc = np.zeros(len(todo), dtype=_REFRESH_DT)
        cd, cs = [], []
        for i, (p, _, _, val, _) in enumerate(todo):
            desc[i]["src"], desc[i]["dst"] = p.data_ptr(), val[1].data_ptr()
            for k, v in val[2].items():
                desc[i][k] = v
            for s0 in range(0, int(val[2]["n_items"]), REFRESH_CHUNK):
                cd.append(i)
                cs.append(s0)
        host = torch.from_numpy(desc.view(np.uint8).reshape(-1)).pin_memory()
        table = (host.to(dev, non_blocking=True), torch.tensor(cd, dtype=torch.int32, device=dev),
                 torch.tensor(np.array(cs, dtype=np.uint32).view(np.int32), dtype=torch.int32, device=dev), len(cd),
                 int(any(int(v[3][2]["kind"]) in (8, 9) for v in todo)))
        if len(_REFRESH_TABLES) > 16:
            _REFRESH_TABLES.clear()
        _REFRESH_TABLES[ident] = table
    global _REFRESH_EPOCH
    _REFRESH_EPOCH += 1          # (epoch of the filters' absmax slots this refresh raises: larger at every call)
    _lib.check(lib.dwc_weight_refresh_multi(table[0].data_ptr(), table[1].data_ptr(), table[2].data_ptr(), table[3], table[4],
                                            _REFRESH_EPOCH, _stream()), "weight_refresh_multi")
    for p, ent, key, val, stamp in todo:
        ent[key] = (stamp, val[1], val[2])
    REFRESH_STATS["launches"] += 1
    REFRESH_STATS["layouts"] += len(todo)
    return len(todo)


# --------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------
# bf16 path: halo-tiled kernel for the stride-1 "same" 3x3 / 5x5 layers (0: im2col GEMM everywhere; development knob)
# fp32 stride-1 "same" 5x5 (and with DWC_X3=2 also 3x3) convolutions as exact three-way bf16 splits on the bf16 MFMA
# (csrc/conv_halo_x3.hip): 0 = native fp32 MFMA kernels only
# (r04: every stride-1 3x3 / 5x5 layer as split products, forward, data gradient and weight gradient.  The Winograd F(2x2,3x3) /
# F(4x4,3x3) family of rounds 1-4 -- fp32 MFMA, 2.25x / 4x fewer multiply-adds, 3 - 30x the rounding error -- lost every A/B
# against the split products from r04 on and was removed in r05; DWC_X3=1 keeps the 3x3 layers on the im2col kernels.)
X3 = int(os.environ.get("DWC_X3", "2"))
HALO = int(os.environ.get("DWC_BF16_HALO", "1"))
WGRAD_HALO = int(os.environ.get("DWC_BF16_WGRAD_HALO", "1"))
STEM = int(os.environ.get("DWC_BF16_STEM", "1"))
NARROW = int(os.environ.get("DWC_BF16_NARROW", "1"))      # 64 -> 8-plane 7x7 convolutions on csrc/conv_narrow_bf16.hip
NARROW_X3 = int(os.environ.get("DWC_X3_NARROW", "1"))     # fp32: 64 -> 4-plane 7x7 convolutions on csrc/conv_narrow_x3.hip (split products)
SMALLK_X3 = int(os.environ.get("DWC_X3_SMALLK", "1"))     # fp32 7x7 weight gradients on smallk_wgrad_x3_kernel (0: the im2col weight gradient)
LSTM_SEQ = int(os.environ.get("DWC_LSTM_SEQ", "1"))       # text-encoder LSTM forward: all time steps in one persistent launch
DGRAD_FOLD = int(os.environ.get("DWC_DGRAD_FOLD", "1"))   # data gradients through a reflect pad: interior straight into dx + band fold
X3_S2 = int(os.environ.get("DWC_X3_S2", "1"))             # fp32 stride-2 4x4 forwards as split products (csrc/conv_halo_x3.hip, S2)
S2HALO = int(os.environ.get("DWC_BF16_S2_HALO", "1"))     # stride-2 4x4 forwards on the halo kernel over the space-to-depth image
S2DGRAD = int(os.environ.get("DWC_S2_DGRAD_HALO", "1"))   # stride-2 4x4 DATA GRADIENTS in halo form (interior) + ring strips, both precisions
X3_WGRAD_HALO3 = int(os.environ.get("DWC_X3_WGRAD_HALO3", "1"))   # 0: 3x3 weight gradients on the im2col kernel (split-product inner product)
PINNED_STAGE = int(os.environ.get("DWC_PINNED_STAGE", "1"))        # small host->device tables through pinned staging (0: pageable copies, which wait for the stream)
ZERO_GRAD_BY_FLAG = int(os.environ.get("DWC_ZERO_GRAD_FLAG", "1"))   # biases whose gradient is identically zero: flagged, not filled (0: torch.zeros per use)
RING_FUSED = int(os.environ.get("DWC_RING_FUSED", "1"))   # stride-1 data gradients: border ring inside the halo launch (0: strip GEMM + fold launches)
S2DGRAD_MIN_WGS = 192        # below this many workgroups (4 classes x blocks x 64-channel tiles) the im2col GEMM keeps the layer


# Planes per operand of the split-product halo kernels (csrc/conv_halo_x3.hip): 3 = exact three-way bf16 split, six MFMAs per
# fp32 MFMA-equivalent; 2 = f16 hi / lo split with per-tensor power-of-two scales, THREE MFMAs (r05 default; same fp32-size error,
# tests/test_x3_parity.py).  The two-plane kernels take each activation operand's largest magnitude from an "absmax slot".
X3_PLANES = int(os.environ.get("DWC_X3_PLANES", "2"))
AMAX_SLOTS = 1 << 16
_AMAX = {}      # device index -> [int64 tensor of AMAX_SLOTS slots, slots handed out so far]


def amax_slot(dev):
    """A fresh (slot address, epoch) pair: slots are handed out round robin, the epoch of a slot grows by one per lap, and a slot is
    raised by atomic max on (epoch << 32 | magnitude bits) -- so nothing is ever zeroed.  A pair stays valid for AMAX_SLOTS later
    pairs (hundreds of iterations); a consumer that meets another epoch poisons its result with NaN."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _AMAX.get(key)
    if pool is None:
        pool = [torch.zeros(AMAX_SLOTS, dtype=torch.int64, device=dev), 0]
        _AMAX[key] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0].data_ptr() + 8 * (i % AMAX_SLOTS), i // AMAX_SLOTS + 1


def amax_live(t, c=False):
    """The (slot, epoch) pair attached to ``t`` (or the record ``c`` taken from it earlier) if it still describes it: same version
    counter and address, and the slot has not come up for re-use -- a pair is trusted for AMAX_SLOTS - 4096 later pairs only (a
    long-lived tensor, e.g. a fixed input batch, is measured again instead of meeting a slot that a later lap of the pool has
    raised to another epoch)."""
    if c is False:
        c = getattr(t, "_dwc_amax", None)
    if c is None or c[2] != t._version or c[3] != t.data_ptr():
        return None
    pool = _AMAX.get(t.device.index if t.device.index is not None else torch.cuda.current_device())
    if pool is None:
        return None
    index = (c[1] - 1) * AMAX_SLOTS + (c[0] - pool[0].data_ptr()) // 8
    return (c[0], c[1]) if 0 <= pool[1] - index < AMAX_SLOTS - 4096 else None


def amax_of(t):
    """(slot, epoch) holding the largest magnitude of the dense fp32 tensor ``t`` -- the pair a producing op attached to it
    (``set_amax``) if it still describes it, else one pass of dwc_absmax."""
    c = amax_live(t)
    if c is not None:
        if AMAX_CHECK:
            _amax_verify(t, c)
        return c
    if t.dtype != torch.float32 or not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("amax_of: dense fp32 tensor expected")
    slot, ep = amax_slot(t.device)
    _lib.check(_lib.load().dwc_absmax(t.data_ptr(), t.numel(), slot, ep, _stream()), "absmax")
    set_amax(t, slot, ep)
    return slot, ep


def h2_fits(t):
    """The two-plane halo kernels address their activation operand through 31-bit byte offsets (bit 31 marks 'reads as zero'):
    larger tensors (fp32, 3 x 128 images of 128 channels at 128 x 128) stay on the three-plane kernels."""
    return X3_PLANES == 2 and t.numel() * 4 < (1 << 31)


def set_amax(t, slot, ep):
    """Tag ``t`` with the slot that bounds its magnitude.  INVARIANT the tag rests on: nothing writes into ``t`` after it was tagged.
    The tag is checked against ``t._version`` and the address, but the HIP entry points write through raw pointers and never bump the
    version counter -- an op that writes a tensor it did not allocate (in-place add into y, buffer re-use) must clear
    ``t._dwc_amax`` itself, or the two-plane kernels would scale by a stale, possibly too small bound (f16 overflow to inf; the NaN
    poison only covers a stale EPOCH).  ``DWC_AMAX_CHECK=1`` re-measures every tagged tensor at the point of use and raises when the
    tag is smaller than the truth."""
    t._dwc_amax = (slot, ep, t._version, t.data_ptr())
    return t


AMAX_CHECK = int(os.environ.get("DWC_AMAX_CHECK", "0"))


def _amax_verify(t, c):
    """DWC_AMAX_CHECK=1 (debugging, synchronises): the slot attached to ``t`` must hold at least max|t|."""
    pool = _AMAX[t.device.index if t.device.index is not None else torch.cuda.current_device()][0]
    word = int(pool[(c[0] - pool.data_ptr()) // 8])
    bits = word & 0xffffffff
    if (word >> 32) != c[1]:
        raise RuntimeError("DWC_AMAX_CHECK: slot epoch %d, tag says %d" % (word >> 32, c[1]))
    true_bits = int(t.detach().abs().max().view(torch.int32)) if t.numel() else 0
    if true_bits > bits:
        raise RuntimeError("DWC_AMAX_CHECK: stale absmax tag -- slot holds %g, the tensor's largest magnitude is %g" % (
            np.uint32(bits).view(np.float32), np.uint32(true_bits).view(np.float32)))


def out_amax(t):
    """(slot, epoch) for a kernel that is about to WRITE the fp32 tensor ``t`` and can raise its absmax slot while it does
    (dwc_*_amax entry points), or (None, 0) when nobody will ask (bf16 tensors, three-plane kernels)."""
    if X3_PLANES != 2 or t.dtype != torch.float32:
        return None, 0
    return amax_slot(t.device)


def pass_amax(src, dst):
    """``dst`` is bounded by ``src`` in magnitude (bilinear up-sampling, average pooling: convex combinations): it inherits src's slot
    -- an upper bound is all the two-plane kernels need (the scale has 2^27 of headroom)."""
    c = amax_live(src)
    if c is not None:
        set_amax(dst, c[0], c[1])
    return dst


def _x3_use(lib, B, H, W, c_in, c_out, KH, KW, stride, pad, free=False):
    """Whether this fp32 stride-1 'same' convolution (c_in gathered channels -> c_out) runs as split-bf16 products.
    5x5: always when the shape is handled.  3x3: with DWC_X3=1 only where ``free`` (a data gradient, or a forward whose weights need
    no gradient), everywhere with the default DWC_X3=2."""
    if not X3 or stride != 1 or KH != KW or 2 * pad != KH - 1 or c_in % 16 or c_out % 16:
        return False
    if KH == 3 and X3 < 2 and not free:
        return False
    return bool(lib.dwc_x3_conv2d_same_ok(B, H, W, c_in, c_out, KH))


class ResGradToken:
    """Hands the identity-branch gradient of a residual block (``y = x + f(x)``, reference networks.py:521) from the op that
    receives it (the norm the add rides on) to the data gradient of f's FIRST convolution, which adds it in its epilogue
    (dwc_*_conv2d_same*_add) instead of autograd summing the two gradients of x in a pass of its own.  Autograd runs the norm's
    backward before that convolution's (the convolution's gradient depends on it), every backward pass sets ``g`` afresh."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


RES_FUSE = int(os.environ.get("DWC_RES_FUSE", "1"))      # 0: autograd adds the two gradients of a ResBlock input itself


def res_token(x):
    """A token for one residual block, or None when nothing would consume it."""
    return ResGradToken() if RES_FUSE and torch.is_grad_enabled() and x.requires_grad else None


_X3_TICKETS = {}


def _x3_ksplit(lib, dev, B, H, W, Cx, cop, K, stride, at_least=0):
    """(scratch arena, its size, ticket row) for the contraction split of small split-product launches
    (dwc_x3_conv2d_same_add_ws / dwc_x3_conv2d_s2_ws), the arena at least ``at_least`` bytes; ticket row None: shape not split.
    One ticket row per (device, stream), zeroed once -- every launch leaves it at zero."""
    need = lib.dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cx, cop, K, stride)
    if not need:
        ws = workspace(at_least, dev) if at_least else None
        return ws, at_least, None
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    row = _X3_TICKETS.get(key)
    if row is None:
        row = torch.zeros(int(lib.dwc_x3_conv2d_ksplit_ticket_words()), dtype=torch.int32, device=dev)
        _X3_TICKETS[key] = row
    n = max(int(need), int(at_least))
    return workspace(n, dev), n, row.data_ptr()


_KSPLIT_POLL = {}            # (device index, stream) -> [pinned host copy of the row's status word, event of the last copy or None]


def ksplit_status_poll(wait=False):
    """Check the sticky status word behind every contraction-split ticket row (dwc_x3_conv2d_ksplit_ticket_words) without stalling
    the stream, the way lstm_status_poll does: each call looks at the copy the PREVIOUS call started (if it has landed, or ``wait``)
    and starts a new one.  Set status = a tile's two halves missed each other (a ticket left dirty by an aborted launch) or ran on
    different XCDs; the kernel wrote NaN into that tile.  Raises, after re-zeroing the ticket row so that later launches do not keep
    timing out on it.  Solver calls this once per step."""
    for key, row in list(_X3_TICKETS.items()):
        if key[1] != _stream():                  # a row is polled -- and, on an error, re-zeroed -- from the stream its launches run on
            continue
        ent = _KSPLIT_POLL.get(key)
        if ent is None:
            ent = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
            _KSPLIT_POLL[key] = ent
        if ent[1] is not None and (wait or ent[1].query()):
            if wait:
                ent[1].synchronize()
            ent[1] = None
            status = int(ent[0][0])
            if status != 0:
                row.zero_()
                ent[0].zero_()
                raise _lib.HipKernelError(
                    "contraction split of a split-product convolution: the two halves of a tile did not meet (status %d); the tile was "
                    "overwritten with NaN and the ticket row has been re-zeroed.  DWC_X3_KSPLIT=0 runs these launches unsplit." % status)
        if ent[1] is None:
            ent[0].copy_(row[-1:], non_blocking=True)
            ent[1] = torch.cuda.Event()
            ent[1].record()


class _Conv2d(torch.autograd.Function):
    """act(conv(reflect_pad(x)) + b).  Output has Cout rounded up to a multiple of 4 (fp32) / 8 (bf16); its dtype is x's.
    ``owner``: the parameter(s) ``w`` is derived from when ``w`` is a fresh tensor on every call (prepared-weight cache)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, bias_grad=True, owner=None, token=None, nograd=False):
        _require_device(x)
        lib = _lib.load()
        x = cl(x)
        half = x.dtype == BF16
        ctx.token = token
        B, Cx, H, W = x.shape
        Cout, Cin, KH, KW = w.shape
        if Cin > Cx:
            raise ValueError("input has %d channels, weight expects %d" % (Cx, Cin))
        ctx.bias_grad = bias_grad
        ctx.zero_flag = bool(b is not None and getattr(b, "_dwc_zero_grad", False))
        ctx.owner = owner
        cop = _padc(Cout, x.dtype)
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        # (inside torch.no_grad() needs_input_grad still reports the parameters' requires_grad: the caller tells, grad mode is always
        # off in here)
        w_grad = ctx.needs_input_grad[1] and not nograd
        use_x3 = (not half) and _x3_use(lib, B, H, W, Cx, cop, KH, KW, stride, pad, free=not w_grad)
        # (>= 160 workgroups of 256 output pixels x 64 channels: below that the launch leaves most CUs idle -- B=16 32x32 256->256 ran
        # 200 us on 64 workgroups against 118 us on the native kernel)
        use_x3s2 = bool((not half) and X3 and X3_S2 and stride == 2 and KH == 4 and KW == 4 and pad == 1 and cop % 64 == 0
                        and B * (H // 32) * (W // 32) * (cop // 64) >= 160 and lib.dwc_x3_conv2d_s2_ok(B, H, W, Cx, cop))
        use_stem = bool(half and NARROW and STEM and Cx == 8 and cop == 64 and stride == 1 and KH == 7 and KW == 7 and pad == 3
                        and lib.dwc_bf16_conv2d_stem_ok(B, H, W, H, W, KH, act))
        use_stem_x3 = bool((not half) and X3 and NARROW_X3 and Cx == 4 and cop == 64 and stride == 1 and KH == 7 and KW == 7 and pad == 3
                           and lib.dwc_x3_conv2d_stem_ok(B, H, W, H, W, KH, act))
        w_hwio = None if use_x3 or use_x3s2 or use_stem or use_stem_x3 else _prepped(w, "fwd", cop, Cx, stride, owner, half)
        bias = None
        if b is not None:
            bias = b.detach() if cop == Cout else torch.nn.functional.pad(b.detach(), (0, cop - Cout))
            bias = bias.contiguous()
        y = empty_cl(B, cop, Ho, Wo, x.device, x.dtype)
        flops = 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
        st = _stream()
        if use_x3 and h2_fits(x):
            w_h2 = _prepped(w, "h2_fwd", cop, Cx, 1, owner)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 1)
            x_amax = amax_of(x)
            ya, yep = out_amax(y)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_same_add_ws(
                x.data_ptr(), x_amax[0], x_amax[1], w_h2.data_ptr(), _p(bias), None, y.data_ptr(), ya, yep, B, H, W, Cx, cop, cop, KH, act, 1,
                _p(ks_ws), ks_n, ks_t, st), detail="fwd-h2 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride), exec_flops=3 * flops),
                "h2_conv2d_same")
            set_amax(y, ya, yep)
        elif use_x3:
            w_x3 = _prepped(w, "x3_fwd", cop, Cx, 1, owner)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 1)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_same_add_ws(
                x.data_ptr(), w_x3.data_ptr(), _p(bias), None, y.data_ptr(), B, H, W, Cx, cop, cop, KH, act, 1, _p(ks_ws), ks_n, ks_t, st),
                detail="fwd-x3 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride), exec_flops=6 * flops), "x3_conv2d_same")
        elif use_x3s2 and h2_fits(x):
            w_h2 = _prepped(w, "h2_fwd", cop, Cx, 1, owner)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 2)
            x_amax = amax_of(x)
            ya, yep = out_amax(y)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_s2_ws(
                x.data_ptr(), x_amax[0], x_amax[1], w_h2.data_ptr(), _p(bias), y.data_ptr(), ya, yep, B, H, W, Cx, cop, cop, act, _p(ks_ws), ks_n,
                ks_t, st), detail="fwd-h2s2 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride), exec_flops=3 * flops), "h2_conv2d_s2")
            set_amax(y, ya, yep)
        elif use_x3s2:
            # stride-2 4x4 layers, fp32: split products, 2x2 taps per input-pixel parity, space-to-depth in the patch gather
            w_x3 = _prepped(w, "x3_fwd", cop, Cx, 1, owner)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 2)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_s2_ws(
                x.data_ptr(), w_x3.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, cop, act, _p(ks_ws), ks_n, ks_t, st),
                detail="fwd-x3s2 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride), exec_flops=6 * flops), "x3_conv2d_s2")
        elif use_stem_x3:
            # fp32 7x7 stem on an NHWC4 image as split products: filter planes resident in LDS, persistent workgroups (csrc/conv_narrow_x3.hip)
            w_st = _prepped(w, "stem_steps_x3", cop, Cx, 1, owner)
            ya, yep = out_amax(y)                     # (the stride-2 layer behind the stem is a two-plane kernel: it wants y's absmax)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_stem_amax(
                x.data_ptr(), w_st.data_ptr(), _p(bias), y.data_ptr(), ya, yep, B, H, W, H, W, KH, -pad, act, 1, st),
                detail="fwd-stem-x3 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride), exec_flops=6 * flops), "x3_conv2d_stem")
            if ya is not None:
                set_amax(y, ya, yep)
        elif use_stem:
            # 7x7 stem on an NHWC8 image: filter resident in LDS, persistent workgroups (csrc/conv_narrow_bf16.hip)
            w_st = _prepped(w, "stem_steps", cop, Cx, 1, owner, True)
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_stem(
                x.data_ptr(), w_st.data_ptr(), _p(bias), y.data_ptr(), B, H, W, H, W, KH, -pad, act, 1, st),
                detail="fwd-stem B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_stem")
        elif (half and HALO and S2HALO and stride == 2 and KH == 4 and KW == 4 and pad == 1
              and lib.dwc_bf16_conv2d_s2_halo_ok(B, H, W, Cx, cop)):
            # stride-2 4x4 layers on the bf16 path: 2x2 taps over the space-to-depth image, space-to-depth done by the loader
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_s2_halo(
                x.data_ptr(), w_hwio.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, act, st),
                detail="fwd-s2halo B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_s2_halo")
        elif half and HALO and stride == 1 and KH == KW and 2 * pad == KH - 1 and lib.dwc_bf16_conv2d_same_halo_ok(B, H, W, Cx, cop, KH):
            # stride-1 "same" 3x3 / 5x5 layers on the bf16 path: halo-tiled kernel (patch staged once per channel slab)
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_same_halo(
                x.data_ptr(), w_hwio.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, KH, act, 1, st),
                detail="fwd-halo B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_same_halo")
        else:
            nws = _fn(lib, "conv2d_fwd_ws_bytes", x)(B, H, W, Cx, cop, KH, KW, stride, pad)     # split-K partials, usually 0
            wsp = workspace(nws, x.device).data_ptr() if nws else None
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_fwd", x)(
                x.data_ptr(), w_hwio.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, KH, KW, stride, pad, act, wsp, nws,
                st), detail="fwd B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_fwd")
        ctx.save_for_backward(x, w, y if act != 0 else None)
        ctx.x_amax = getattr(x, "_dwc_amax", None)        # (slot, epoch, version, address) if something measured x (two-plane kernels)
        ctx.geom = (B, H, W, Cx, cop, KH, KW, stride, pad, act, Cin, Cout, b is not None)
        ctx.bscope = ("bwd:" + SCOPE) if SCOPE else ""
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        B, H, W, Cx, cop, KH, KW, stride, pad, act, Cin, Cout, has_b = ctx.geom
        owner = ctx.owner
        dt = x.dtype
        half = dt == BF16
        dy = cl(dy)
        Ho, Wo = dy.shape[2], dy.shape[3]
        rows = B * Ho * Wo
        st = _stream()
        dev = x.device
        need_db = has_b and ctx.needs_input_grad[2]
        g = dy
        db = None
        if need_db and not ctx.bias_grad:
            # the output goes straight into an instance norm: a per-channel constant is removed by its mean
            # subtraction, so this gradient is identically zero (the reference computes rounding noise here).  The parameter carries
            # `_dwc_zero_grad` (set in forward): FusedAdam steps it with a shared zero vector, the data-parallel reducer does not wait for
            # it -- no gradient tensor, no fill launch (r05: 38 per iteration).  Anything else asks for a real zero tensor.
            db, need_db = (None if ctx.zero_flag else torch.zeros(Cout, dtype=torch.float32, device=dev)), False
        if act != 0 or need_db:
            db_full = torch.empty(cop, dtype=torch.float32, device=dev) if need_db else None
            g_out = empty_cl(B, cop, Ho, Wo, dev, dt) if act != 0 else None
            nws = lib.dwc_act_bwd_bias_ws_bytes(rows, cop)
            ws = workspace(nws, dev)
            ga_, gep_ = out_amax(g_out) if g_out is not None else (None, 0)
            if ga_ is not None:
                _lib.check(lib.dwc_act_bwd_bias_amax(dy.data_ptr(), _p(y), _p(g_out), _p(db_full), rows, cop, act, ws.data_ptr(),
                                                     ws.numel(), ga_, gep_, st), "act_bwd_bias")
                set_amax(g_out, ga_, gep_)
            else:
                _lib.check(_fn(lib, "act_bwd_bias", x)(dy.data_ptr(), _p(y), _p(g_out), _p(db_full), rows, cop, act, ws.data_ptr(),
                                                       ws.numel(), st), "act_bwd_bias")
            _hbm("act_bwd_bias", dy.numel() * dy.element_size() * ((1 if y is None else 2) + (1 if g_out is not None else 0)))
            if g_out is not None:
                g = g_out
            if need_db:
                db = db_full[:Cout]
        dx = dw = None
        # identity-branch gradient of the residual block this convolution opens (ResGradToken), channels-last like dx
        g_res = None
        if ctx.token is not None:
            g_res, ctx.token.g = ctx.token.g, None
            if g_res is not None:
                g_res = cl(g_res)
                if g_res.shape[1] != Cx or g_res.dtype != dt or not ctx.needs_input_grad[0]:
                    raise RuntimeError("residual-gradient token: shape / dtype of the identity branch does not match the block input")
        pow2 = (cop & (cop - 1)) == 0
        if ctx.needs_input_grad[1]:
            dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=dev)
            flops = 2.0 * rows * Cout * Cin * KH * KW
            detail = "wgrad B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)
            if half and NARROW and STEM and Cx == 8 and cop == 64 and stride == 1 and KH == 7 and KW == 7 and pad == 3 and Cout == 64:
                # 7x7 stem on an NHWC8 image: 4 taps x 8 planes per MFMA row tile (csrc/conv_narrow_bf16.hip)
                ws = workspace(lib.dwc_bf16_conv7_smallk_wgrad_ws_bytes(B, H, W, 0), dev)
                _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: lib.dwc_bf16_conv7_smallk_wgrad(
                    x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, H, W, Cin, 0, ws.data_ptr(), ws.numel(), st),
                    scope_name=ctx.bscope, detail="wgrad-stem" + detail[5:]), "conv7_smallk_wgrad")
            elif ((not half) and X3 and SMALLK_X3 and Cx == 4 and cop == 64 and stride == 1 and KH == 7 and KW == 7 and pad == 3
                  and Cout == 64):
                # fp32 7x7 stem on an NHWC4 image: one filter row (8 taps x 4 planes) per MFMA row tile, split products
                # (csrc/conv_narrow_x3.hip smallk_wgrad_x3_kernel)
                ws = workspace(lib.dwc_x3_conv7_smallk_wgrad_ws_bytes(B, H, W, 0), dev)
                _lib.check(_timed("wgrad_x3_kernel+reduce", flops, lambda: lib.dwc_x3_conv7_smallk_wgrad(
                    x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, H, W, Cin, 0, ws.data_ptr(), ws.numel(), st),
                    scope_name=ctx.bscope, detail="wgrad-stem-x3" + detail[5:], exec_flops=6 * flops), "x3_conv7_smallk_wgrad")
            elif ((not half) and (_x3_use(lib, B, H, W, Cx, cop, KH, KW, stride, pad)
                                  or (X3 and X3_S2 and stride == 2 and KH == 4 and KW == 4 and pad == 1))
                  and (KH != 3 or X3_WGRAD_HALO3)
                  and lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, W, Cx, cop, KH)):
                ws = workspace(lib.dwc_x3_conv2d_wgrad_ws_bytes(B, H, W, Cx, cop, KH), dev)
                if X3_PLANES == 2:
                    xa = amax_live(x, ctx.x_amax) or amax_of(x)          # the forward's measurement if it still stands
                    ga = amax_of(g)
                    _lib.check(_timed("wgrad_x3_kernel+reduce", flops, lambda: lib.dwc_h2_conv2d_wgrad(
                        x.data_ptr(), xa[0], xa[1], g.data_ptr(), ga[0], ga[1], dw.data_ptr(), B, H, W, Cx, cop, KH, Cin, Cout, ws.data_ptr(),
                        ws.numel(), st), scope_name=ctx.bscope, detail="wgrad-h2" + detail[5:], exec_flops=3 * flops), "h2_conv2d_wgrad")
                else:
                    _lib.check(_timed("wgrad_x3_kernel+reduce", flops, lambda: lib.dwc_x3_conv2d_wgrad(
                        x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, H, W, Cx, cop, KH, Cin, Cout, ws.data_ptr(), ws.numel(), st),
                        scope_name=ctx.bscope, detail="wgrad-x3" + detail[5:], exec_flops=6 * flops), "x3_conv2d_wgrad")
            elif (half and WGRAD_HALO and KH == KW and ((stride == 1 and KH in (3, 5) and 2 * pad == KH - 1)
                                                         or (S2HALO and stride == 2 and KH == 4 and pad == 1))
                  and lib.dwc_bf16_conv2d_wgrad_halo_ws_bytes(B, H, W, Cx, cop, KH)):
                ws = workspace(lib.dwc_bf16_conv2d_wgrad_halo_ws_bytes(B, H, W, Cx, cop, KH), dev)
                _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: lib.dwc_bf16_conv2d_wgrad_halo(
                    x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, H, W, Cx, cop, KH, Cin, Cout, ws.data_ptr(), ws.numel(), st),
                    scope_name=ctx.bscope, detail="wgrad-halo" + detail[5:]), "conv2d_wgrad_halo")
            else:
                nws = _fn(lib, "conv2d_bwd_weight_ws_bytes", x)(B, H, W, Cx, cop, KH, KW, stride, pad)
                ws = workspace(nws, dev)
                _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: _fn(lib, "conv2d_bwd_weight", x)(
                    x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, H, W, Cx, cop, KH, KW, stride, pad, Cin, Cout, ws.data_ptr(),
                    ws.numel(), st), scope_name=ctx.bscope, detail=detail), "conv2d_bwd_weight")
        same = stride == 1 and pad > 0 and 2 * pad == KH - 1 and KH == KW and cop >= 32 and pow2
        if ctx.needs_input_grad[0] and Cx == image_planes(dt) and same:
            # gradient w.r.t. an NHWC4 / NHWC8 image (7x7 stems): 32/Cx pixels x Cx planes per GEMM row,
            # see dwc_conv2d_bwd_data_image
            dx = empty_cl(B, Cx, H, W, dev, dt)
            flops = 2.0 * rows * Cout * Cin * KH * KW
            nws = _fn(lib, "conv2d_bwd_data_image_ws_bytes", x)(B, H, W, cop, KH, KW, pad)
            ws = workspace(nws, dev)
            wg4 = (W + 2 * pad + 3) // 4
            if half and NARROW and lib.dwc_bf16_conv2d_narrow_ok(B, H, W, cop, H + 2 * pad, wg4, KH, KW + 3):
                w_frag = _prepped(w, "dgrad_image_narrow", cop, Cx, 1, owner, True)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_bwd_data_image_narrow(
                    g.data_ptr(), w_frag.data_ptr(), dx.data_ptr(), B, H, W, cop, KH, KW, pad, ws.data_ptr(), ws.numel(), st),
                    scope_name=ctx.bscope, detail="dgrad-image-narrow B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)),
                    "conv2d_bwd_data_image_narrow")
            elif ((not half) and X3 and NARROW_X3 and Cx == 4 and cop == 64
                  and lib.dwc_x3_conv2d_narrow_ok(B, H, W, cop, H + 2 * pad, (W + 2 * pad + 7) // 8, KH, KW + 7)):
                # fp32: the padded gradient image on the split-product narrow kernel (zero rule), folded by the reflect adjoint
                wg8 = (W + 2 * pad + 7) // 8
                w_frag = _prepped(w, "dgrad_image_narrow_x3", cop, Cx, 1, owner)
                _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_narrow(
                    g.data_ptr(), w_frag.data_ptr(), None, ws.data_ptr(), B, H, W, cop, H + 2 * pad, wg8, KH, KW + 7, -(KH - 1), -(KW - 1),
                    0, 0, st), scope_name=ctx.bscope, exec_flops=6 * flops,
                    detail="dgrad-image-nx3 B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "x3_conv2d_narrow dgrad")
                _lib.check(lib.dwc_reflect_pad_adjoint_pitch(ws.data_ptr(), dx.data_ptr(), B, H, W, Cx, pad, 8 * wg8, st),
                           "reflect_pad_adjoint_pitch")
            else:
                w_img = _prepped(w, "dgrad_image", cop, Cx, 1, owner, half)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data_image", x)(
                    g.data_ptr(), w_img.data_ptr(), dx.data_ptr(), B, H, W, cop, KH, KW, pad, ws.data_ptr(), ws.numel(), st),
                    scope_name=ctx.bscope, detail="dgrad-image B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)),
                    "conv2d_bwd_data_image")
        elif ctx.needs_input_grad[0] and same and min(H, W) >= 2 * pad + 2 and (not half or cop >= 64):
            # "same" convolutions: interior on the H x W grid straight into dx + the thin border ring (no padded image)
            w_dg = lambda: _prepped(w, "dgrad", cop, Cx, 1, owner, half).data_ptr()        # (lazily: the fused fp32 form reads neither)
            w_dg_t = lambda: _prepped(w, "dgrad_t", cop, Cx, 1, owner, half).data_ptr()     # (only the strip GEMMs of the ring read it)
            dx = empty_cl(B, Cx, H, W, dev, dt)
            flops = 2.0 * rows * Cout * Cin * KH * KW
            nws = _fn(lib, "conv2d_bwd_data_same_ws_bytes", x)(B, H, W, Cx, cop, KH, KW, pad)
            x3 = (not half) and _x3_use(lib, B, H, W, cop, Cx, KH, KW, stride, pad, free=True)
            if x3:
                # interior = zero-padded convolution of dY with the rotated filter on the split-product kernel; ring direct
                w_x3 = _prepped(w, "x3_dgrad", cop, Cx, 1, owner) if not h2_fits(g) else None
                # (one arena: the interior's half sums -- small launches, contraction split -- are dead when the ring strips start)
                fused = bool(RING_FUSED and h2_fits(g) and min(H, W) >= 32)
                ws, ks_n, ks_t = _x3_ksplit(lib, dev, B, H, W, cop, Cx, KH, 1, at_least=0 if fused else nws)

                shape = " B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)
                # (two spans: the interior launch carries the layer's flops, the ring strips + fold are time on top of it)
                if fused:
                    # interior AND border ring in one launch: border tiles read pre-summed patch pixels (r06, conv_halo_x3_kernel RING)
                    w_h2 = _prepped(w, "h2_dgrad", cop, Cx, 1, owner)
                    ga = amax_of(g)
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_bwd_data_same_fused(
                        g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), _p(g_res), dx.data_ptr(), B, H, W, cop, Cx, Cx, KH,
                        _p(ws), ks_n, ks_t, st), scope_name=ctx.bscope, exec_flops=3 * flops, detail="dgrad-h2" + shape),
                        "h2_conv2d_bwd_data_same_fused")
                elif h2_fits(g):
                    w_h2 = _prepped(w, "h2_dgrad", cop, Cx, 1, owner)
                    ga = amax_of(g)
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_same_add_ws(
                        g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), None, _p(g_res), dx.data_ptr(), None, 0, B, H, W, cop, Cx, Cx, KH, 0, 0,
                        ws.data_ptr(), ks_n, ks_t, st), scope_name=ctx.bscope, exec_flops=3 * flops, detail="dgrad-h2" + shape),
                        "h2_conv2d_same dgrad")
                else:
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_same_add_ws(
                        g.data_ptr(), w_x3.data_ptr(), None, _p(g_res), dx.data_ptr(), B, H, W, cop, Cx, Cx, KH, 0, 0, ws.data_ptr(), ks_n,
                        ks_t, st), scope_name=ctx.bscope, exec_flops=6 * flops, detail="dgrad-x3" + shape), "x3_conv2d_same dgrad")
                if not fused:
                    _lib.check(_timed("conv_halo_x3_kernel", 0.0, lambda: lib.dwc_conv2d_bwd_data_ring(
                        g.data_ptr(), w_dg(), w_dg_t(), dx.data_ptr(), B, H, W, Cx, cop, KH, KW, pad, ws.data_ptr(), nws, st),
                        scope_name=ctx.bscope, detail="dgrad-ring" + shape), "conv2d_bwd_data_ring")
                g_res = None                                   # consumed by the kernel's epilogue
            elif half and HALO and RING_FUSED and lib.dwc_bf16_conv2d_bwd_data_same_fused_ok(B, H, W, Cx, cop, KH):
                # interior AND border ring in one launch: the border tiles of the halo kernel fold the ring in as extra MFMAs (r06)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_bwd_data_same_fused(
                    g.data_ptr(), w_dg(), _p(g_res), dx.data_ptr(), B, H, W, Cx, cop, KH, st),
                    scope_name=ctx.bscope, detail="dgrad-halo B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)),
                    "conv2d_bwd_data_same_fused")
                g_res = None                                   # consumed by the kernel's epilogue
            elif half and HALO and lib.dwc_bf16_conv2d_same_halo_ok(B, H, W, cop, Cx, KH):
                # interior on the halo-tiled kernel (zero rule, dgrad weights), the ring stays on the strip GEMMs
                ws = workspace(nws, dev)

                shape = " B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_same_halo_add(
                    g.data_ptr(), w_dg(), None, _p(g_res), dx.data_ptr(), B, H, W, cop, Cx, KH, 0, 0, st),
                    scope_name=ctx.bscope, detail="dgrad-halo" + shape), "conv2d_same_halo dgrad")
                _lib.check(_timed("conv_gemm_kernel", 0.0, lambda: lib.dwc_bf16_conv2d_bwd_data_ring(
                    g.data_ptr(), w_dg(), w_dg_t(), dx.data_ptr(), B, H, W, Cx, cop, KH, KW, pad, ws.data_ptr(), nws, st),
                    scope_name=ctx.bscope, detail="dgrad-ring" + shape), "conv2d_bwd_data_ring")
                g_res = None                                   # consumed by the kernel's epilogue
            else:
                ws = workspace(nws, dev)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data_same", x)(
                    g.data_ptr(), w_dg(), w_dg_t(), dx.data_ptr(), B, H, W, Cx, cop, KH, KW, pad, ws.data_ptr(),
                    ws.numel(), st), scope_name=ctx.bscope, detail="dgrad B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)),
                    "conv2d_bwd_data_same")
        elif (ctx.needs_input_grad[0] and S2DGRAD and stride == 2 and KH == 4 and KW == 4 and pad == 1
              and 4 * B * (H // 32) * (W // 32) * (Cx // 64) >= S2DGRAD_MIN_WGS
              and ((half and HALO and lib.dwc_bf16_conv2d_s2_halo_bwd_data_ok(B, H, W, Cx, cop))
                   or ((not half) and X3 and lib.dwc_x3_conv2d_s2_bwd_data_ok(B, H, W, Cx, cop)))):
            # stride-2 4x4 layers: the interior (all H x W pixels of dx) as four output-parity classes of 2x2-tap halo convolutions
            # over dY, then the border ring of the padded image as eight thin GEMM strips + band fold (reflect-pad adjoint)
            fused = bool(RING_FUSED and (half or h2_fits(g)))
            # (the fused fp32 form reads neither the im2col data-gradient layout nor the padded scratch image)
            w_dg = _prepped(w, "dgrad", cop, Cx, 2, owner, half) if half or not fused else None
            dx = empty_cl(B, Cx, H, W, dev, dt)
            flops = 2.0 * rows * Cout * Cin * KH * KW
            dxp = None if fused else workspace(B * (H + 2) * (W + 2) * Cx * (2 if half else 4), dev)
            shape = " B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)
            if half and fused:     # interior + border ring in one launch (r06: conv_halo16_bf16.inc RING, S2 == 2)
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_s2_halo_bwd_data_fused(
                    g.data_ptr(), w_dg.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, st), scope_name=ctx.bscope,
                    detail="dgrad-s2halo" + shape), "conv2d_s2_halo_bwd_data_fused")
            elif half:
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_s2_halo_bwd_data(
                    g.data_ptr(), w_dg.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, st), scope_name=ctx.bscope,
                    detail="dgrad-s2halo" + shape), "conv2d_s2_halo_bwd_data")
                _lib.check(_timed("conv_gemm_kernel", 0.0, lambda: lib.dwc_bf16_conv2d_bwd_data_s2_ring(
                    g.data_ptr(), w_dg.data_ptr(), dxp.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, st), scope_name=ctx.bscope,
                    detail="dgrad-ring" + shape), "conv2d_bwd_data_s2_ring")
            else:
                if fused:          # interior + border ring in one launch (r06: pre-summed patch rows / columns, conv_halo_x3_kernel RING)
                    w_h2 = _prepped(w, "h2_dgrad", cop, Cx, 1, owner)
                    ga = amax_of(g)
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_s2_bwd_data_fused(
                        g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, Cx, st), scope_name=ctx.bscope,
                        exec_flops=3 * flops, detail="dgrad-h2s2" + shape), "h2_conv2d_s2_bwd_data_fused")
                elif h2_fits(g):
                    w_h2 = _prepped(w, "h2_dgrad", cop, Cx, 1, owner)
                    ga = amax_of(g)
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_s2_bwd_data(
                        g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, Cx, st), scope_name=ctx.bscope,
                        exec_flops=3 * flops, detail="dgrad-h2s2" + shape), "h2_conv2d_s2_bwd_data")
                else:
                    w_x3 = _prepped(w, "x3_dgrad", cop, Cx, 1, owner)
                    _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_s2_bwd_data(
                        g.data_ptr(), w_x3.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, Cx, st), scope_name=ctx.bscope,
                        exec_flops=6 * flops, detail="dgrad-x3s2" + shape), "x3_conv2d_s2_bwd_data")
                if not fused:
                    _lib.check(_timed("conv_halo_x3_kernel", 0.0, lambda: lib.dwc_conv2d_bwd_data_s2_ring(
                        g.data_ptr(), w_dg.data_ptr(), dxp.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, st), scope_name=ctx.bscope,
                        detail="dgrad-ring" + shape), "conv2d_bwd_data_s2_ring")
        elif ctx.needs_input_grad[0]:
            w_dg = _prepped(w, "dgrad", cop, Cx, stride, owner, half)
            dx = empty_cl(B, Cx, H, W, dev, dt)
            flops = 2.0 * rows * Cout * Cin * KH * KW
            # scratch arena: [gradient of the padded image (folded back below)] [split-K partials]
            esz = 2 if half else 4
            pad_bytes = 0 if pad == 0 else (B * (H + 2 * pad) * (W + 2 * pad) * Cx * esz + 255) // 256 * 256
            nws = _fn(lib, "conv2d_bwd_data_ws_bytes", x)(B, H, W, Cx, cop, KH, KW, stride, pad)
            base = workspace(pad_bytes + nws, dev).data_ptr() if pad_bytes + nws else 0
            target = dx.data_ptr() if pad == 0 else base
            wsp = base + pad_bytes if nws else None
            if pad > 0 and DGRAD_FOLD and min(H, W) >= 2 * pad + 2 and Cx % (8 if half else 4) == 0:
                # GEMM + reflect-pad adjoint in one call: interior straight into dx, only the border ring through the padded scratch
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data_fold", x)(
                    g.data_ptr(), w_dg.data_ptr(), target, dx.data_ptr(), B, H, W, Cx, cop, KH, KW, stride, pad, wsp, nws, st),
                    scope_name=ctx.bscope, detail="dgrad B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_bwd_data_fold")
            else:
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data", x)(
                    g.data_ptr(), w_dg.data_ptr(), target, B, H, W, Cx, cop, KH, KW, stride, pad, wsp, nws, st),
                    scope_name=ctx.bscope, detail="dgrad B%d %dx%d %d>%d k%d s%d" % (B, H, W, Cx, cop, KH, stride)), "conv2d_bwd_data")
                if pad > 0:
                    _lib.check(_fn(lib, "reflect_pad_adjoint", x)(target, dx.data_ptr(), B, H, W, Cx, pad, st), "reflect_pad_adjoint")
        if g_res is not None and dx is not None:               # no fused form on this path: the plain sum
            dx = dx + g_res.to(dx.dtype)
        return dx, dw, db, None, None, None, None, None, None, None


def conv2d(x, w, b, stride, pad, act="none", bias_grad=True, owner=None, token=None):
    """Reflect-padded convolution + bias + activation.  Returns Cout channels (a channel
    slice of the 4-aligned buffer when Cout is not a multiple of 4).  ``bias_grad=False``: the caller feeds the
    result to an instance norm, whose mean subtraction makes the bias gradient identically zero -- it is returned
    as zeros instead of being reduced from dY."""
    if not bias_grad and ZERO_GRAD_BY_FLAG and isinstance(b, torch.nn.Parameter):
        b._dwc_zero_grad = True            # (see _Conv2d.backward: no gradient tensor for it; FusedAdam / the DP reducer know the flag)
    y = _Conv2d.apply(x, w, b, int(stride), int(pad), ACT[act], bool(bias_grad), owner, token, not torch.is_grad_enabled())
    return y if y.shape[1] == w.shape[0] else y[:, :w.shape[0]]


def conv2d_padded(x, w, b, stride, pad, act="none"):
    """As conv2d, but returns the 4-aligned channel buffer itself."""
    return _Conv2d.apply(x, w, b, int(stride), int(pad), ACT[act], True, None)


class _HeadsConvWide(torch.autograd.Function):
    """The fused image heads (tanh x3 + sigmoid, P = 4 planes; 8 with four zero planes on the bf16 path) as a "wide"
    convolution: px = 32/P horizontally adjacent output pixels x P planes = 32 output channels of a KHx(KW+px-1),
    stride-(1,px) filter bank whose p-th copy is the real filter shifted right by p taps.  The product then fills a
    32-wide MFMA tile (2x zero work instead of 8x).  The [B,H,W/px,32] result IS the NHWC-P image."""

    @staticmethod
    def forward(ctx, x, w4, b4, owner=None):
        _require_device(x)
        lib = _lib.load()
        x = cl(x)
        half = x.dtype == BF16
        B, C, H, W = x.shape
        P, ci, KH, KW = w4.shape
        px = 32 // P
        assert P == image_planes(x.dtype) and ci == C and W % px == 0
        pad = KH // 2
        w_prep = _prepped(w4, "heads_wide", 32, C, 1, owner, half)
        bias = b4.detach().repeat(px).contiguous()
        y = empty_cl(B, P, H, W, x.device, x.dtype)
        st = _stream()
        flops = 2.0 * B * H * W * 4 * C * KH * KW
        if half and NARROW and lib.dwc_bf16_conv2d_narrow_ok(B, H, W, C, H, W // px, KH, KW + px - 1):
            # patch staged once per 16x32-pixel block, taps dealt to the waves (csrc/conv_narrow_bf16.hip)
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_narrow(
                x.data_ptr(), _prepped(w4, "heads_narrow", 32, C, 1, owner, True).data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W,
                C, H, W // px, KH, KW + px - 1, -pad, -pad, ACT["heads8"], 1, st),
                detail="fwd-heads-narrow B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_narrow")
        elif ((not half) and X3 and NARROW_X3 and P == 4
              and lib.dwc_x3_conv2d_narrow_ok(B, H, W, C, H, W // px, KH, KW + px - 1)):
            # fp32: the same form as split products (csrc/conv_narrow_x3.hip): patch split once per 16-channel slab, taps dealt to the waves
            w_frag = _prepped(w4, "heads_narrow_x3", 32, C, 1, owner)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_narrow(
                x.data_ptr(), w_frag.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, C, H, W // px, KH, KW + px - 1, -pad, -pad,
                ACT["heads"], 1, st), detail="fwd-heads-nx3 B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH), exec_flops=6 * flops),
                "x3_conv2d_narrow")
        else:
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_fwd_ex", x)(
                x.data_ptr(), w_prep.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, C, 32, KH, KW + px - 1, 1, px, pad, pad,
                ACT["heads8" if P == 8 else "heads"], st), detail="fwd-heads B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)),
                "conv2d_fwd_ex")
        ctx.save_for_backward(x, w4, y)
        ctx.owner = owner
        ctx.bscope = ("bwd:" + SCOPE) if SCOPE else ""
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w4, y = ctx.saved_tensors
        half = x.dtype == BF16
        B, C, H, W = x.shape
        P, _, KH, KW = w4.shape
        px = 32 // P
        pad = KH // 2
        dev = x.device
        st = _stream()
        dy = cl(dy)
        rows = B * H * W
        g = empty_cl(B, P, H, W, dev, x.dtype)
        db = torch.empty(P, dtype=torch.float32, device=dev)
        ws = workspace(lib.dwc_act_bwd_bias_ws_bytes(rows, P), dev)
        _lib.check(_fn(lib, "act_bwd_bias", x)(dy.data_ptr(), y.data_ptr(), g.data_ptr(), db.data_ptr(), rows, P,
                                               ACT["heads8" if P == 8 else "heads"], ws.data_ptr(), ws.numel(), st), "act_bwd_bias")
        dx = dw = None
        flops = 2.0 * rows * 4 * C * KH * KW
        if (ctx.needs_input_grad[0] and half and NARROW and STEM and P == 8 and C == 64 and KH == 7 and KW == 7
                and lib.dwc_bf16_conv2d_stem_ok(B, H, W, H + 2 * pad, W + 2 * pad, KH, 0)):
            # data gradient = 7x7 convolution of the 8-plane gradient image with the rotated filter on the padded grid (zero
            # rule), folded by the reflect adjoint: the stem kernel of csrc/conv_narrow_bf16.hip
            w_st = _prepped(w4, "stem_steps_dgrad", 64, P, 1, ctx.owner, True)
            dx = empty_cl(B, C, H, W, dev, x.dtype)
            base = workspace(B * (H + 2 * pad) * (W + 2 * pad) * C * 2, dev).data_ptr()
            if DGRAD_FOLD and min(H, W) >= 2 * pad + 2:
                # interior of the padded gradient image straight into dx, only its border ring through the scratch image + band fold
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_stem_crop(
                    g.data_ptr(), w_st.data_ptr(), None, base, dx.data_ptr(), pad, B, H, W, H + 2 * pad, W + 2 * pad, KH, -(KH - 1), 0, 0,
                    st), scope_name=ctx.bscope, detail="dgrad-heads-stem B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_stem dgrad")
                _lib.check(lib.dwc_bf16_reflect_pad_adjoint_band(base, dx.data_ptr(), B, H, W, C, pad, st), "reflect_pad_adjoint_band")
            else:
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_stem(
                    g.data_ptr(), w_st.data_ptr(), None, base, B, H, W, H + 2 * pad, W + 2 * pad, KH, -(KH - 1), 0, 0, st),
                    scope_name=ctx.bscope, detail="dgrad-heads-stem B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_stem dgrad")
                _lib.check(lib.dwc_bf16_reflect_pad_adjoint(base, dx.data_ptr(), B, H, W, C, pad, st), "reflect_pad_adjoint")
        elif (ctx.needs_input_grad[0] and (not half) and X3 and NARROW_X3 and P == 4 and C == 64 and KH == 7 and KW == 7
              and DGRAD_FOLD and min(H, W) >= 2 * pad + 2 and lib.dwc_x3_conv2d_stem_ok(B, H, W, H + 2 * pad, W + 2 * pad, KH, 0)):
            # fp32: the same as split products (conv_stem_x3_kernel): interior of the padded gradient image straight into dx, its
            # border ring through the scratch image + band fold
            w_st = _prepped(w4, "stem_steps_dgrad_x3", 64, P, 1, ctx.owner)
            dx = empty_cl(B, C, H, W, dev, x.dtype)
            base = workspace(B * (H + 2 * pad) * (W + 2 * pad) * C * 4, dev).data_ptr()
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_stem_crop(
                g.data_ptr(), w_st.data_ptr(), None, base, dx.data_ptr(), pad, B, H, W, H + 2 * pad, W + 2 * pad, KH, -(KH - 1), 0, 0,
                st), scope_name=ctx.bscope, exec_flops=6 * flops, detail="dgrad-heads-stem-x3 B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)),
                "x3_conv2d_stem dgrad")
            _lib.check(lib.dwc_reflect_pad_adjoint_band(base, dx.data_ptr(), B, H, W, C, pad, st), "reflect_pad_adjoint_band")
        elif ctx.needs_input_grad[0]:        # data gradient: the ordinary P-channel formulation (N = C columns)
            w_dg = _prepped(w4, "dgrad", P, C, 1, ctx.owner, half)
            dx = empty_cl(B, C, H, W, dev, x.dtype)
            pad_bytes = (B * (H + 2 * pad) * (W + 2 * pad) * C * (2 if half else 4) + 255) // 256 * 256
            nws = _fn(lib, "conv2d_bwd_data_ws_bytes", x)(B, H, W, C, P, KH, KW, 1, pad)
            base = workspace(pad_bytes + nws, dev).data_ptr()
            if DGRAD_FOLD and min(H, W) >= 2 * pad + 2 and C % (8 if half else 4) == 0:
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data_fold", x)(
                    g.data_ptr(), w_dg.data_ptr(), base, dx.data_ptr(), B, H, W, C, P, KH, KW, 1, pad, (base + pad_bytes) if nws else None,
                    nws, st), scope_name=ctx.bscope, detail="dgrad-heads B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_bwd_data_fold")
            else:
                _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data", x)(
                    g.data_ptr(), w_dg.data_ptr(), base, B, H, W, C, P, KH, KW, 1, pad, (base + pad_bytes) if nws else None, nws,
                    st), scope_name=ctx.bscope, detail="dgrad-heads B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_bwd_data")
                _lib.check(_fn(lib, "reflect_pad_adjoint", x)(base, dx.data_ptr(), B, H, W, C, pad, st), "reflect_pad_adjoint")
        if ctx.needs_input_grad[1] and half and NARROW and STEM and P == 8 and C == 64 and KH == 7 and KW == 7:
            dw = torch.empty((P, C, KH, KW), dtype=torch.float32, device=dev)
            ws = workspace(lib.dwc_bf16_conv7_smallk_wgrad_ws_bytes(B, H, W, 1), dev)
            _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: lib.dwc_bf16_conv7_smallk_wgrad(
                g.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, P, 1, ws.data_ptr(), ws.numel(), st), scope_name=ctx.bscope,
                detail="wgrad-heads-small B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv7_smallk_wgrad")
        elif ctx.needs_input_grad[1] and (not half) and X3 and SMALLK_X3 and P == 4 and C == 64 and KH == 7 and KW == 7:
            # fp32 image heads: the same kernel with the roles swapped (A = the gradient image, Bt = x on the padded grid)
            dw = torch.empty((P, C, KH, KW), dtype=torch.float32, device=dev)
            ws = workspace(lib.dwc_x3_conv7_smallk_wgrad_ws_bytes(B, H, W, 1), dev)
            _lib.check(_timed("wgrad_x3_kernel+reduce", flops, lambda: lib.dwc_x3_conv7_smallk_wgrad(
                g.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, P, 1, ws.data_ptr(), ws.numel(), st), scope_name=ctx.bscope,
                detail="wgrad-heads-small-x3 B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH), exec_flops=6 * flops), "x3_conv7_smallk_wgrad")
        elif ctx.needs_input_grad[1]:        # weight gradient of the wide filter bank, folded back onto the real taps
            dwide = torch.empty((32, C, KH, KW + px - 1), dtype=torch.float32, device=dev)
            nws = _fn(lib, "conv2d_bwd_weight_ex_ws_bytes", x)(B, H, W, C, 32, KH, KW + px - 1, 1, px, pad, pad)
            ws = workspace(nws, dev)
            _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: _fn(lib, "conv2d_bwd_weight_ex", x)(
                x.data_ptr(), g.data_ptr(), dwide.data_ptr(), B, H, W, C, 32, KH, KW + px - 1, 1, px, pad, pad, C, 32,
                ws.data_ptr(), ws.numel(), st), scope_name=ctx.bscope,
                detail="wgrad-heads B%d %dx%d %d>%d k%d" % (B, H, W, C, P, KH)), "conv2d_bwd_weight_ex")
            # fold the px shifted copies back onto the real taps: the diagonal view (stride along p = the bank's plus one tap), one sum
            dw = dwide.as_strided((px, P, C, KH, KW), (dwide.stride(0) * P + 1,) + tuple(dwide.stride())).sum(0)
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None


def conv2d_heads(x, w4, b4, owner=None):
    """tanh/sigmoid image heads: [B,C,H,W] features, [P,C,7,7] weights (P = 4 fp32, 8 bf16 with planes 4..7 zero) ->
    NHWC-P image [B,P,H,W]."""
    px = 32 // w4.shape[0]
    if x.shape[3] % px == 0 and w4.shape[2] == w4.shape[3] == 7:
        return _HeadsConvWide.apply(x, w4, b4, owner)
    return _Conv2d.apply(x, w4, b4, 1, w4.shape[2] // 2, ACT["heads8" if w4.shape[0] == 8 else "heads"], True, owner)


class _Conv2dZeroPad(torch.autograd.Function):
    """relu?(conv(zero_pad(x)) + b) with FROZEN weights (the VGG16 of the perceptual loss, reference
    networks.py:639-688): forward and data gradient only."""

    @staticmethod
    def forward(ctx, x, w, b, pad, act):
        _require_device(x)
        if w.requires_grad or (b is not None and b.requires_grad):
            raise NotImplementedError("zero-padded convolutions are built for frozen weights (no weight gradient)")
        lib = _lib.load()
        x = cl(x)
        half = x.dtype == BF16
        B, Cx, H, W = x.shape
        Cout, Cin, KH, KW = w.shape
        if Cin > Cx:
            raise ValueError("input has %d channels, weight expects %d" % (Cx, Cin))
        cop = _padc(Cout, x.dtype)
        # fp32: the split-product halo kernel with the zero rule where the shape allows (r05; rounds 1-4 ran Winograd F(2x2,3x3) here)
        sp = bool((not half) and _x3_use(lib, B, H, W, Cx, cop, KH, KW, 1, pad, free=True))
        halo = bool(half and HALO and KH == KW and 2 * pad == KH - 1 and lib.dwc_bf16_conv2d_same_halo_ok(B, H, W, Cx, cop, KH))
        w_prep = None if sp else _prepped(w, "fwd", cop, Cx, 1, None, half)
        bias = None
        if b is not None:
            bias = b.detach() if cop == Cout else torch.nn.functional.pad(b.detach(), (0, cop - Cout))
        y = empty_cl(B, cop, H, W, x.device, x.dtype)
        st = _stream()
        flops = 2.0 * B * H * W * Cout * Cin * KH * KW
        if sp and h2_fits(x):
            w_h2 = _prepped(w, "h2_fwd", cop, Cx, 1)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 1)
            xa = amax_of(x)
            ya, yep = out_amax(y)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_same_add_ws(
                x.data_ptr(), xa[0], xa[1], w_h2.data_ptr(), _p(bias), None, y.data_ptr(), ya, yep, B, H, W, Cx, cop, cop, KH, act, 0,
                _p(ks_ws), ks_n, ks_t, st), detail="fwd-h2 B%d %dx%d %d>%d k%d zeropad" % (B, H, W, Cx, cop, KH), exec_flops=3 * flops),
                "h2_conv2d_same zeropad")
            set_amax(y, ya, yep)
        elif sp:
            w_x3 = _prepped(w, "x3_fwd", cop, Cx, 1)
            ks_ws, ks_n, ks_t = _x3_ksplit(lib, x.device, B, H, W, Cx, cop, KH, 1)
            _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_same_add_ws(
                x.data_ptr(), w_x3.data_ptr(), _p(bias), None, y.data_ptr(), B, H, W, Cx, cop, cop, KH, act, 0, _p(ks_ws), ks_n, ks_t, st),
                detail="fwd-x3 B%d %dx%d %d>%d k%d zeropad" % (B, H, W, Cx, cop, KH), exec_flops=6 * flops), "x3_conv2d_same zeropad")
        elif halo:               # bf16: the halo-tiled kernel with the zero rule
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_same_halo(
                x.data_ptr(), w_prep.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, KH, act, 0, st),
                detail="fwd-zeropad-halo B%d %dx%d %d>%d k%d" % (B, H, W, Cx, cop, KH)), "conv2d_same_halo zeropad")
        else:
            nws = _fn(lib, "conv2d_fwd_ws_bytes", x)(B, H, W, Cx, cop, KH, KW, 1, pad)
            wsp = workspace(nws, x.device).data_ptr() if nws else None
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_fwd_zeropad", x)(
                x.data_ptr(), w_prep.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cx, cop, KH, KW, 1, pad, act, wsp, nws, st),
                detail="fwd-zeropad B%d %dx%d %d>%d k%d" % (B, H, W, Cx, cop, KH)), "conv2d_fwd_zeropad")
        ctx.save_for_backward(w, y if act != 0 else None)
        ctx.geom = (B, H, W, Cx, cop, KH, KW, pad, act, Cin, Cout)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        w, y = ctx.saved_tensors
        B, H, W, Cx, cop, KH, KW, pad, act, Cin, Cout = ctx.geom
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        dy = cl(dy)
        half = dy.dtype == BF16
        dev, st, rows = dy.device, _stream(), B * H * W
        g = dy
        if act != 0:
            g = empty_cl(B, cop, H, W, dev, dy.dtype)
            ws = workspace(lib.dwc_act_bwd_bias_ws_bytes(rows, cop), dev)
            _lib.check(_fn(lib, "act_bwd_bias", dy)(dy.data_ptr(), y.data_ptr(), g.data_ptr(), None, rows, cop, act, ws.data_ptr(),
                                                    ws.numel(), st), "act_bwd_bias")
        dx = empty_cl(B, Cx, H, W, dev, dy.dtype)
        flops = 2.0 * rows * Cout * Cin * KH * KW
        if (not half) and _x3_use(lib, B, H, W, cop, Cx, KH, KW, 1, pad, free=True):
            # the adjoint of zero padding is a crop: the zero-rule convolution of dY with the rotated filter, no ring at all
            ws, ks_n, ks_t = _x3_ksplit(lib, dev, B, H, W, cop, Cx, KH, 1)
            if h2_fits(g):
                w_h2 = _prepped(w, "h2_dgrad", cop, Cx, 1)
                ga = amax_of(g)
                _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_h2_conv2d_same_add_ws(
                    g.data_ptr(), ga[0], ga[1], w_h2.data_ptr(), None, None, dx.data_ptr(), None, 0, B, H, W, cop, Cx, Cx, KH, 0, 0, _p(ws),
                    ks_n, ks_t, st), detail="dgrad-h2 B%d %dx%d %d>%d k%d zeropad" % (B, H, W, Cx, cop, KH), exec_flops=3 * flops),
                    "h2_conv2d_same zeropad dgrad")
            else:
                w_x3 = _prepped(w, "x3_dgrad", cop, Cx, 1)
                _lib.check(_timed("conv_halo_x3_kernel", flops, lambda: lib.dwc_x3_conv2d_same_add_ws(
                    g.data_ptr(), w_x3.data_ptr(), None, None, dx.data_ptr(), B, H, W, cop, Cx, Cx, KH, 0, 0, _p(ws), ks_n, ks_t, st),
                    detail="dgrad-x3 B%d %dx%d %d>%d k%d zeropad" % (B, H, W, Cx, cop, KH), exec_flops=6 * flops), "x3_conv2d_same zeropad dgrad")
            return dx, None, None, None, None
        w_dg = _prepped(w, "dgrad", cop, Cx, 1, None, half)
        if half and HALO and KH == KW and 2 * pad == KH - 1 and lib.dwc_bf16_conv2d_same_halo_ok(B, H, W, cop, Cx, KH):
            _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_bf16_conv2d_same_halo(
                g.data_ptr(), w_dg.data_ptr(), None, dx.data_ptr(), B, H, W, cop, Cx, KH, 0, 0, st),
                detail="dgrad-zeropad-halo B%d %dx%d %d>%d k%d" % (B, H, W, Cx, cop, KH)), "conv2d_same_halo zeropad dgrad")
            return dx, None, None, None, None
        nws = _fn(lib, "conv2d_bwd_data_zeropad_ws_bytes", dy)(B, H, W, Cx, cop, KH, KW, pad)
        wsp = workspace(nws, dev).data_ptr() if nws else None
        _lib.check(_timed("conv_gemm_kernel", flops, lambda: _fn(lib, "conv2d_bwd_data_zeropad", dy)(
            g.data_ptr(), w_dg.data_ptr(), dx.data_ptr(), B, H, W, Cx, cop, KH, KW, pad, wsp, nws, st),
            detail="dgrad-zeropad B%d %dx%d %d>%d k%d" % (B, H, W, Cx, cop, KH)), "conv2d_bwd_data_zeropad")
        return dx, None, None, None, None


def conv2d_zeropad(x, w, b, pad, act="none"):
    """Zero-padded stride-1 convolution + bias + activation with frozen weights."""
    y = _Conv2dZeroPad.apply(x, w, b, int(pad), ACT[act])
    return y if y.shape[1] == w.shape[0] else y[:, :w.shape[0]]


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_device(x)
        lib = _lib.load()
        x = cl(x)
        B, C, H, W = x.shape
        y = empty_cl(B, C, H // 2, W // 2, x.device, x.dtype)
        _lib.check(_fn(lib, "maxpool2_fwd", x)(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), "maxpool2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dx = empty_cl(B, C, H, W, x.device, x.dtype)
        _lib.check(_fn(lib, "maxpool2_bwd", x)(x.data_ptr(), cl(dy.to(x.dtype)).data_ptr(), dx.data_ptr(), B, H, W, C, _stream()),
                   "maxpool2_bwd")
        return dx


def max_pool2(x):
    """F.max_pool2d(x, kernel_size=2, stride=2) (even H, W; C % 4 == 0)."""
    return _MaxPool2.apply(x)


LINEAR_SMALL = int(os.environ.get("DWC_LINEAR_SMALL", "1"))      # nn.Linear on few rows on csrc/linear_small.hip (0: the 1x1-convolution path)


class _LinearSmall(torch.autograd.Function):
    """y = relu?(x w^T + b) for fp32 activations of a few rows: one launch forward, two backward (csrc/linear_small.hip; r06).  The
    weights are read as nn.Linear stores them: no prepared layout to refresh."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _lib.check(_timed("linear_small_kernel", 2.0 * M * N * K, lambda: lib.dwc_linear_small_fwd(
            x.data_ptr(), w.data_ptr(), _p(b), y.data_ptr(), M, N, K, int(relu), _stream()),
            detail="fwd-lin B%d 1x1 %d>%d k1 s1" % (M, K, N)), "linear_small_fwd")
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, dtype=torch.float32, device=x.device) if (ctx.has_b and ctx.needs_input_grad[2] and dw is not None) else None
        _lib.check(_timed("linear_small_kernel", 2.0 * M * N * K * ((dx is not None) + (dw is not None)), lambda: lib.dwc_linear_small_bwd(
            dy.data_ptr(), _p(y), x.data_ptr(), w.data_ptr(), _p(dx), _p(dw), _p(db), M, N, K, _stream()),
            scope_name=("bwd:" + SCOPE) if SCOPE else "", detail="bwd-lin B%d 1x1 %d>%d k1 s1" % (M, K, N)), "linear_small_bwd")
        if ctx.has_b and ctx.needs_input_grad[2] and db is None:        # (bias gradient alone: column sums of the masked dy)
            g = dy if y is None else dy * (y > 0)
            db = g.sum(0)
        return dx, dw, db, None


def linear(x, w, b, act="none", owner=None):
    """nn.Linear (+ReLU) (reference networks.py:587-634): on csrc/linear_small.hip where the shape fits (fp32, widths multiples of 16,
    at most 4096 rows), else as a 1x1 convolution over a 1x1 image (input width a power of two >= 4)."""
    if (LINEAR_SMALL and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and act in ("none", "relu", None)
            and _lib.load().dwc_linear_small_ok(x.shape[0], w.shape[0], x.shape[1])):
        return _LinearSmall.apply(x, w, b, act == "relu")
    y = conv2d(x.reshape(x.shape[0], x.shape[1], 1, 1), w.reshape(w.shape[0], w.shape[1], 1, 1), b, 1, 0, act, owner=owner or w)
    return y.reshape(x.shape[0], -1)


# --------------------------------------------------------------------------------------
# Dense products with widths that are not powers of two (the text encoder: 364 / 600 / 1200 / 2400) on the im2col GEMM kernels.
# A row of K = taps * c values (c = the largest power of two <= 32 dividing K, >= 4) is read as an NHWC image of `taps` pixels
# x c channels and contracted by ONE filter of `taps` taps: the kernels' gather wants a power-of-two channel count, not a
# power-of-two K.  No copy of the activations; the weight views are re-laid-out once per optimiser step (prepared-weight cache).
# fp32 in / out, inner products as exact split products on the bf16 matrix cores like every other fp32 layer (dwc_x3_gemm_mode).
# --------------------------------------------------------------------------------------
def _pow2_div(k, cap=32):
    c = 1
    while c < cap and k % (2 * c) == 0:
        c *= 2
    return c


def gemm_ok(K, N):
    """Shapes `_gemm_nt` / `_gemm_nn` / `_gemm_tn` take: both widths multiples of 4 whose tap counts the gather can enumerate."""
    return K % 4 == 0 and N % 4 == 0 and K // _pow2_div(K) <= 256 and N // _pow2_div(N) <= 256


def _gemm_nt(x, w, bias, owner, out=None):
    """x:[M,K] @ w:[N,K]^T (+ bias:[N]) -> [M,N] (into `out` when given).  `owner`: the parameter(s) `w` derives from
    (prepared-layout cache key)."""
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[0]
    c = _pow2_div(K)
    taps = K // c
    w4 = w.detach().reshape(N, taps, c).permute(0, 2, 1).unsqueeze(2)                 # OIHW [N, c, 1, taps]
    wp = _prepped(w4, "fwd", N, c, 1, owner)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device) if out is None else out
    assert y.shape == (M, N) and y.is_contiguous() and y.dtype == torch.float32
    nws = lib.dwc_conv2d_fwd_ws_bytes(M, 1, taps, c, N, 1, taps, 1, 0)
    wsp = workspace(nws, x.device).data_ptr() if nws else None
    flops = 2.0 * M * N * K
    _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_conv2d_fwd(
        x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), M, 1, taps, c, N, 1, taps, 1, 0, 0, wsp, nws, _stream()),
        detail="fwd B%d 1x%d %d>%d k%d s1" % (M, taps, c, N, taps)), "gemm_nt")
    return y


def _gemm_nn(dy, w, owner):
    """dy:[M,N] @ w:[N,K] -> [M,K] (the input gradient of `_gemm_nt`): the same kernel over dy's row with w^T as the filter."""
    lib = _lib.load()
    M, N = dy.shape
    K = w.shape[1]
    c = _pow2_div(N)
    taps = N // c
    # taps along the filter's HEIGHT here: the cache entry can never collide with the forward layout of a square weight
    wt4 = w.detach().t().reshape(K, taps, c).permute(0, 2, 1).unsqueeze(3)              # OIHW [K, c, taps, 1]
    wp = _prepped(wt4, "fwd", K, c, 1, owner)
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    nws = lib.dwc_conv2d_fwd_ws_bytes(M, taps, 1, c, K, taps, 1, 1, 0)
    wsp = workspace(nws, dy.device).data_ptr() if nws else None
    flops = 2.0 * M * N * K
    _lib.check(_timed("conv_gemm_kernel", flops, lambda: lib.dwc_conv2d_fwd(
        dy.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), M, taps, 1, c, K, taps, 1, 1, 0, 0, wsp, nws, _stream()),
        detail="dgrad B%d %dx1 %d>%d k%d s1" % (M, taps, c, K, taps)), "gemm_nn")
    return dx


def _gemm_tn(dy, x, out=None):
    """dy:[M,N]^T @ x:[M,K] -> [N,K] (the weight gradient of `_gemm_nt`; into `out` when given): the weight-gradient kernel of the
    same one-filter layer."""
    lib = _lib.load()
    M, N = dy.shape
    K = x.shape[1]
    c = _pow2_div(K)
    taps = K // c
    dw4 = torch.empty((N, c, 1, taps), dtype=torch.float32, device=x.device)
    nws = lib.dwc_conv2d_bwd_weight_ws_bytes(M, 1, taps, c, N, 1, taps, 1, 0)
    ws = workspace(nws, x.device)
    flops = 2.0 * M * N * K
    _lib.check(_timed("conv_wgrad_kernel+reduce", flops, lambda: lib.dwc_conv2d_bwd_weight(
        x.data_ptr(), dy.data_ptr(), dw4.data_ptr(), M, 1, taps, c, N, 1, taps, 1, 0, c, N, ws.data_ptr(), ws.numel(), _stream()),
        detail="wgrad B%d 1x%d %d>%d k%d s1" % (M, taps, c, N, taps)), "gemm_tn")
    if out is None:
        return dw4.squeeze(2).permute(0, 2, 1).reshape(N, K)
    out.view(N, taps, c).copy_(dw4.squeeze(2).permute(0, 2, 1))
    return out


class _LinearAny(torch.autograd.Function):
    """nn.Linear for fp32 x:[M,K], w:[N,K] with K and N multiples of 4, no power-of-two requirement (reference
    networks_v2.py:204-205: the 2 x num_class text heads read a 2400-wide feature row)."""

    @staticmethod
    def forward(ctx, x, w, b, owner):
        _require_device(x)
        if x.dtype != torch.float32 or not gemm_ok(x.shape[1], w.shape[0]):
            raise ValueError("linear_any: fp32, widths multiples of 4 (got %s x %s)" % (tuple(x.shape), tuple(w.shape)))
        x = x.contiguous()
        ctx.owner = owner if owner is not None else w
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return _gemm_nt(x, w, None if b is None else b.detach().contiguous(), ctx.owner)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _gemm_nn(dy, w, ctx.owner) if ctx.needs_input_grad[0] else None
        dw = _gemm_tn(dy, x) if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


def linear_any(x, w, b, owner=None):
    return _LinearAny.apply(x, w, b, owner)


# --------------------------------------------------------------------------------------
# text encoder: one bidirectional LSTM layer over padded sequences with per-sample lengths
# --------------------------------------------------------------------------------------
LSTM_GEMM = int(os.environ.get("DWC_LSTM_GEMM", "1"))     # 0: the layer's dense products through torch (rocBLAS / hipBLASLt)
_LSTM_STATUS = {}            # device index -> (persistent int32 device word, pinned host copy, event of the last copy or None)
LSTM_SEQ_MAX_WORKGROUPS = 0  # > 0: cap on the persistent launches' grid (hipdwc.dp sets it while collectives share the CUs)


def _lstm_status(device):
    """The sticky hand-off status word of the persistent LSTM launches on `device` (dwc_lstm_seq_*: bit 0 forward, bit 1
    backward timed out).  It is NOT in the scratch arena: no later op overwrites it."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    ent = _LSTM_STATUS.get(key)
    if ent is None:
        ent = [torch.zeros(1, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32).pin_memory(), None]
        _LSTM_STATUS[key] = ent
    return ent


def lstm_status_poll(device, wait=False):
    """Check the persistent LSTM launches' status word without stalling the stream: each call looks at the copy the PREVIOUS
    call started (if it has landed, or `wait`) and starts a new one.  Raises when a hand-off between workgroups timed out --
    the affected results were overwritten with NaN by the kernel, so nothing silently wrong was consumed; the run must stop
    (DWC_LSTM_SEQ=0 selects the per-step kernels, which have no cross-workgroup hand-off).  Solver calls this once per step."""
    if not _LSTM_STATUS:
        return
    ent = _lstm_status(device)
    if ent[2] is not None and (wait or ent[2].query()):
        if wait:
            ent[2].synchronize()
        if int(ent[1][0]) != 0:
            raise _lib.HipKernelError(
                "persistent LSTM kernel: a workgroup hand-off timed out (status %d: not all workgroups of the launch were resident); "
                "its results were overwritten with NaN.  Re-run with DWC_LSTM_SEQ=0 (per-step kernels)." % int(ent[1][0]))
        ent[2] = None
    if ent[2] is None:
        ent[1].copy_(ent[0], non_blocking=True)
        ent[2] = torch.cuda.Event()
        ent[2].record()


class _LSTMBidir(torch.autograd.Function):
    """Both directions of one nn.LSTM layer on a packed batch (reference networks_v2.py:226-233), zero initial state.
    x:[T,B,I]; lens:[B] int32 on the device (sample b is active at steps t < lens[b]); w_ih:[2,4H,I], w_hh:[2,4H,H],
    b_ih, b_hh:[2,4H] (direction 0 forward, 1 reverse).  Returns out (h_t) and c, both [2,T,B,H] with zeros at the
    inactive slots.  The input projection and the four weight/input gradient products are library GEMMs; the T sequential
    steps run in dwc_lstm_fwd / dwc_lstm_bwd."""

    @staticmethod
    def forward(ctx, x, lens, w_ih, w_hh, b_ih, b_hh, owners=None):
        _require_device(x)
        lib = _lib.load()
        T, B, I = x.shape
        H = w_hh.shape[2]
        dev = x.device
        X = x.reshape(T * B, I)
        # owners: the two direction parameters weight_ih is stacked from (cache key of the prepared layouts; without them the
        # dense products stay with torch: the stacked tensor is a cached buffer whose address may be reused)
        ctx.hip_gemm = bool(LSTM_GEMM and owners is not None and gemm_ok(I, 4 * H) and gemm_ok(H, 4 * H))
        ctx.owners = owners
        if ctx.hip_gemm:
            bsum = b_ih + b_hh
            X = X.contiguous()
            xproj = torch.empty((2, T * B, 4 * H), dtype=torch.float32, device=dev)                               # [2,TB,4H]
            for d in range(2):
                _gemm_nt(X, w_ih[d], bsum[d], owners[d], out=xproj[d])
        else:
            xproj = torch.baddbmm((b_ih + b_hh).unsqueeze(1), X.unsqueeze(0).expand(2, -1, -1), w_ih.transpose(1, 2))   # [2,TB,4H]
        w_hh_c = w_hh.contiguous()
        out = torch.empty((2, T, B, H), dtype=torch.float32, device=dev)
        c = torch.empty((2, T, B, H), dtype=torch.float32, device=dev)
        gates = torch.empty((2, T, B, 4 * H), dtype=torch.float32, device=dev)
        rc = -1
        if LSTM_SEQ:
            # all T steps in ONE persistent launch (csrc/lstm.hip lstm_seq_fwd); shapes it does not take fall to the step kernels
            ws = workspace(lib.dwc_lstm_seq_ws_bytes(B, 2), dev)
            rc = lib.dwc_lstm_seq_fwd(xproj.data_ptr(), w_hh_c.data_ptr(), lens.data_ptr(), out.data_ptr(), c.data_ptr(),
                                      gates.data_ptr(), T, B, H, 2, ws.data_ptr(), ws.numel(), _lstm_status(dev)[0].data_ptr(),
                                      LSTM_SEQ_MAX_WORKGROUPS, _stream())
            if rc not in (0, _lib.EINVAL):
                _lib.check(rc, "lstm_seq_fwd")
        if rc != 0:
            _lib.check(lib.dwc_lstm_fwd(xproj.data_ptr(), w_hh_c.data_ptr(), lens.data_ptr(), out.data_ptr(), c.data_ptr(),
                                        gates.data_ptr(), T, B, H, 2, _stream()), "lstm_fwd")
        ctx.save_for_backward(X, lens, w_ih, w_hh_c, out, c, gates)
        ctx.shape = (T, B, I, H)
        return out, c

    @staticmethod
    def backward(ctx, d_out, d_c):
        lib = _lib.load()
        X, lens, w_ih, w_hh, out, c, gates = ctx.saved_tensors
        T, B, I, H = ctx.shape
        dev = X.device
        d_out = None if d_out is None else d_out.contiguous()
        d_c = None if d_c is None else d_c.contiguous()
        w_hh_t = w_hh.transpose(1, 2).contiguous()
        dgates = torch.empty((2, T, B, 4 * H), dtype=torch.float32, device=dev)
        carry = torch.empty((2, B, H), dtype=torch.float32, device=dev)
        rc = -1
        if LSTM_SEQ:
            ws = workspace(lib.dwc_lstm_seq_ws_bytes(B, 2), dev)
            rc = lib.dwc_lstm_seq_bwd(_p(d_out), _p(d_c), w_hh_t.data_ptr(), lens.data_ptr(), c.data_ptr(), gates.data_ptr(),
                                      dgates.data_ptr(), T, B, H, 2, ws.data_ptr(), ws.numel(), _lstm_status(dev)[0].data_ptr(),
                                      LSTM_SEQ_MAX_WORKGROUPS, _stream())
            if rc not in (0, _lib.EINVAL):
                _lib.check(rc, "lstm_seq_bwd")
        if rc != 0:
            _lib.check(lib.dwc_lstm_bwd(_p(d_out), _p(d_c), w_hh_t.data_ptr(), lens.data_ptr(), c.data_ptr(), gates.data_ptr(),
                                        dgates.data_ptr(), carry.data_ptr(), T, B, H, 2, _stream()), "lstm_bwd")
        dG = dgates.view(2, T * B, 4 * H)
        dGt = dG.transpose(1, 2)
        hip = ctx.hip_gemm
        dw_ih = None
        if ctx.needs_input_grad[2]:                                                             # [2,4H,I]
            if hip:
                dw_ih = torch.empty((2, 4 * H, I), dtype=torch.float32, device=dev)
                for d in range(2):
                    _gemm_tn(dG[d], X, out=dw_ih[d])
            else:
                dw_ih = torch.matmul(dGt, X)
        db = dG.sum(1) if (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]) else None
        dw_hh = None
        if ctx.needs_input_grad[3]:
            # state entering step t: the previous slot in processing order (zeros at the ends and past each length)
            hprev = torch.zeros((2, T, B, H), dtype=torch.float32, device=dev)
            hprev[0, 1:] = out[0, :-1]
            hprev[1, :-1] = out[1, 1:]
            hp = hprev.view(2, T * B, H)
            if hip:
                dw_hh = torch.empty((2, 4 * H, H), dtype=torch.float32, device=dev)
                for d in range(2):
                    _gemm_tn(dG[d], hp[d], out=dw_hh[d])
            else:
                dw_hh = torch.bmm(dGt, hp)                                                                     # [2,4H,H]
        dx = None
        if ctx.needs_input_grad[0]:
            if hip:
                dx = (_gemm_nn(dG[0], w_ih[0], ctx.owners[0]) + _gemm_nn(dG[1], w_ih[1], ctx.owners[1])).view(T, B, I)
            else:
                dx = torch.bmm(dG, w_ih).sum(0).view(T, B, I)
        return dx, None, dw_ih, dw_hh, db, db, None


def lstm_bidir(x, lens, w_ih, w_hh, b_ih, b_hh, owners=None):
    """``owners``: (weight_ih of direction 0, of direction 1) -- the parameters ``w_ih`` was stacked from; given, the layer's
    dense products (input projection, its input and weight gradients, the recurrent weight gradient) run on the HIP GEMM kernels."""
    return _LSTMBidir.apply(x.contiguous(), lens, w_ih, w_hh, b_ih, b_hh, owners)


# --------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------
class _InstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, eps, token=None):
        _require_device(x)
        ctx.token = token if residual is not None else None
        lib = _lib.load()
        x = cl(x)
        B, C, H, W = x.shape
        dev = x.device
        if gamma is not None:
            gamma, beta = gamma.float().contiguous(), beta.float().contiguous()
        if residual is not None:
            residual = cl(residual.to(x.dtype))
        y = empty_cl(B, C, H, W, dev, x.dtype)
        mean = torch.empty(B * C, dtype=torch.float32, device=dev)
        rstd = torch.empty(B * C, dtype=torch.float32, device=dev)
        ws = workspace(lib.dwc_instnorm_ws_bytes(B, H * W, C), dev)
        ya, yep = out_amax(y)
        if ya is not None:
            _lib.check(lib.dwc_instnorm_fwd_amax(x.data_ptr(), _p(gamma), _p(beta), _p(residual), y.data_ptr(), mean.data_ptr(),
                                                 rstd.data_ptr(), B, H * W, C, eps, int(relu), ws.data_ptr(), ws.numel(),
                                                 ya, yep, _stream()), "instnorm_fwd")
            set_amax(y, ya, yep)
        else:
            _lib.check(_fn(lib, "instnorm_fwd", x)(x.data_ptr(), _p(gamma), _p(beta), _p(residual), y.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), B, H * W, C, eps, int(relu), ws.data_ptr(), ws.numel(),
                                                   _stream()), "instnorm_fwd")
        _hbm("instnorm_fwd", x.numel() * x.element_size() * (3 if residual is not None else 2))
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.relu = int(relu)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dy = cl(dy)
        B, C, H, W = x.shape
        dev = x.device
        dx = empty_cl(B, C, H, W, dev, x.dtype)
        dgamma = dbeta = None
        if gamma is not None:
            dgamma = torch.empty(B * C, dtype=torch.float32, device=dev)
            dbeta = torch.empty(B * C, dtype=torch.float32, device=dev)
        ws = workspace(lib.dwc_instnorm_ws_bytes(B, H * W, C), dev)
        da, dep = out_amax(dx)
        if da is not None:
            _lib.check(lib.dwc_instnorm_bwd_amax(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma), _p(beta),
                                                 dx.data_ptr(), _p(dgamma), _p(dbeta), B, H * W, C, ctx.relu, ws.data_ptr(),
                                                 ws.numel(), da, dep, _stream()), "instnorm_bwd")
            set_amax(dx, da, dep)
        else:
            _lib.check(_fn(lib, "instnorm_bwd", x)(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma), _p(beta),
                                                   dx.data_ptr(), _p(dgamma), _p(dbeta), B, H * W, C, ctx.relu, ws.data_ptr(),
                                                   ws.numel(), _stream()), "instnorm_bwd")
        _hbm("instnorm_bwd", x.numel() * x.element_size() * 3)
        if ctx.token is not None:                 # the first convolution of the block adds it in its data-gradient epilogue
            ctx.token.g = dy
            return dx, dgamma, dbeta, None, None, None, None
        return dx, dgamma, dbeta, (dy if ctx.has_res else None), None, None, None


def instance_norm(x, gamma=None, beta=None, residual=None, relu=False, eps=1e-5, token=None):
    """IN / AdaIN (+ReLU) (+residual add).  gamma/beta: flat [B*C] per-sample scale/shift or None.  ``token``: see
    ResGradToken (the residual's gradient is handed to the block's first convolution instead of being returned)."""
    return _InstNorm.apply(x, gamma, beta, residual, bool(relu), float(eps), token)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, relu, eps):
        _require_device(x)
        lib = _lib.load()
        x = cl(x)
        B, C, H, W = x.shape
        dev = x.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = empty_cl(B, C, H, W, dev, x.dtype)
        mean = torch.empty(B, dtype=torch.float32, device=dev)
        inv = torch.empty(B, dtype=torch.float32, device=dev)
        ws = workspace(lib.dwc_layernorm_ws_bytes(B, H * W, C), dev)
        ya, yep = out_amax(y)
        if ya is not None:
            _lib.check(lib.dwc_layernorm_fwd_amax(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                  B, H * W, C, eps, int(relu), ws.data_ptr(), ws.numel(), ya, yep, _stream()),
                       "layernorm_fwd")
            set_amax(y, ya, yep)
        else:
            _lib.check(_fn(lib, "layernorm_fwd", x)(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                    inv.data_ptr(), B, H * W, C, eps, int(relu), ws.data_ptr(), ws.numel(), _stream()),
                       "layernorm_fwd")
        _hbm("layernorm_fwd", x.numel() * x.element_size() * 2)
        ctx.save_for_backward(x, mean, inv, g, b)
        ctx.relu, ctx.eps = int(relu), eps
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mean, inv, g, b = ctx.saved_tensors
        dy = cl(dy)
        B, C, H, W = x.shape
        dev = x.device
        dx = empty_cl(B, C, H, W, dev, x.dtype)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = workspace(lib.dwc_layernorm_ws_bytes(B, H * W, C), dev)
        da, dep = out_amax(dx)
        if da is not None:
            _lib.check(lib.dwc_layernorm_bwd_amax(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(),
                                                  b.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), B, H * W, C,
                                                  ctx.eps, ctx.relu, ws.data_ptr(), ws.numel(), da, dep, _stream()), "layernorm_bwd")
            set_amax(dx, da, dep)
        else:
            _lib.check(_fn(lib, "layernorm_bwd", x)(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(),
                                                    b.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), B, H * W, C,
                                                    ctx.eps, ctx.relu, ws.data_ptr(), ws.numel(), _stream()), "layernorm_bwd")
        _hbm("layernorm_bwd", x.numel() * x.element_size() * 3)
        return dx, dgamma, dbeta, None, None


def layer_norm_munit(x, gamma, beta, relu=False, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, bool(relu), float(eps))


# --------------------------------------------------------------------------------------
# resampling
# --------------------------------------------------------------------------------------
class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, up):
        _require_device(x)
        lib = _lib.load()
        x = cl(x)
        B, C, H, W = x.shape
        ctx.shape, ctx.up = (B, C, H, W), up
        if up:
            y = empty_cl(B, C, 2 * H, 2 * W, x.device, x.dtype)
            _lib.check(_fn(lib, "upsample2x_fwd", x)(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), "upsample2x_fwd")
        else:
            y = empty_cl(B, C, H // 2, W // 2, x.device, x.dtype)
            _lib.check(_fn(lib, "avgpool2_fwd", x)(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), "avgpool2_fwd")
        _hbm("upsample2x_fwd" if up else "avgpool2_fwd", (x.numel() + y.numel()) * x.element_size())
        return pass_amax(x, y)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, C, H, W = ctx.shape
        dy = cl(dy)
        dx = empty_cl(B, C, H, W, dy.device, dy.dtype)
        if ctx.up:
            _lib.check(_fn(lib, "upsample2x_bwd", dy)(dy.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()), "upsample2x_bwd")
        else:
            _lib.check(_fn(lib, "avgpool2_bwd", dy)(dy.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()), "avgpool2_bwd")
        _hbm("upsample2x_bwd" if ctx.up else "avgpool2_bwd", (dx.numel() + dy.numel()) * dy.element_size())
        return dx, None


def upsample2x(x):
    return _Resample.apply(x, True)


def downsample_half(x):
    return _Resample.apply(x, False)


# --------------------------------------------------------------------------------------
# image boundary / blend / L1
# --------------------------------------------------------------------------------------
class _Pack4(torch.autograd.Function):
    """[B,3,H,W] fp32 (any strides) -> internal image buffer, channels-last with zero padding planes:
    NHWC4 fp32 [B,4,H,W], or NHWC8 bf16 [B,8,H,W] on the bf16 path."""

    @staticmethod
    def forward(ctx, x, half):
        _require_device(x)
        lib = _lib.load()
        x = x.float().contiguous()
        B, C, H, W = x.shape
        ctx.shape, ctx.half = (B, C, H, W), half
        if half:
            y = empty_cl(B, 8, H, W, x.device, BF16)
            _lib.check(lib.dwc_pack_nchw_to_nhwc8_bf16(x.data_ptr(), y.data_ptr(), B, C, H, W, _stream()), "pack8")
        else:
            y = empty_cl(B, 4, H, W, x.device)
            _lib.check(lib.dwc_pack_nchw_to_nhwc4(x.data_ptr(), y.data_ptr(), B, C, H, W, _stream()), "pack")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, C, H, W = ctx.shape
        dy = cl(dy)
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        if ctx.half:
            _lib.check(lib.dwc_unpack_nhwc8_bf16_to_nchw(dy.data_ptr(), dx.data_ptr(), B, C, H, W, _stream()), "unpack8")
        else:
            _lib.check(lib.dwc_unpack_nhwc4_to_nchw(dy.data_ptr(), dx.data_ptr(), B, C, H, W, _stream()), "unpack")
        return dx, None


def is_image(x):
    """True for an internal image buffer (NHWC4 fp32 / NHWC8 bf16, channels-last)."""
    return x.dim() == 4 and x.dtype in (torch.float32, BF16) and x.shape[1] == image_planes(x.dtype) \
        and x.is_contiguous(memory_format=torch.channels_last)


def pack_image(x):
    """Bring an image batch to the internal form of the active precision (no-op when it already is one)."""
    if x.dtype == act_dtype() and is_image(x):
        return x
    if x.shape[1] > 3:
        raise ValueError("image tensors have at most 3 channels (or are already internal image buffers)")
    return _Pack4.apply(x, PRECISION == "bf16")


class _Blend(torch.autograd.Function):
    """x = img*att + real*(1-att) on NHWC4 tensors (att = plane 3 of `heads`); plane 3 of the result is 0."""

    @staticmethod
    def forward(ctx, heads, real):
        _require_device(heads)
        lib = _lib.load()
        heads, real = cl(heads), cl(real)
        B, C, H, W = heads.shape
        if real.dtype != heads.dtype or real.shape != heads.shape or C != image_planes(heads.dtype):
            raise ValueError("attention_blend needs two internal image buffers of one precision")
        out = empty_cl(B, C, H, W, heads.device, heads.dtype)
        _lib.check(_fn(lib, "blend_fwd", heads)(heads.data_ptr(), real.data_ptr(), out.data_ptr(), B * H * W, _stream()), "blend_fwd")
        ctx.save_for_backward(heads, real)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        heads, real = ctx.saved_tensors
        dout = cl(dout)
        B, C, H, W = heads.shape
        dh = empty_cl(B, C, H, W, heads.device, heads.dtype)
        _lib.check(_fn(lib, "blend_bwd", heads)(dout.data_ptr(), heads.data_ptr(), real.data_ptr(), dh.data_ptr(), B * H * W,
                                                _stream()), "blend_bwd")
        return dh, None


def attention_blend(heads, real):
    return _Blend.apply(heads, real)


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, skip4):
        _require_device(a)
        lib = _lib.load()
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        if a.dtype != b.dtype:
            a, b = a.float(), b.float()
        if a.dim() == 4:
            a, b = cl(a), cl(b)
        else:
            a, b = a.contiguous(), b.contiguous()
        n = a.numel()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = workspace(lib.dwc_l1_ws_bytes(n), a.device)
        _lib.check(_fn(lib, "l1_mean_fwd", a)(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, int(skip4), ws.data_ptr(), ws.numel(),
                                              _stream()), "l1_mean_fwd")
        ctx.save_for_backward(a, b)
        ctx.skip4 = int(skip4)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        dout = dout.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.check(_fn(lib, "l1_mean_bwd", a)(a.data_ptr(), b.data_ptr(), dout.float().data_ptr(), _p(da), _p(db), a.numel(),
                                              ctx.skip4, _stream()), "l1_mean_bwd")
        return da, db, None


def l1_mean(a, b, image=False):
    """mean |a-b| (reference solver.py:113-114).  image=True: internal image buffers (NHWC4 / NHWC8), mean over
    planes 0..2 only."""
    return _L1Mean.apply(a, b, a.shape[1] if image else 0)


GMM_FUSED = int(os.environ.get("DWC_GMM_FUSED", "1"))      # the KL style-space term on dwc_gmm_kl_sp_* (0: torch's elementwise algebra)
_SIGMA_F = {}                                               # id(sigma tensor) -> (weakref, version, python float)


def _scalar_value(t):
    """Python float of a 0-dim CONSTANT tensor (Solver.sigma), read from the device once per tensor and version -- not once per call,
    which would make the host wait for the stream inside every iteration."""
    if not torch.is_tensor(t):
        return float(t)
    ent = _SIGMA_F.get(id(t))
    if ent is None or ent[0]() is not t or ent[1] != t._version:
        ent = (weakref.ref(t), t._version, float(t.detach().float().item()))
        _SIGMA_F[id(t)] = ent
    return ent[2]


class _GmmKlSp(torch.autograd.Function):
    """sum_k mean_b sum_d KL(N(mu, e^lv) || N(centre[b][k], sigma)) over [B, K, D] heads (reference gmm.py:13-22): one launch forward,
    one backward (r06; the batched torch expression was 12 + 24 launches on 16 x 8 x 8 numbers, twice per iteration)."""

    @staticmethod
    def forward(ctx, mu, lv, centre, sigma):
        lib = _lib.load()
        mu, lv, centre = mu.contiguous(), lv.contiguous(), centre.float()
        if centre.stride(-1) != 1:
            centre = centre.contiguous()
        B, K, D = mu.shape
        out = torch.empty((), dtype=torch.float32, device=mu.device)
        _lib.check(lib.dwc_gmm_kl_sp_fwd(mu.data_ptr(), lv.data_ptr(), centre.data_ptr(), centre.stride(0), B, K, D, sigma, out.data_ptr(),
                                         _stream()), "gmm_kl_sp_fwd")
        ctx.save_for_backward(mu, lv, centre)
        ctx.sigma = sigma
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        mu, lv, centre = ctx.saved_tensors
        B, K, D = mu.shape
        dout = dout.float().contiguous()
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[0] else None
        dlv = torch.empty_like(lv) if ctx.needs_input_grad[1] else None
        _lib.check(lib.dwc_gmm_kl_sp_bwd(mu.data_ptr(), lv.data_ptr(), centre.data_ptr(), centre.stride(0), B, K, D, ctx.sigma,
                                         dout.data_ptr(), _p(dmu), _p(dlv), _stream()), "gmm_kl_sp_bwd")
        return dmu, dlv, None, None


def gmm_kl_sp(mu, lv, centre, sigma):
    """mu, lv: [B, K, D] fp32 device tensors; centre: [B, >= K]; sigma: python number or 0-dim tensor (a constant)."""
    return _GmmKlSp.apply(mu, lv, centre, _scalar_value(sigma))


# --------------------------------------------------------------------------------------
# loss tails
# --------------------------------------------------------------------------------------
class _AdvTail(torch.autograd.Function):
    """Adversarial loss of ONE discriminator scale over a batched pass (reference networks.py:116-170), one launch each way:
    sum_s w_src[s] * mean_s((src - target[s])^2) + w_cls[s] * mean_s(bce_with_logits(cls, labels))."""

    @staticmethod
    def forward(ctx, src, cls, labels, B, targets, w_src, w_cls):
        _require_device(src)
        lib = _lib.load()
        src, cls, labels = src.float().contiguous(), cls.float().contiguous(), labels.float().contiguous()
        segs = src.shape[0] // B
        if segs * B != src.shape[0] or cls.shape[0] != src.shape[0] or not 1 <= segs <= 4 or len(targets) != segs:
            raise ValueError("adv_tail: the batch must be 1..4 segments of B samples")
        sps, ncls = src[0].numel(), cls[0].numel()
        if labels.shape[0] != B or labels[0].numel() != ncls:
            raise ValueError("adv_tail: labels must be [B, ncls]")
        spec = _lib.AdvSpec()
        for s in range(segs):
            spec.target[s], spec.w_src[s], spec.w_cls[s] = float(targets[s]), float(w_src[s]), float(w_cls[s])
        out = torch.empty((), dtype=torch.float32, device=src.device)
        _lib.check(lib.dwc_adv_tail_fwd(src.data_ptr(), cls.data_ptr(), labels.data_ptr(), out.data_ptr(), segs, B, sps, ncls, spec,
                                        _stream()), "adv_tail_fwd")
        ctx.save_for_backward(src, cls, labels)
        ctx.geom = (segs, B, sps, ncls, spec)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        src, cls, labels = ctx.saved_tensors
        segs, B, sps, ncls, spec = ctx.geom
        dsrc, dcls = torch.empty_like(src), torch.empty_like(cls)
        _lib.check(lib.dwc_adv_tail_bwd(src.data_ptr(), cls.data_ptr(), labels.data_ptr(), dout.float().contiguous().data_ptr(),
                                        dsrc.data_ptr(), dcls.data_ptr(), segs, B, sps, ncls, spec, _stream()), "adv_tail_bwd")
        return dsrc, dcls, None, None, None, None, None


def adv_tail(src, cls, labels, B, targets, w_src, w_cls):
    return _AdvTail.apply(src, cls, labels, int(B), tuple(targets), tuple(w_src), tuple(w_cls))


class _WeightedSum(torch.autograd.Function):
    """sum_i w_i * t_i over 0-dim loss tensors: stack + one dot product forward, ONE scaled copy of the weights backward
    (the reference's loss_gen_total, solver.py:226-238, is thirteen scalar multiply-adds = ~40 launches with their backward)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        # (pinned staging: a copy from pageable memory makes the host wait for the stream -- here for the whole G forward, r06)
        w = torch.tensor(weights, dtype=torch.float32)
        w = (w.pin_memory() if terms[0].is_cuda and PINNED_STAGE else w).to(terms[0].device, non_blocking=True)
        ctx.save_for_backward(w)
        return torch.dot(torch.stack([t.float().reshape(()) for t in terms]), w)

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        return (None,) + (dout * w).unbind(0)


def weighted_sum(pairs):
    """pairs: (weight, 0-dim tensor or python number).  Numbers and zero-weight tensors fold into a constant."""
    const = sum(float(w) * float(t) for w, t in pairs if not torch.is_tensor(t))
    ts = [(float(w), t) for w, t in pairs if torch.is_tensor(t)]
    out = _WeightedSum.apply(tuple(w for w, _ in ts), *[t for _, t in ts])
    return out + const if const != 0.0 else out
