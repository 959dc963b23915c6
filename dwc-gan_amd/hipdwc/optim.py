"""Multi-tensor Adam and EMA on the HIP kernels (``dwc_adam_multi`` / ``dwc_ema_multi``).

``FusedAdam`` is a ``torch.optim.Adam`` (same constructor, ``param_groups``, ``state_dict`` layout,
LR-scheduler compatibility) whose ``step()`` updates every parameter tensor of the group in ONE
kernel launch instead of torch's ~12 multi-tensor launches: coupled L2 weight decay, bias-corrected
moments, eps outside the sqrt, and parameters without a gradient skipped entirely — the behaviour
of the optimiser the reference builds at solver.py:62-68 under torch >= 2 (SURVEY.md section 7, quirk viii).
``FusedEMA`` is the reference's ``moving_average`` (utils.py:52-54) in one launch per network.
On CPU tensors both fall back to ... nothing: they raise, like every other HIP op (the CPU
restatement lives in oracle/).
"""
import math

import numpy as np
import torch

from . import _lib
from . import ops as _ops

CHUNK = 8192           # must equal DWC_OPT_CHUNK in include/dwcgan_hip.h
REFRESH_WITH_STEP = True   # FusedAdam.step() ends with hipdwc.ops.refresh_prepared (False: every layout lazily at its next use)

_ADAM_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<u8"),
                     ("step_size", "<f4"), ("bc2_sqrt", "<f4")])      # struct dwc_adam_tensor
_EMA_DT = np.dtype([("p", "<u8"), ("ema", "<u8"), ("n", "<u8")])      # struct dwc_ema_tensor


def _chunk_maps(sizes, device):
    tid, start = [], []
    for i, n in enumerate(sizes):
        for s in range(0, n, CHUNK):
            tid.append(i)
            start.append(s)
    assert max(sizes) < 2 ** 31
    return torch.tensor(tid, dtype=torch.int32, device=device), torch.tensor(start, dtype=torch.int32, device=device), len(tid)


def _to_device_bytes(arr, device):
    """Upload a descriptor array without stalling the host: pinned staging block (torch's caching host
    allocator keeps it alive until the copy has executed) + asynchronous copy on the current stream."""
    host = torch.from_numpy(arr.view(np.uint8).reshape(-1)).pin_memory()
    return host.to(device, non_blocking=True)


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._maps = {}
        self._host = {}        # group index -> host-side tables of step() (_host_tables)
        self._zeros = {}       # device -> a zero vector that stands in for the gradient of `_dwc_zero_grad` parameters

    def _zero_grad_for(self, p):
        """Parameters marked ``_dwc_zero_grad`` by the op that consumes them (convolution biases in front of an instance norm: the mean
        subtraction removes them, their gradient is identically zero, hipdwc.ops._Conv2d) get NO gradient tensor from autograd -- no fill
        launch per layer and use -- and are stepped here with a shared zero vector: the same update an explicit zero gradient gave
        (weight decay included)."""
        z = self._zeros.get(p.device)
        if z is None or z.numel() < p.numel():
            z = torch.zeros(max(4096, p.numel()), dtype=torch.float32, device=p.device)
            self._zeros[p.device] = z
        return z

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._host = {}            # step counts and moment pointers are re-read from the loaded state

    def _host_tables(self, gi, params):
        """Per group, built once (and again when the parameter list or a loaded state changes it): the descriptor array with the constant
        fields filled in, the step counts as a numpy vector mirroring ``state[p]['step']``, the element counts.  step() used to do all of
        this per parameter and call in Python -- ~1 ms per call for G's ~130 tensors, at a moment when the GPU has just drained the
        backward's last small kernels and sits idle (r06, `benchmarks/gpu_idle_gaps.py`: an 850 us gap per iteration in front of the
        descriptor upload)."""
        key = tuple(p.data_ptr() for p in params)
        ent = self._host.get(gi)
        if ent is not None and ent["key"] == key:
            return ent
        desc = np.zeros(len(params), dtype=_ADAM_DT)
        steps = np.zeros(len(params), dtype=np.float64)
        for i, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if not p.is_contiguous():
                raise RuntimeError("FusedAdam expects contiguous parameters")
            desc[i]["p"], desc[i]["m"], desc[i]["v"] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            desc[i]["n"] = p.numel()
            steps[i] = float(st["step"])
        ent = {"key": key, "desc": desc, "steps": steps, "numel": desc["n"].astype(np.int64),
               "step_tensors": [self.state[p]["step"] for p in params],
               "flagged": np.array([bool(getattr(p, "_dwc_zero_grad", False)) for p in params])}
        self._host[gi] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure not supported")
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            dev = params[0].device
            if not params[0].is_cuda:
                raise RuntimeError("FusedAdam needs device parameters: the product path has no CPU fallback")
            b1, b2 = group["betas"]
            tab = self._host_tables(gi, params)
            desc = tab["desc"].copy()
            keep = []          # keeps contiguous gradient copies alive until the launch is queued
            gptr = np.zeros(len(params), dtype=np.uint64)
            flagged = tab["flagged"]
            for i, p in enumerate(params):
                g = p.grad
                if g is None:
                    if not flagged[i]:
                        if getattr(p, "_dwc_zero_grad", False):          # (flag set after the tables were built: first iteration)
                            flagged[i] = True
                        else:
                            continue                                      # g stays NULL: tensor skipped, step not advanced
                    g = self._zero_grad_for(p)
                elif g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                    keep.append(g)
                gptr[i] = g.data_ptr()
            has = gptr != 0
            steps = tab["steps"]
            steps[has] += 1.0
            torch._foreach_add_([t for t, h in zip(tab["step_tensors"], has) if h], 1.0)      # state_dict()'s view of the counts
            t = np.where(has, steps, 1.0)
            desc["g"] = gptr
            desc["step_size"] = np.where(has, group["lr"] / (1.0 - np.power(b1, t)), 0.0)
            desc["bc2_sqrt"] = np.where(has, np.sqrt(1.0 - np.power(b2, t)), 0.0)
            maps = self._maps.get(gi)
            if maps is None or maps[4] != tab["key"]:
                sizes = [int(n) for n in tab["numel"]]
                tid, start, n = _chunk_maps(sizes, dev)
                maps = (tid, start, n, sizes, tab["key"])
                self._maps[gi] = maps
            ddev = _to_device_bytes(desc, dev)
            _lib.check(lib.dwc_adam_multi(ddev.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2], b1, b2,
                                          group["eps"], group["weight_decay"], _ops._stream()),
                       "adam_multi")
            del keep
            stepped = [p for p, h in zip(params, has) if h]
            _ops._hbm("adam_multi", 28 * int(tab["numel"][has].sum()))                                # p, g, m, v read; p, m, v written
            # the kernel wrote through raw pointers: tell autograd (and the prepared-weight cache in
            # hipdwc.ops, which keys on the version counter) that these tensors changed
            touched = stepped
            torch.autograd.graph.increment_version(touched)
            # ... and rebuild every prepared layout of the updated weights in ONE launch, here, instead of one launch per layout
            # at each layer's next use (Adam + weight refresh as one pass over the parameters)
            if REFRESH_WITH_STEP:
                from . import ops
                ops.refresh_prepared(touched)
        return None


class FusedEMA:
    """copy <- lerp(param, copy, beta) over the parameters of ``model`` / ``model_copy`` (buffers are not averaged)."""

    def __init__(self, model, model_copy):
        src = [p for p in model.parameters()]
        dst = [p for p in model_copy.parameters()]
        assert len(src) == len(dst) and all(a.shape == b.shape for a, b in zip(src, dst))
        if not src[0].is_cuda:
            raise RuntimeError("FusedEMA needs device parameters: the product path has no CPU fallback")
        self.src, self.dst = src, dst
        dev = src[0].device
        desc = np.zeros(len(src), dtype=_EMA_DT)
        for i, (a, b) in enumerate(zip(src, dst)):
            if not (a.is_contiguous() and b.is_contiguous()):
                raise RuntimeError("FusedEMA expects contiguous parameters")
            desc[i]["p"], desc[i]["ema"], desc[i]["n"] = a.data_ptr(), b.data_ptr(), a.numel()
        self.ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in zip(src, dst)]
        self.desc = _to_device_bytes(desc, dev)
        self.tid, self.start, self.n_chunks = _chunk_maps([p.numel() for p in src], dev)

    def still_valid(self):
        return all(a.data_ptr() == pa and b.data_ptr() == pb for (a, b), (pa, pb) in zip(zip(self.src, self.dst), self.ptrs))

    @torch.no_grad()
    def step(self, beta=0.999):
        lib = _lib.load()
        _lib.check(lib.dwc_ema_multi(self.desc.data_ptr(), self.tid.data_ptr(), self.start.data_ptr(), self.n_chunks,
                                     beta, _ops._stream()), "ema_multi")
        _ops._hbm("ema_multi", 12 * sum(p.numel() for p in self.src))                                # p, copy read; copy written
        torch.autograd.graph.increment_version(self.dst)
