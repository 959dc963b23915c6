"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-device; this is the new capability.  Every
sample is independent through the conv/norm stack (IN/AdaIN per (n,c), LayerNorm per n, no
BatchNorm) and every loss is a batch mean, so equal shards + gradient AVERAGING reproduce the
global-batch gradient.  There is exactly one exchange per optimiser step: an all-reduce of
that optimiser's gradients (D: 13.99 M floats = 55.9 MB, G: 20.36 M = 81.4 MB at 128x128),
packed into flat fp32 buckets so that RCCL sees a few large messages (xGMI is point-to-point:
per-link bandwidth, not message rate, is the limit).

Two forms:
  * ``OverlappedGradReducer`` (what ``bench.py`` and the trainer use for world > 1): the gradients of a network ARE views
    of persistent flat fp32 buckets laid out in reverse parameter order (the order backward produces them), so there is
    no packing pass and no copy back; a post-accumulate-grad hook counts each bucket down and launches its all-reduce
    the moment its last gradient has been written, i.e. while the rest of backward still runs.  The
    layout is fixed at construction from the full parameter list, so it agrees across ranks by construction.
  * ``GradAllReduce``: the simple form (pack after backward, reduce, unpack) kept for callers that hand over an arbitrary
    parameter list.

The one piece that does not shard is the text encoder's batch-mixing ``view`` (reference
networks_v2.py:249): an N-rank run equals N independent local batches, not one global batch.
"""
import torch
import torch.distributed as dist

# Flat gradient buckets of this size: D's 53.4 MB of gradients make 3 buckets, G's 77.6 MB 4 (r05 shipped 64 MB: D was ONE bucket, whose
# all-reduce could only start when D's whole backward was done -- the overlap the reducer is built for needs several).  A ring all-reduce
# over the 7 xGMI links of a GPU moves a 20 MB message in ~0.3 ms: still bandwidth-, not latency-bound.
BUCKET_BYTES = 20 << 20


class GradAllReduce:
    """Callable handed to ``Solver`` (``solver.grad_sync``): averages the gradients of the given
    parameters across the process group through flat buckets.

    Parameters without a gradient on this step (e.g. the attention head while attention is
    switched off) are skipped on every rank alike — which parameters receive a gradient is a
    function of the iteration number only — so the bucket layout agrees across ranks and Adam's
    "skip parameters without grad" behaviour (SURVEY.md section 7 quirk viii) is preserved.
    """

    def __init__(self, group=None, bucket_bytes=BUCKET_BYTES):
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.calls = 0
        self.bytes = 0

    def _buckets(self, grads):
        cur, n = [], 0
        for g in grads:
            if cur and n + g.numel() > self.bucket_elems:
                yield cur
                cur, n = [], 0
            cur.append(g)
            n += g.numel()
        if cur:
            yield cur

    @torch.no_grad()
    def __call__(self, params):
        if self.world == 1:
            return
        grads = [p.grad for p in params if p.grad is not None]
        handles = []
        for bucket in self._buckets(grads):
            flat = torch.cat([g.reshape(-1) for g in bucket])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            handles.append((work, flat, bucket))
            self.calls += 1
            self.bytes += flat.numel() * 4
        inv = 1.0 / self.world
        for work, flat, bucket in handles:
            work.wait()
            flat.mul_(inv)
            views, off = [], 0
            for g in bucket:
                n = g.numel()
                views.append(flat[off:off + n].view_as(g))
                off += n
            torch._foreach_copy_(bucket, views)          # one multi-tensor launch instead of one copy per gradient


class OverlappedGradReducer:
    """Gradient averaging overlapped with backward for ONE optimiser's parameters.

    ``prepare(skip=())`` replaces ``optimizer.zero_grad()``: zero-fills the flat buckets (one memset each) and points every
    parameter's ``.grad`` at its view, so autograd accumulates in place.  ``skip``: parameters known to receive no gradient
    on this step (the attention head while attention is switched off, reference solver.py:109-111) -- they are not waited
    for.  During backward a hook per parameter counts its bucket down; a bucket whose count reaches zero is all-reduced
    asynchronously at once.  ``finish()`` (after backward) launches whatever is left, waits, and sets ``.grad = None`` on
    every parameter that received no gradient so that Adam skips it exactly as in a single-process run
    (SURVEY.md section 7 quirk viii: no weight decay / momentum for gradient-less parameters).  Averaging uses
    ReduceOp.AVG where the backend has it (RCCL) and SUM followed by one scale per bucket otherwise (gloo)."""

    def __init__(self, params, group=None, bucket_bytes=BUCKET_BYTES):
        self.group = group
        self.world = dist.get_world_size(group)
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, cap = self.params[0].device, max(1, bucket_bytes // 4)
        self.avg = dist.get_backend(group) == "nccl"
        # reverse parameter order: the last layers' gradients are produced first
        layout, cur, n = [], [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                layout.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        layout.append(cur)
        self.buckets = []          # dicts: flat buffer, params, views
        self.where = {}            # id(param) -> bucket index
        for bi, plist in enumerate(layout):
            flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self.where[id(p)] = bi
            self.buckets.append({"flat": flat, "params": plist, "views": views, "pending": 0, "work": None})
        self.touched = set()
        self.force = False         # development: issue the collectives even in a one-rank group (exercises the RCCL path)
        self.active = False
        self.launched_early = 0    # buckets whose all-reduce started from inside backward (overlap evidence)
        self.calls = 0
        self.bytes = 0
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _launch(self, b):
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)
        self.calls += 1
        self.bytes += b["flat"].numel() * 4

    def _hook(self, p):
        # (the hook also fires when autograd had NO gradient for p -- an op returned None: parameters flagged `_dwc_zero_grad` are
        # not counted, they end the step gradient-less as in a single-process run and FusedAdam steps them with zeros)
        if not self.active or id(p) in self.touched or getattr(p, "_dwc_zero_grad", False):
            return
        self.touched.add(id(p))
        b = self.buckets[self.where[id(p)]]
        b["pending"] -= 1
        if b["pending"] == 0 and b["work"] is None and (self.world > 1 or self.force):
            self._launch(b)
            self.launched_early += 1

    @torch.no_grad()
    def prepare(self, skip=()):
        skip_ids = {id(p) for p in skip}
        # (biases whose gradient is identically zero -- hipdwc.ops marks them `_dwc_zero_grad` on their first forward -- receive no
        # gradient from autograd: not waited for either; their zero-filled bucket slice is all-reduced as zeros)
        skip_ids |= {id(p) for p in self.params if getattr(p, "_dwc_zero_grad", False)}
        self.touched.clear()
        for b in self.buckets:
            b["flat"].zero_()
            b["work"] = None
            b["pending"] = sum(1 for p in b["params"] if id(p) not in skip_ids)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        self.active = True

    @torch.no_grad()
    def finish(self):
        self.active = False
        if self.world > 1 or self.force:
            for b in self.buckets:
                if b["work"] is None:
                    self._launch(b)
            for b in self.buckets:
                b["work"].wait()
                if not self.avg:
                    b["flat"].mul_(1.0 / self.world)
        for p in self.params:
            if id(p) not in self.touched:
                p.grad = None          # no gradient this step: Adam must skip it (same set on every rank)


def broadcast_module(module, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    with torch.no_grad():
        tensors = list(module.parameters()) + list(module.buffers())
        for t in tensors:
            dist.broadcast(t.data, src=src, group=group)
        # written through .data: bump the version counters so caches keyed on them (hipdwc.ops._prepped) notice
        torch.autograd.graph.increment_version(tensors)


def shard_batch(batch, rank, world):
    """Rank r takes samples [r*B/W, (r+1)*B/W) of every tensor in the batch dict."""
    out = {}
    for k, v in batch.items():
        b = v.shape[0]
        if b % world:
            raise ValueError("global batch %d not divisible by world size %d" % (b, world))
        per = b // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out
