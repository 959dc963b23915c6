"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-device; this is the new capability.  Every
sample is independent through the conv/norm stack (IN/AdaIN per (n,c), LayerNorm per n, no
BatchNorm) and every loss is a batch mean, so equal shards + gradient AVERAGING reproduce the
global-batch gradient.  There is exactly one exchange per optimiser step: an all-reduce of
that optimiser's gradients (D: 13.99 M floats = 55.9 MB, G: 20.36 M = 81.4 MB at 128x128),
packed into flat fp32 buckets so that RCCL sees a few large messages (xGMI is point-to-point:
per-link bandwidth, not message rate, is the limit).  ~1 ms on the wire against a step of
~100 ms, so the exchange is issued right after backward without further overlap machinery.

The one piece that does not shard is the text encoder's batch-mixing ``view`` (reference
networks_v2.py:249): an N-rank run equals N independent local batches, not one global batch.
"""
import torch
import torch.distributed as dist


class GradAllReduce:
    """Callable handed to ``Solver`` (``solver.grad_sync``): averages the gradients of the given
    parameters across the process group through flat buckets.

    Parameters without a gradient on this step (e.g. the attention head while attention is
    switched off) are skipped on every rank alike — which parameters receive a gradient is a
    function of the iteration number only — so the bucket layout agrees across ranks and Adam's
    "skip parameters without grad" behaviour (SURVEY.md section 7 quirk viii) is preserved.
    """

    def __init__(self, group=None, bucket_bytes=64 << 20):
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
