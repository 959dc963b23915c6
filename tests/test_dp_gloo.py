"""Data-parallel path on CPU: world size 2, gloo backend (the GPU runs use the same code over RCCL).

Checks (1) the flat-bucket gradient all-reduce of hipdwc.dp.GradAllReduce, including parameters
that have no gradient on a step and several buckets, (2) that equal shards + gradient averaging
reproduce the full-batch gradient of the discriminator objective (computed with the CPU oracle,
which is allowed in tests), (3) parameter broadcast and batch sharding.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hipdwc import dp, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(100 + rank)

        # (1) bucketed all-reduce with a grad-less parameter and a tiny bucket size (forces 3 buckets)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 7, 1200, 64)]
        for i, p in enumerate(params):
            if i != 2:                                   # params[2] has no gradient on this "step"
                p.grad = torch.full((p.numel(),), float(rank + 1) * (i + 1))
        sync = dp.GradAllReduce(bucket_bytes=4 * 400)
        sync(params)
        ok1 = params[2].grad is None and sync.calls >= 3
        for i, p in enumerate(params):
            if i != 2:
                ok1 = ok1 and torch.allclose(p.grad, torch.full_like(p.grad, 1.5 * (i + 1)))   # mean of ranks 1x,2x

        # (2) sharded discriminator gradient == full-batch gradient
        from oracle import dwcgan_oracle as orc
        ops_g = np.load(os.path.join(GOLDEN, "ops_golden.npz"))
        D = {k[len("dis/sd/"):]: torch.from_numpy(ops_g[k]).clone().requires_grad_(True)
             for k in ops_g.files if k.startswith("dis/sd/")}
        cfg = {"n_layer": 3, "num_scales": 2, "activ": "lrelu"}
        g = torch.Generator().manual_seed(5)
        full = {"fake": torch.randn(4, 3, 32, 32, generator=g), "real": torch.randn(4, 3, 32, 32, generator=g),
                "lab": (torch.rand(4, 8, generator=g) < 0.5).float()}
        mine = dp.shard_batch(full, rank, world)
        loss = orc.calc_dis_loss(D, mine["fake"], mine["real"], mine["lab"], 1.0, 1.0, cfg)
        names = list(D.keys())
        grads = torch.autograd.grad(loss, [D[k] for k in names])
        plist = []
        for k, gr in zip(names, grads):
            p = torch.nn.Parameter(D[k].detach().clone())
            p.grad = gr.clone()
            plist.append(p)
        dp.GradAllReduce()(plist)
        loss_full = orc.calc_dis_loss(D, full["fake"], full["real"], full["lab"], 1.0, 1.0, cfg)
        gfull = torch.autograd.grad(loss_full, [D[k] for k in names])
        err = max(((p.grad - gf).abs().max() / (gf.abs().max() + 1e-12)).item() for p, gf in zip(plist, gfull))

        # (3) broadcast
        lin = torch.nn.Linear(4, 3)
        dp.broadcast_module(lin, src=0)
        w = lin.weight.detach().clone()
        gathered = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        ok3 = all(torch.equal(gathered[0], t) for t in gathered)
        # (4) overlapped reducer: gradients are views of flat buckets, buckets are all-reduced from inside backward,
        #     a parameter that is off the tape gets grad None (Adam skips it), result == full-batch gradient
        torch.manual_seed(7)                              # same weights on both ranks
        net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                  torch.nn.Linear(32, 3))
        unused = torch.nn.Parameter(torch.ones(5))       # like the attention head while attention is off
        plist4 = list(net.parameters()) + [unused]
        red = dp.OverlappedGradReducer(plist4, bucket_bytes=4 * 600)
        xs = torch.randn(8, 6, generator=torch.Generator().manual_seed(3))
        ys = torch.randn(8, 3, generator=torch.Generator().manual_seed(4))
        ok4 = len(red.buckets) >= 3
        for step in range(2):                             # twice: buffers are re-used, hooks fire again
            red.prepare(skip=[unused])
            sl = slice(rank * 4, rank * 4 + 4)
            ((net(xs[sl]) - ys[sl]) ** 2).mean().backward()
            early = red.launched_early
            red.finish()
            full_loss = ((net(xs) - ys) ** 2).mean()
            gfull4 = torch.autograd.grad(full_loss, list(net.parameters()))
            ok4 = ok4 and unused.grad is None and early >= 1
            for p_, gf in zip(net.parameters(), gfull4):
                ok4 = ok4 and p_.grad is not None and torch.allclose(p_.grad, gf, rtol=1e-5, atol=1e-6)
                bk = red.buckets[red.where[id(p_)]]
                idx = [i for i, q_ in enumerate(bk["params"]) if q_ is p_][0]
                ok4 = ok4 and p_.grad.data_ptr() == bk["views"][idx].data_ptr()       # the gradient IS the bucket slice
        q.put((rank, bool(ok1 and ok4), float(err), bool(ok3), mine["fake"].shape[0]))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        q.put((rank, False, repr(e), False, -1))


def test_data_parallel_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok1, err, ok3, nshard in sorted(results):
        assert ok1, (rank, err)
        assert isinstance(err, float) and err < 1e-5, (rank, err)
        assert ok3 and nshard == 2


def _solver_worker(rank, world, port, q):
    """The REAL Solver's data-parallel bookkeeping (enable_data_parallel / _zero_grad / _sync_grads, hipdwc.dp.OverlappedGradReducer)
    on two gloo ranks.  The networks' kernels need a GPU, so backward is replaced by a synthetic loss that hands every parameter
    the step would reach a rank-dependent gradient THROUGH autograd (the reducers' post-accumulate hooks fire as in training);
    everything else -- bucket layout from the real parameter lists, the attention head's skip set at iterations 0 / 1, gradients
    as bucket views, averaging, grad = None for untouched parameters -- is the product code."""
    try:
        import contextlib
        import io
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from solver import Solver
        cfg = synth.make_config(image_size=32, tiny=True)
        torch.manual_seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            s = Solver(cfg, torch.device("cpu"), None)
        s.copy_nets()
        s.enable_data_parallel(bucket_bytes=16 << 10)              # tiny buckets: several per network
        ok, why = True, []

        def check(cond, msg):
            nonlocal ok
            if not cond:
                ok = False
                why.append(msg)
        for which, net in (("dis", s.dis), ("gen", s.gen)):
            red = s._reducers[which]
            trainable = [p for p in net.parameters() if p.requires_grad]
            check(len(red.buckets) >= 3, "%s: expected several buckets" % which)
            # every trainable parameter sits in exactly one bucket, buckets in REVERSE parameter order
            flat_order = [p for b in red.buckets for p in b["params"]]
            check(len(flat_order) == len(trainable) and all(a is b for a, b in zip(flat_order, reversed(trainable))), "%s: layout" % which)
            # the layout agrees across ranks
            sizes = torch.tensor([b["flat"].numel() for b in red.buckets], dtype=torch.int64)
            both = [torch.zeros_like(sizes) for _ in range(world)]
            dist.all_gather(both, sizes)
            check(all(torch.equal(both[0], t) for t in both), "%s: bucket sizes differ across ranks" % which)
        att = {id(p) for p in s.gen.dec.image_attention.parameters()}
        check(len(att) > 0, "attention head has parameters")
        for it in (0, 1):
            s.use_attention = it == 0                               # what update_attention_status leaves after iterations 0 / 1
            for which, net in (("dis", s.dis), ("gen", s.gen)):
                red = s._reducers[which]
                s._zero_grad(which)
                named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
                reached = [(i, p) for i, (n, p) in enumerate(named) if not (which == "gen" and it == 1 and id(p) in att)]
                loss = sum((p * float((rank + 1) * (1 + 0.001 * i))).sum() for i, p in reached)
                loss.backward()
                early = red.launched_early
                s._sync_grads(which)
                check(early >= 1, "%s it%d: no bucket was reduced from inside backward" % (which, it))
                for i, (n, p) in enumerate(named):
                    if which == "gen" and it == 1 and id(p) in att:
                        check(p.grad is None, "%s it%d: %s must have grad None (Adam skips it)" % (which, it, n))
                        continue
                    want = 1.5 * (1 + 0.001 * i)                    # mean over the two ranks of (rank + 1) * c_i
                    check(p.grad is not None and torch.allclose(p.grad, torch.full_like(p.grad, want), rtol=1e-6, atol=0), "%s it%d: %s" % (which, it, n))
                    bk = red.buckets[red.where[id(p)]]
                    j = [k for k, q_ in enumerate(bk["params"]) if q_ is p][0]
                    check(p.grad.data_ptr() == bk["views"][j].data_ptr(), "%s it%d: %s is not its bucket's view" % (which, it, n))
        q.put((rank, ok, "; ".join(why[:5])))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()[-600:]))


def test_solver_data_parallel_bookkeeping_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_solver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, why in sorted(results):
        assert ok, (rank, why)


def test_shard_batch_rejects_ragged():
    b = synth.make_batch(3, 8)
    with pytest.raises(ValueError):
        dp.shard_batch(b, 0, 2)
    halves = [dp.shard_batch(synth.make_batch(4, 8), r, 2) for r in range(2)]
    assert torch.equal(torch.cat([h["txt"] for h in halves]), synth.make_batch(4, 8)["txt"])
