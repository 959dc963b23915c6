"""Rank process of tests/test_dp_gpu.py (launched with torch.distributed.run, one rank per GPU, backend nccl = RCCL).

Runs two data-parallel training iterations of the tiny configuration on the HIP kernels with the overlapped gradient
reducer, then checks that (a) every parameter of G and D is bit-identical across the ranks, (b) rank-averaged HIP gradients
of D and G at step 0 equal the average of the CPU oracle's per-shard gradients (the text encoder mixes samples within a
LOCAL batch, reference networks_v2.py:249, so the emulation applies the oracle per shard), (c) all-reduces were launched
from inside backward."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "dwc-gan_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from hipdwc import dp, host, synth            # noqa: E402
from oracle import dwcgan_oracle as orc       # noqa: E402
from solver import Solver                     # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    cfg = synth.make_config(image_size=32, tiny=True, lstm_dropout=0.0)
    per = 2
    host.set_noise(host.HostNoise())
    torch.manual_seed(1234)
    trainer = Solver(cfg, dev, None).to(dev)
    dp.broadcast_module(trainer.gen)
    dp.broadcast_module(trainer.dis)
    trainer.copy_nets()
    trainer.enable_data_parallel(bucket_bytes=64 << 10)            # small buckets: several per network
    full = synth.make_batch(per * world, 32, seed=5)
    mine = {k: v.to(dev) for k, v in dp.shard_batch(full, rank, world).items()}
    init_g = {k: v.detach().cpu().clone() for k, v in trainer.gen.state_dict().items()}
    init_d = {k: v.detach().cpu().clone() for k, v in trainer.dis.state_dict().items()}

    grads0 = {}
    for it in range(2):
        torch.manual_seed(777 + 10 * it + rank)                    # per-rank random stream, replayed by the oracle below
        a = (mine["x_real"], mine["c_src"], mine["c_trg"], mine["txt"], mine["txt_lens"], mine["label_src"], mine["label_trg"],
             cfg, it)
        trainer.dis_update(*a)
        if it == 0:
            grads0["dis"] = {k: p.grad.detach().cpu().clone() for k, p in trainer.dis.named_parameters() if p.grad is not None}
        trainer.gen_update(*a)
        if it == 0:
            grads0["gen"] = {k: p.grad.detach().cpu().clone() for k, p in trainer.gen.named_parameters() if p.grad is not None}
        trainer.smooth_moving()
        trainer.update_learning_rate()
        trainer.update_attention_status(it)
    torch.cuda.synchronize()

    # (a) identical parameters on every rank
    flat = torch.cat([p.detach().reshape(-1) for p in list(trainer.gen.parameters()) + list(trainer.dis.parameters())])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    early = {k: r.launched_early for k, r in trainer._reducers.items()}
    buckets = {k: len(r.buckets) for k, r in trainer._reducers.items()}

    res = {"rank": rank, "same_params": bool(same), "launched_early": early, "buckets": buckets}
    if rank == 0:
        # (b) oracle: per-shard gradients at step 0, averaged over the shards
        acc = {"dis": {}, "gen": {}}
        for r in range(world):
            sh = dp.shard_batch(full, r, world)
            o = orc.OracleSolver(cfg, init_g, init_d)
            o.copy_nets()
            torch.manual_seed(777 + r)
            a = (sh["x_real"], sh["c_src"], sh["c_trg"], sh["txt"], sh["txt_lens"], sh["label_src"], sh["label_trg"], cfg, 0)
            o.dis_update(*a)
            # the HIP ranks stepped D with the AVERAGED gradient before their G step; the oracle shard stepped with its own:
            # compare D gradients (taken before any step) exactly, and G gradients only loosely (D differs by one Adam step
            # of lr 1e-4 between the two runs)
            for k, g in o.last_dis_grads.items():
                if g is not None:
                    acc["dis"][k] = acc["dis"].get(k, 0) + g.detach() / world
            o.gen_update(*a)
            for k, g in o.last_gen_grads.items():
                if g is not None:
                    acc["gen"][k] = acc["gen"].get(k, 0) + g.detach() / world

        def worst(name, tol_keys=None):
            w = 0.0
            for k, g in grads0[name].items():
                ref = acc[name][k]
                if float(ref.abs().max()) < 1e-6:      # biases in front of an instance norm: exact zeros here, rounding noise there
                    continue
                w = max(w, float((g - ref).abs().max() / ref.abs().max()))
            return w
        res["dis_grad_rel_err"] = worst("dis")
        res["gen_grad_rel_err"] = worst("gen")
        res["dis_grad_keys"] = [len(grads0["dis"]), len(acc["dis"])]
    print("DPRESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
