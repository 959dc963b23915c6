"""One-rank RCCL run of the data-parallel training path (tests/test_dp_gpu.py::test_one_rank_rccl_path_equals_plain_trainer).

A 1-GPU box cannot run the two-rank test, so without this the path bench.py and Solver.enable_data_parallel default to
(OverlappedGradReducer on backend nccl = RCCL: gradients as views of flat buckets, ReduceOp.AVG, async all-reduce launched from
the post-accumulate-grad hook on RCCL's side stream, stream ordering against backward and the optimiser) would never touch
hardware.  A world_size = 1 group with ``reducer.force`` issues every collective for real; averaging over one rank is the
identity, so gradients and parameters after two iterations must EQUAL those of the plain single-process trainer."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "dwc-gan_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from hipdwc import host, synth                # noqa: E402
from solver import Solver                     # noqa: E402


def run(dp, dev, cfg, batch):
    host.set_noise(host.HostNoise())
    torch.manual_seed(1234)
    trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    if dp:
        trainer.enable_data_parallel(bucket_bytes=64 << 10)        # small buckets: several per network
        for r in trainer._reducers.values():
            r.force = True
    grads = {}
    for it in range(2):
        torch.manual_seed(99 + it)
        a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"], batch["label_src"],
             batch["label_trg"], cfg, it)
        trainer.dis_update(*a)
        grads["dis%d" % it] = {k: p.grad.detach().clone() for k, p in trainer.dis.named_parameters() if p.grad is not None}
        trainer.gen_update(*a)
        grads["gen%d" % it] = {k: p.grad.detach().clone() for k, p in trainer.gen.named_parameters() if p.grad is not None}
        trainer.smooth_moving()
        trainer.update_learning_rate()
        trainer.update_attention_status(it)
    torch.cuda.synchronize()
    return trainer, grads


def compare(dev):
    """Plain trainer vs the forced one-rank RCCL trainer (process group already initialised by the caller)."""
    cfg = synth.make_config(image_size=32, tiny=True, lstm_dropout=0.0)
    batch = {k: v.to(dev) for k, v in synth.make_batch(3, 32, seed=5).items()}
    try:
        plain, g_plain = run(False, dev, cfg, batch)
        dp, g_dp = run(True, dev, cfg, batch)
    finally:
        host.set_noise(host.DeviceNoise())
    res = {"grad_keys_equal": True, "max_grad_diff": 0.0, "max_param_diff": 0.0}
    for step in g_plain:
        if set(g_plain[step]) != set(g_dp[step]):          # same parameters gradient-less (attention head while attention is off)
            res["grad_keys_equal"] = False
            res.setdefault("key_diff", {})[step] = {"only_plain": sorted(set(g_plain[step]) - set(g_dp[step]))[:6],
                                                    "only_dp": sorted(set(g_dp[step]) - set(g_plain[step]))[:6]}
        for k in g_plain[step]:
            if k in g_dp[step]:
                res["max_grad_diff"] = max(res["max_grad_diff"], float((g_plain[step][k] - g_dp[step][k]).abs().max()))
    for (k, a), (_, b) in zip(list(plain.gen.named_parameters()) + list(plain.dis.named_parameters()),
                              list(dp.gen.named_parameters()) + list(dp.dis.named_parameters())):
        res["max_param_diff"] = max(res["max_param_diff"], float((a - b).abs().max()))
    res["all_reduces"] = {k: r.calls for k, r in dp._reducers.items()}
    res["launched_early"] = {k: r.launched_early for k, r in dp._reducers.items()}
    res["buckets"] = {k: len(r.buckets) for k, r in dp._reducers.items()}
    res["avg_op"] = all(r.avg for r in dp._reducers.values())
    return res


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    print("DPFORCE " + json.dumps(compare(dev)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
