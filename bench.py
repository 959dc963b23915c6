#!/usr/bin/env python3
"""Throughput benchmark of the DWC-GAN training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one full training iteration of the reference loop body (reference train.py:102-111):
dis_update + gen_update + smooth_moving + update_learning_rate + update_attention_status, on a
synthetic CelebA-shaped batch already resident in HBM.  Workload = BASELINE.json configs[1]:
128x128, per-GPU batch 16, fp32 end to end (weak scaling: the global batch is 16*N).
Rank 0 prints ONE JSON line with the metric, the live roofline figure of the dominant kernel
and (N == 1) the CPU baseline (the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "dwc-gan_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from hipdwc import host, ops, synth  # noqa: E402

IMAGE_SIZE = 128
PER_GPU_BATCH = 16
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
ALGO_GFLOP_PER_IMAGE = 569.6           # BASELINE.md section 4 / SURVEY.md 8(d): necessary fwd+bwd conv+linear work
DOMINANT = "conv_gemm_kernel"          # the forward / data-gradient GEMM family: one span = one conv call = conv_gemm_kernel, or for a 3x3
                                       # layer wino_input + wino_fused_kernel (or conv_gemm_batched_kernel + wino_output) (+ ring strips and fold for a data gradient)
# HBM-side bytes per launch of that kernel from the PMC passes committed as profiles/r01_pmc_hbm_traffic.json
# (rocprofv3 --pmc FETCH_SIZE and, separately, WRITE_SIZE, same command; FETCH_SIZE doubled per the gfx950 note
# in MI355X_MICROARCH.md).  A profile-time constant: bench.py cannot read PMCs itself.
DOMINANT_TRAFFIC_BYTES_PER_LAUNCH = 421611563


def run_iteration(trainer, batch, cfg, it):
    a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"], batch["label_src"],
         batch["label_trg"], cfg, it)
    trainer.dis_update(*a)
    trainer.gen_update(*a)
    trainer.smooth_moving()
    trainer.update_learning_rate()
    trainer.update_attention_status(it)


def cpu_baseline(gen_sd, dis_sd, cfg, sample_batch=4):
    """The CPU oracle (as_written: with the work the reference also does and discards) on a bounded
    sample: one warm-up + one timed iteration at the same 128x128 graph with a quarter batch."""
    from oracle import dwcgan_oracle as orc
    torch.manual_seed(4321)
    solver = orc.OracleSolver(cfg, gen_sd, dis_sd, as_written=True)
    solver.copy_nets()
    batch = synth.make_batch(sample_batch, IMAGE_SIZE, seed=99)
    solver.iteration(batch, 0)
    t0 = time.time()
    solver.iteration(batch, 1)
    dt = time.time() - t0
    return {"value": sample_batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle (torch CPU fp32, as-written graph) 1 warm-up + 1 timed iteration, 128x128, batch %d "
                      "(%.1f s)" % (sample_batch, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-gpu-batch", type=int, default=PER_GPU_BATCH,
                    help="development knob; the contract workload (and the default) is 16")
    ap.add_argument("--winograd", type=int, default=None, choices=(0, 2, 4),
                    help="development knob: Winograd output tile of the 3x3 convolutions (default 2 = F(2x2,3x3); 4 = F(4x4,3x3), "
                         "faster but ~10x the rounding error, see hipdwc/ops.py; 0 = direct)")
    ap.add_argument("--vgg-w", type=float, default=0.0,
                    help="development knob: perceptual-loss weight (the reference's shipped default is 0.1) with a RANDOMLY "
                         "initialised VGG16 (the trained weights cannot be fetched here); the contract workload is 0")
    args = ap.parse_args()

    per_gpu_batch = args.per_gpu_batch
    if args.winograd is not None:
        ops.WINOGRAD_TILE = args.winograd
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run for N > 1" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    from solver import Solver
    cfg = synth.make_config(image_size=IMAGE_SIZE)           # shipped config, vgg_w = 0 (weights not obtainable offline)
    if args.vgg_w > 0:
        import tempfile
        from networks.networks import Vgg16
        vgg_dir = tempfile.mkdtemp(prefix="dwc_vgg_%d_" % rank)
        os.makedirs(os.path.join(vgg_dir, "models"))
        torch.manual_seed(777)
        torch.save(Vgg16().state_dict(), os.path.join(vgg_dir, "models", "vgg16.weight"))
        cfg["vgg_w"], cfg["vgg_model_path"] = args.vgg_w, vgg_dir
    torch.manual_seed(1234)                                  # same seed on every rank: identical initial weights
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    torch.cuda.manual_seed(1234 + rank)                      # per-rank style samples / dropout masks
    host.set_noise(host.DeviceNoise())
    if world > 1:
        from hipdwc import dp
        dp.broadcast_module(trainer.gen)
        dp.broadcast_module(trainer.dis)
        trainer.copy_nets()
        trainer.grad_sync = dp.GradAllReduce()
    init_gen = {k: v.detach().cpu().clone() for k, v in trainer.gen.state_dict().items()}
    init_dis = {k: v.detach().cpu().clone() for k, v in trainer.dis.state_dict().items()}

    # fresh batch per iteration, pre-generated on the device (4 distinct, cycled)
    batches = [synth.make_batch(per_gpu_batch, IMAGE_SIZE, seed=1000 * rank + i, device=dev) for i in range(4)]
    for b in batches:
        b["txt_lens"] = b["txt_lens"].cpu()                  # lengths stay on the host (pack_padded_sequence needs them there)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        run_iteration(trainer, batches[it % 4], cfg, it)
        it += 1
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        if s == args.steps - 1:
            ops.TIMER = ops.KernelTimer()                    # HIP events around the conv launches of the last timed step
        run_iteration(trainer, batches[it % 4], cfg, it)
        it += 1
    sync()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        images = per_gpu_batch * world * args.steps
        value = images / elapsed
        spans = timer.summary() if timer is not None else {}

        def total(pred):
            ent = {"launches": 0, "ms": 0.0, "flops": 0.0, "exec_flops": 0.0}
            for tag, v in spans.items():
                if pred(tag):
                    for k in ent:
                        ent[k] += v[k]
            return ent

        def rate(ent):
            return None if ent["ms"] <= 0 else {
                "launches": ent["launches"], "ms": round(ent["ms"], 3),
                "tflops": round(ent["flops"] / (ent["ms"] * 1e-3) / 1e12, 2),
                "frac_of_fp32_mfma_peak": round(ent["flops"] / (ent["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}

        dom = total(lambda t: t.endswith(DOMINANT))
        # the generator decode conv stack (8 AdaIN-ResBlock 3x3 convs, two 5x5 upsampling convs, fused heads)
        decode_stack = {"forward": rate(total(lambda t: t == "decode/conv_gemm_kernel")),
                        "backward": rate(total(lambda t: t.startswith("bwd:decode/")))}
        roof = None
        if dom and dom["ms"] > 0:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            executed = dom["exec_flops"] / (dom["ms"] * 1e-3) / 1e12
            # achieved: ALGORITHMIC flops (the direct convolution's) over the spans' time.  The 3x3 layers
            # run as Winograd F(2x2,3x3) and issue 2.25x fewer multiply-adds than that, so `executed` (what the matrix
            # cores actually did, transforms' time included in the spans) is the figure to hold against the MFMA roof.
            roof = {"bound": "mfma", "kernel": DOMINANT, "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "executed": round(executed, 2), "executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": DOMINANT_TRAFFIC_BYTES_PER_LAUNCH,
                    "launches_per_step": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                    "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3)}
        out = {
            "metric": "CelebA 128x128 training images/sec", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CelebA-shaped 128x128, per-GPU batch 16, fp32, full iteration "
                                   "(dis_update + gen_update + EMA + LR step), vgg_w=%g, 3x3 convs as Winograd tile %d" % (
                                       args.vgg_w, ops.WINOGRAD_TILE),
                       "image_size": IMAGE_SIZE, "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * world,
                       "parallelism": "dp%d" % world},
            "whole_step_tflops": round(ALGO_GFLOP_PER_IMAGE * value / 1e3, 2),
            "whole_step_frac_of_fp32_mfma_peak": round(ALGO_GFLOP_PER_IMAGE * value / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 4),
            "loss_dis_all": round(float(trainer.loss_dis_all.detach()), 5),
            "loss_gen_total": round(float(trainer.loss_gen_total.detach()), 5),
            "roofline": roof,
            "decode_conv_stack": decode_stack,
            "kernel_spans": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                 "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in spans.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(init_gen, init_dis, cfg)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
