#!/usr/bin/env python3
"""Throughput benchmark of the DWC-GAN training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4] [--scaling weak|strong] [--ledger out.json]

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or
plainly as ``python bench.py --gpus N``: the parent then starts N rank processes itself BEFORE anything touches a GPU
(the parent never initialises HIP) and relays rank 0's JSON line.

A step is one full training iteration of the reference loop body (reference train.py:102-111): dis_update + gen_update +
smooth_moving + update_learning_rate + update_attention_status, on a synthetic CelebA-shaped batch already resident in
HBM.  Workloads (BASELINE.json ``configs``; weak scaling — the per-GPU batch is fixed, the global batch is it times N):

    c1  configs[1]  128x128, 16 per GPU, fp32 end to end          <- the default and the headline metric
    c2  configs[2]  128x128, 128 per GPU, bf16 activations + bf16 MFMA conv path (fp32 accumulate / statistics / weights)
    c3  configs[3]  128x128, 64 per GPU, fp32 (global batch 512 on 8 GPUs)
    c4  configs[4]  256x256, 8 per GPU, fp32 (global batch 64 on 8 GPUs)

``--scaling strong`` fixes the GLOBAL batch at the configuration's 8-GPU value (c1 128, c2 1024, c3 512, c4 64; or
``--global-batch``) and gives every rank 1/N of it.  With N > 1 the record also carries BASELINE's own multi-GPU
configurations, c3 (configs[3]) and c4 (configs[4]), as legs under "also" (N == 1: c2).

Rank 0 prints ONE JSON line with the metric, the live roofline figure of the SINGLE kernel that owns the largest share of
the instrumented step (HIP events on the launch stream around every conv launch of the last timed step; executed flops
over that kernel's own dense peak) and, for N == 1, the CPU baseline (the oracle timed on the host cores by BASELINE.md
section 3's protocol).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "dwc-gan_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIGS = {
    "c1": {"image_size": 128, "per_gpu_batch": 16, "precision": "fp32", "strong_global_batch": 128,
           "label": "BASELINE configs[1]: CelebA-shaped 128x128, per-GPU batch 16, fp32"},
    "c2": {"image_size": 128, "per_gpu_batch": 128, "precision": "bf16", "strong_global_batch": 1024,
           "label": "BASELINE configs[2]: CelebA-shaped 128x128, per-GPU batch 128, bf16 activations + bf16 MFMA conv path "
                    "(fp32 accumulation, statistics, master weights)"},
    "c3": {"image_size": 128, "per_gpu_batch": 64, "precision": "fp32", "strong_global_batch": 512,
           "label": "BASELINE configs[3]: CelebA-shaped 128x128, per-GPU batch 64 (global 512 on 8 GPUs), fp32"},
    "c4": {"image_size": 256, "per_gpu_batch": 8, "precision": "fp32", "strong_global_batch": 64,
           "label": "BASELINE configs[4]: CelebA-HQ-shaped 256x256, per-GPU batch 8 (global 64 on 8 GPUs), fp32"},
}
CONFIGS["c5"] = CONFIGS["c4"]            # SURVEY.md section 8(d) numbers the same workloads C2..C5
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense v_mfma_f32_32x32x2_f32 / bf16 MFMA (no sparsity)
# BASELINE.md section 4 / SURVEY.md 8(d): necessary fwd+bwd conv+linear work per image per iteration at 128x128
# (x 1/4 at 64, x 4 at 256).  The step as built executes a little LESS than that (x_real's content code is encoded once
# instead of twice and D(x_real) evaluated once instead of twice): `whole_step_tflops` is therefore computed from the
# flops of the launches actually made (summed over the instrumented step), never from this constant.
ALGO_GFLOP_PER_IMAGE_128 = 569.6
DOMINANT = "conv_gemm_kernel"            # the forward / data-gradient GEMM family: one span = one conv call

# Launch kind (ops.KernelTimer.ledger: first word of the span detail) -> the kernel the C-ABI call runs, as
# (label, regex matching its rocprofv3 kernel-stats "Name", matrix pipe).  `{k}` = filter size.  Kinds whose call makes
# several launches of comparable weight (ring strips + fold) are labelled as such and never chosen as
# "the" roofline kernel; small helper launches inside a call (split-K / slab reduces, folds) are part of its span time.
KIND_KERNEL = {
    "fwd-x3": ("conv_halo_x3_kernel<{k}>", r"conv_halo_x3_kernel<{k}, .*, 3, 0>\(", "bf16x3"),
    "dgrad-x3": ("conv_halo_x3_kernel<{k}>", r"conv_halo_x3_kernel<{k}, .*, 3, 0>\(", "bf16x3"),
    "fwd-stem-x3": ("conv_stem_x3_kernel", r"conv_stem_x3_kernel", "bf16x3"),
    "dgrad-heads-stem-x3": ("conv_stem_x3_kernel", r"conv_stem_x3_kernel", "bf16x3"),
    "fwd-heads-nx3": ("conv_narrow_x3_kernel", r"conv_narrow_x3_kernel<", "bf16x3"),
    "dgrad-image-nx3": ("conv_narrow_x3_kernel", r"conv_narrow_x3_kernel<", "bf16x3"),
    # (template arguments: ..., PB, S2, KSP, NPL, RING -- S2: 1 = stride-2 forward, 2 = its data gradient; RING (r06, last): the border ring
    # of the data gradient inside the launch; conv_halo16_kernel: ..., PROBE, S2, RING)
    "fwd-x3s2": ("conv_halo_x3_kernel<2,S2>", r"conv_halo_x3_kernel<2, .*, 1, [12], 3, 0>\(", "bf16x3"),
    "dgrad-x3s2": ("conv_halo_x3_kernel<2,S2-dgrad>", r"conv_halo_x3_kernel<2, .*, 2, 1, 3, 0>\(", "bf16x3"),
    "dgrad-s2halo": ("conv_halo16_kernel<2,S2-dgrad>", r"conv_halo16_kernel<2, .*, 2, [01]>\(", "bf16"),
    "wgrad-x3": ("wgrad_x3_kernel<{k}>", r"wgrad_x3_kernel<{k}, .*, 3>\(", "bf16x3"),
    # r05: the same kernels with TWO f16 planes per operand (template argument NPL = 2, the last one): 3 MFMAs per fp32 multiply-add
    "fwd-h2": ("conv_halo_x3_kernel<{k},h2>", r"conv_halo_x3_kernel<{k}, .*, 2, [01]>\(", "f16x2"),
    "dgrad-h2": ("conv_halo_x3_kernel<{k},h2>", r"conv_halo_x3_kernel<{k}, .*, 2, [01]>\(", "f16x2"),
    "fwd-h2s2": ("conv_halo_x3_kernel<2,S2,h2>", r"conv_halo_x3_kernel<2, .*, 1, [12], 2, 0>\(", "f16x2"),
    "dgrad-h2s2": ("conv_halo_x3_kernel<2,S2-dgrad,h2>", r"conv_halo_x3_kernel<2, .*, 2, 1, 2, [01]>\(", "f16x2"),
    "wgrad-h2": ("wgrad_x3_kernel<{k},h2>", r"wgrad_x3_kernel<{k}, .*, 2>\(", "f16x2"),
    "fwd-halo": ("conv_halo16_kernel<{k}>", r"conv_halo16_kernel<{k}, ", "bf16"),
    "dgrad-halo": ("conv_halo16_kernel<{k}>", r"conv_halo16_kernel<{k}, ", "bf16"),
    "fwd-zeropad-halo": ("conv_halo16_kernel<{k}>", r"conv_halo16_kernel<{k}, ", "bf16"),
    "dgrad-zeropad-halo": ("conv_halo16_kernel<{k}>", r"conv_halo16_kernel<{k}, ", "bf16"),
    "fwd-s2halo": ("conv_halo16_kernel<2,S2>", r"conv_halo16_kernel<2, .*, 1, 0>\(", "bf16"),
    "wgrad-halo": ("wgrad_halo_kernel<{k}>", r"wgrad_halo_kernel<{k}, ", "bf16"),
    "fwd-stem": ("conv_stem_kernel", r"conv_stem_kernel", "bf16"),
    "dgrad-heads-stem": ("conv_stem_kernel", r"conv_stem_kernel", "bf16"),
    "fwd-heads-narrow": ("conv_narrow_kernel", r"conv_narrow(_persist)?_kernel", "bf16"),
    "dgrad-image-narrow": ("conv_narrow_kernel", r"conv_narrow(_persist)?_kernel", "bf16"),
    "wgrad-stem": ("smallk_wgrad_kernel", r"smallk_wgrad_kernel", "bf16"),
    "wgrad-heads-small": ("smallk_wgrad_kernel", r"smallk_wgrad_kernel", "bf16"),
    "wgrad-stem-x3": ("smallk_wgrad_x3_kernel", r"smallk_wgrad_x3_kernel", "bf16x3"),
    "wgrad-heads-small-x3": ("smallk_wgrad_x3_kernel", r"smallk_wgrad_x3_kernel", "bf16x3"),
    "dgrad-ring": ("conv_gemm_strips_kernel + fold_ring_kernel (multi-launch call)", r"(conv_)?gemm_strips_kernel|fold_ring_kernel", None),
}
# the fp32 im2col kernels with split-product inner products (ops._timed appends -g3 to the kind)
for _k, _lab, _rx in (("fwd", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("), ("dgrad", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("),
                      ("dgrad-image", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("), ("fwd-heads", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("),
                      ("dgrad-heads", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("), ("fwd-zeropad", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("),
                      ("dgrad-zeropad", "conv_gemm_kernel<X3>", r"conv_gemm_kernel<.*true>\("),
                      ("wgrad", "conv_wgrad_kernel<X3>", r"conv_wgrad_kernel<.*true>\("), ("wgrad-heads", "conv_wgrad_kernel<X3>", r"conv_wgrad_kernel<.*true>\("),
                      ):
    KIND_KERNEL[_k + "-g3"] = (_lab, _rx, "bf16x3")
GENERIC_KERNEL = {   # kinds on the generic im2col kernels: name depends on the activation precision
    "fwd": ("conv_gemm_kernel", "gemm_kernel_h"), "dgrad": ("conv_gemm_kernel", "gemm_kernel_h"),
    "fwd-heads": ("conv_gemm_kernel", "gemm_kernel_h"), "dgrad-heads": ("conv_gemm_kernel", "gemm_kernel_h"),
    "dgrad-image": ("conv_gemm_kernel", "gemm_kernel_h"), "fwd-zeropad": ("conv_gemm_kernel", "gemm_kernel_h"),
    "dgrad-zeropad": ("conv_gemm_kernel", "gemm_kernel_h"),
    "wgrad": ("conv_wgrad_kernel", "wgrad_kernel_h"), "wgrad-heads": ("conv_wgrad_kernel", "wgrad_kernel_h"),
}
PIPE_PEAK = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "f16x2": 2500.0}


def kernel_of_kind(kind_key, precision, split_generic=False):
    """(label, csv regex, pipe) of a ledger key such as "fwd-x3/k5".  The generic im2col kernels serve many layer shapes through
    several tile instantiations; `split_generic` labels them per launch kind ("conv_gemm_kernel[dgrad/k4]") so that a bag of
    unrelated small launches is not mistaken for one kernel."""
    kind, _, ksz = kind_key.partition("/")
    k = ksz[1:] if ksz else ""
    if kind in KIND_KERNEL:
        label, rx, pipe = KIND_KERNEL[kind]
        if split_generic and kind.endswith("-g3") and "multi-launch" not in label:
            label = label + "[" + kind_key + "]"
        return label.replace("{k}", k), rx.replace("{k}", k), pipe
    if kind in GENERIC_KERNEL:
        name = GENERIC_KERNEL[kind][1 if precision == "bf16" else 0]
        return (name + "[" + kind_key + "]") if split_generic else name, name + "<", "bf16" if precision == "bf16" else "fp32"
    return kind, None, None


def _shape_bytes(detail, elem):
    """Algorithmic bytes of one conv launch from its span detail "kind B<b> <h>x<w> <cin>><cout> k<k> [s<s>]": input once +
    output once + weights once (minimum traffic)."""
    try:
        w = detail.split()
        B = int(w[1][1:])
        H, W = (int(v) for v in w[2].split("x"))
        ci, co = (int(v) for v in w[3].split(">"))
        k = int(w[4][1:])
        st = int(w[5][1:]) if len(w) > 5 and w[5][0] == "s" else 1
    except (IndexError, ValueError):
        return 0
    return elem * B * H * W * ci + elem * B * (H // st) * (W // st) * co + 4 * ci * co * k * k


def kernel_table(ledger, precision, split_generic=False):
    """Ledger kinds folded per kernel: launches, ms, algorithmic / executed GFLOP and algorithmic bytes per launch, and the
    fraction of the kernel's pipe peak (executed flops / time / peak)."""
    elem = 2 if precision == "bf16" else 4
    rows = {}
    for key, ent in ledger.items():
        label, rx, pipe = kernel_of_kind(key, precision, split_generic)
        r = rows.setdefault(label, {"kernel": label, "csv_regex": rx, "pipe": pipe, "kinds": [], "launches": 0, "ms": 0.0,
                                    "flops": 0.0, "exec_flops": 0.0, "bytes": 0.0})
        r["kinds"].append(key)
        for f in ("launches", "ms", "flops", "exec_flops"):
            r[f] += ent[f]
        r["bytes"] += sum(_shape_bytes(d, elem) * n for d, n in ent["shapes"].items())
    out = []
    for r in rows.values():
        n, ms = max(r["launches"], 1), max(r["ms"], 1e-9)
        peak = PIPE_PEAK.get(r["pipe"])
        ex_tf = r["exec_flops"] / (ms * 1e-3) / 1e12
        out.append({"kernel": r["kernel"], "csv_regex": r["csv_regex"], "pipe": r["pipe"], "kinds": sorted(r["kinds"]),
                    "launches_per_step": r["launches"], "ms_per_step": round(r["ms"], 3), "avg_launch_us": round(ms * 1e3 / n, 2),
                    "algorithmic_gflop_per_launch": round(r["flops"] / n / 1e9, 3),
                    "executed_gflop_per_launch": round(r["exec_flops"] / n / 1e9, 3),
                    "algorithmic_mbytes_per_launch": round(r["bytes"] / n / 1e6, 2),
                    "algorithmic_tflops": round(r["flops"] / (ms * 1e-3) / 1e12, 2), "executed_tflops": round(ex_tf, 2),
                    "peak_tflops": peak, "frac": round(ex_tf / peak, 4) if peak else None})
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def workload_note(ops, precision):
    """How the fp32 path computes its inner products (part of config.workload: the record says what was measured)."""
    if precision != "fp32":
        return ""
    six = ("fp32-accurate bf16x3 split products on the bf16 MFMA (fp32 operands, results and accumulation; 6 of the 9 split terms, the "
           "rest < 2^-23 relative)")
    if ops.X3 >= 2 and ops.X3_PLANES == 2:
        return (", stride-1 3x3 / 5x5 and stride-2 4x4 layers (forward, data and weight gradient) as fp32-accurate TWO-plane f16 split "
                "products on the f16 MFMA (hi + lo 2^-11 per operand, per-tensor power-of-two scales, 3 products per multiply-add, "
                "matrix-core partial sums flushed to fp32 VALU sums every 16-25 k-steps), the remaining conv / linear products as "
                "three-plane bf16 split products (6 per multiply-add)")
    if ops.X3 >= 2:
        return ", every convolution / linear product as " + six
    return ", 5x5 convs and 3x3 data gradients as " + six if ops.X3 else ""


def run_iteration(trainer, batch, cfg, it):
    a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"], batch["label_src"],
         batch["label_trg"], cfg, it)
    trainer.dis_update(*a)
    trainer.gen_update(*a)
    trainer.smooth_moving()
    trainer.update_learning_rate()
    trainer.update_attention_status(it)


def cpu_baseline(gen_sd, dis_sd, cfg, image_size, batch, warmup, timed, thread_counts):
    """The CPU oracle (as_written: with the work the reference also does and discards) on the host cores, by the protocol of
    BASELINE.md section 3 / SURVEY.md 8(d): same synthetic inputs and shapes as the GPU workload's parity configuration
    (128x128, batch 16), `warmup` untimed + `timed` timed full iterations per thread count, MEDIAN images/s, best thread
    count reported.  A quick one-iteration probe at a quarter batch first drops thread counts that are clearly slower
    (more threads than memory channels can feed are slower on this graph), so that the default run stays bounded."""
    import torch
    from hipdwc import synth
    from oracle import dwcgan_oracle as orc

    def make():
        torch.manual_seed(4321)
        s = orc.OracleSolver(cfg, gen_sd, dis_sd, as_written=True)
        s.copy_nets()
        return s

    ncpu = os.cpu_count() or 1
    counts = sorted({min(max(1, t), ncpu) for t in thread_counts})
    # BASELINE.md section 3 names "N = all physical cores" (hardware threads / 2 on the SMT-2 hosts of this pool).  This graph
    # gets SLOWER beyond ~16 threads, so the protocol proper runs at the best count; the all-physical-cores figure is
    # measured by the one-iteration probe and reported beside it (`all_physical_cores_probe`).
    phys = max(1, ncpu // 2)
    probe_counts = sorted(set(counts) | {phys})
    probe = {}
    if len(probe_counts) > 1:
        small = synth.make_batch(max(1, batch // 4), image_size, seed=98)
        solver = make()
        torch.set_num_threads(counts[len(counts) // 2])
        solver.iteration(small, 0)                               # page everything in once
        for t in probe_counts:
            torch.set_num_threads(t)
            t0 = time.time()
            solver.iteration(small, 1)
            probe[t] = small["x_real"].shape[0] / (time.time() - t0)
        best_probe = max(probe.values())
        counts = [t for t in counts if probe[t] >= 0.8 * best_probe]
    full = synth.make_batch(batch, image_size, seed=99)
    results = {}
    for t in counts:
        torch.set_num_threads(t)
        solver = make()
        for i in range(warmup):
            solver.iteration(full, i)
        rates = []
        for i in range(timed):
            t0 = time.time()
            solver.iteration(full, warmup + i)
            rates.append(batch / (time.time() - t0))
        rates.sort()
        results[t] = rates[len(rates) // 2]
    best = max(results, key=results.get)
    return {"value": round(results[best], 4), "unit": "images/s", "cores": best, "kind": "port",
            "host_cpus": ncpu,
            "all_physical_cores_probe": ({"threads": phys, "value": round(probe[phys], 4), "unit": "images/s",
                                          "sample": "one iteration at batch %d" % max(1, batch // 4)} if phys in probe else None),
            "sample": "oracle (torch CPU fp32, as-written reference graph), %dx%d batch %d, %d warm-up + %d timed full iterations "
                      "per thread count, median; thread counts measured %s (one-iteration probe at batch %d: %s)" % (
                          image_size, image_size, batch, warmup, timed,
                          {k: round(v, 4) for k, v in results.items()}, max(1, batch // 4),
                          {k: round(v, 3) for k, v in probe.items()})}


def traffic_from_profiles(config, dominant, family="conv_gemm_family"):
    """HBM-side bytes per span of the dominant kernel family from the newest committed PMC summary for this workload
    (profiles/rNN_pmc_hbm_traffic*.json, written by benchmarks/pmc_summary.py from two rocprofv3 --pmc passes with the
    gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  bench.py cannot read PMCs itself: the figure is a property of
    the profiled build, so its source file and commit are printed beside it."""
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("config", "c1") != config or family not in d:
            continue
        best = (path, d)
    if best is None:
        return None, None
    path, d = best
    return int(d[family]["hbm_bytes_per_span_corrected"]), "%s (profiled at commit %s)" % (
        os.path.relpath(path, REPO), d.get("commit", "of round 1, 11e6a6c"))


def kernel_traffic_from_profiles(config, kernel_label):
    """HBM bytes per launch of ONE kernel instantiation (label like "conv_halo_x3_kernel<5>") from the newest committed PMC
    summary of this workload (`kernels_k` of profiles/rNN_pmc_hbm_traffic_<config>.json, benchmarks/pmc_summary.py)."""
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        ent = d.get("kernels_k", {}).get(kernel_label)
        if d.get("config", "c1") == config and ent:
            best = (path, d, ent)
    if best is None:
        return None, None
    path, d, ent = best
    return int(ent["hbm_bytes_per_launch_corrected"]), "%s (profiled at commit %s)" % (os.path.relpath(path, REPO), d.get("commit", "?"))


def spawn_ranks(args, timeout_s=3600):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as child processes of a parent that never touches
    the GPU, supervise ALL of them (the first non-zero exit or the overall timeout terminates the rest instead of leaving
    rank 0 waiting in a collective), relay rank 0's stdout (the JSON record) and every rank's stderr prefixed by its rank."""
    import tempfile
    import threading
    # file:// rendezvous created by the parent: no port to lose between probing and binding
    rdv = tempfile.NamedTemporaryFile(prefix="dwc_bench_rdv_", delete=False)
    rdv.close()
    os.unlink(rdv.name)
    procs, pumps = [], []

    def pump(stream, rank):
        for line in iter(stream.readline, b""):
            sys.stderr.write("[rank %d] %s" % (rank, line.decode(errors="replace")))
        stream.close()

    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), DWC_BENCH_INIT="file://" + rdv.name,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(p)
        t = threading.Thread(target=pump, args=(p.stderr, r), daemon=True)
        t.start()
        pumps.append(t)
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline, rc = time.time() + timeout_s, 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = max(abs(c) for c in codes)
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = abs(bad[0]) if bad else 124
            for p in procs:                      # exactly the children started above, by handle
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    for t in pumps:
        t.join(timeout=5)
    if os.path.exists(rdv.name):
        os.unlink(rdv.name)
    if out and out[0]:
        sys.stdout.write(out[0].decode())
        sys.stdout.flush()
    return rc


def measure(args, config_name, steps, warmup, dev, dist, rank, world, force_dp):
    """One workload: build the trainer, `warmup` untimed + `steps` timed full iterations between barriers / device
    synchronisations (MAX over ranks), HIP-event spans of the conv launches of the last timed step.  Returns
    (record dict on rank 0 else None, initial G / D state dicts for the CPU baseline)."""
    import contextlib
    import io

    import torch
    from hipdwc import host, ops, synth
    from solver import Solver

    conf = CONFIGS[config_name]
    image_size, precision = conf["image_size"], conf["precision"]
    per_gpu_batch = args.per_gpu_batch or conf["per_gpu_batch"]
    if args.scaling == "strong":          # fixed GLOBAL batch (the configuration's 8-GPU value), 1/N of it per rank
        gb = args.global_batch or conf["strong_global_batch"]
        if gb % world:
            raise SystemExit("--scaling strong: global batch %d is not a multiple of %d ranks" % (gb, world))
        per_gpu_batch = gb // world
    peak = MFMA_PEAK_TFLOPS[precision]
    ops.set_precision(precision)
    cfg = synth.make_config(image_size=image_size)           # shipped config, vgg_w = 0 (weights not obtainable offline)
    if args.vgg_w > 0:
        import tempfile
        from networks.networks import Vgg16
        vgg_dir = tempfile.mkdtemp(prefix="dwc_vgg_%d_" % rank)
        os.makedirs(os.path.join(vgg_dir, "models"))
        torch.manual_seed(777)
        torch.save(Vgg16().state_dict(), os.path.join(vgg_dir, "models", "vgg16.weight"))
        cfg["vgg_w"], cfg["vgg_model_path"] = args.vgg_w, vgg_dir
    torch.manual_seed(1234)                                  # same seed on every rank: identical initial weights
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    torch.cuda.manual_seed(1234 + rank)                      # per-rank style samples / dropout masks
    host.set_noise(host.DeviceNoise())
    if dist is not None:
        from hipdwc import dp
        dp.broadcast_module(trainer.gen)
        dp.broadcast_module(trainer.dis)
        trainer.copy_nets()
        trainer.enable_data_parallel()                       # flat gradient buckets, all-reduce overlapped with backward
    init_gen = {k: v.detach().cpu().clone() for k, v in trainer.gen.state_dict().items()}
    init_dis = {k: v.detach().cpu().clone() for k, v in trainer.dis.state_dict().items()}

    # fresh batch per iteration, pre-generated on the device (4 distinct, cycled)
    batches = [synth.make_batch(per_gpu_batch, image_size, seed=1000 * rank + i, device=dev) for i in range(4)]
    for b in batches:
        b["txt_lens"] = b["txt_lens"].cpu()                  # lengths stay on the host (pack_padded_sequence needs them there)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    it = 0
    for _ in range(warmup):
        run_iteration(trainer, batches[it % 4], cfg, it)
        it += 1
    sync()
    t0 = time.perf_counter()
    for s in range(steps):
        if s == steps - 1:
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
    if rank != 0:
        return None, None

    images = per_gpu_batch * world * steps
    value = images / elapsed
    spans = timer.summary() if timer is not None else {}

    def total(pred):
        ent = {"launches": 0, "ms": 0.0, "flops": 0.0, "exec_flops": 0.0}
        for tag, v in spans.items():
            if pred(tag):
                for k in ent:
                    ent[k] += v[k]
        return ent

    X3 = "conv_halo_x3_kernel"          # fp32 layers computed as bf16x3 split products: their matrix work is bf16 MFMA

    def rate(ent, roof_peak=None, exec_label="executed"):
        if ent["ms"] <= 0:
            return None
        pk = peak if roof_peak is None else roof_peak
        tf, ex = ent["flops"] / (ent["ms"] * 1e-3) / 1e12, ent["exec_flops"] / (ent["ms"] * 1e-3) / 1e12
        return {"launches": ent["launches"], "ms": round(ent["ms"], 3), "tflops": round(tf, 2),
                "frac_of_mfma_peak": round(tf / peak, 4), exec_label + "_tflops": round(ex, 2),
                exec_label + "_frac_of_mfma_peak": round(ex / pk, 4), "mfma_peak": pk}

    def stack(pred):
        """One conv stack, split by the matrix path its launches run on (fractions are against THAT path's dense peak;
        `tflops` is always the algorithmic fp32-equivalent rate)."""
        both = total(lambda t: pred(t) and (t.endswith(DOMINANT) or t.endswith(X3)))
        if both["ms"] <= 0:
            return None
        tf = both["flops"] / (both["ms"] * 1e-3) / 1e12
        out = {"launches": both["launches"], "ms": round(both["ms"], 3), "tflops": round(tf, 2)}
        native = rate(total(lambda t: pred(t) and t.endswith(DOMINANT)))
        split = rate(total(lambda t: pred(t) and t.endswith(X3)), MFMA_PEAK_TFLOPS["bf16"], "executed_bf16")
        if split is None:                      # one path only: the flat record of earlier rounds
            return native
        out["native_mfma"] = native
        out["split_bf16x3"] = split
        # the same stack in one line each way (fp32 path: its layers run as split products on the 16-bit matrix pipe):
        #   frac_of_mfma_peak            = matrix work EXECUTED (3 or 6 products per multiply-add) / the pipe's dense peak
        #   frac_algorithmic_of_pipe     = direct-convolution (useful) flops / the same peak
        #   algorithmic_vs_fp32_mfma_peak = useful flops / the fp32 MFMA peak the reference's arithmetic would be held to
        ex_all = total(lambda t: pred(t) and (t.endswith(DOMINANT) or t.endswith(X3)))
        if native is None:
            out["pipe_peak"] = MFMA_PEAK_TFLOPS["bf16"]
            out["frac_of_mfma_peak"] = split["executed_bf16_frac_of_mfma_peak"]
            out["frac_algorithmic_of_pipe"] = round(tf / MFMA_PEAK_TFLOPS["bf16"], 4)
            out["algorithmic_vs_fp32_mfma_peak"] = round(tf / MFMA_PEAK_TFLOPS["fp32"], 4)
            out["products_per_mac"] = round(ex_all["exec_flops"] / max(ex_all["flops"], 1.0), 2)
        return out

    ledger = timer.ledger() if timer is not None else {}
    table = kernel_table(ledger, precision, split_generic=True)
    if getattr(args, "ledger", None):
        with open(args.ledger if config_name == args.config else args.ledger + "." + config_name, "w") as f:
            json.dump({"config": config_name, "precision": precision, "per_gpu_batch": per_gpu_batch, "image_size": image_size,
                       "ledger": ledger, "kernels": kernel_table(ledger, precision),
                       "hbm_ops": timer.hbm_ledger() if timer is not None else {}}, f, indent=1)
    dom = total(lambda t: t.endswith(DOMINANT))
    x3 = total(lambda t: t.endswith(X3))
    # the generator decode conv stack (8 AdaIN-ResBlock 3x3 convs, two 5x5 upsampling convs, fused heads)
    decode_stack = {"forward": stack(lambda t: t.startswith("decode/")),
                    "backward": stack(lambda t: t.startswith("bwd:decode/") and "wgrad" not in t)}
    roof = roof_family = None
    if dom and dom["ms"] > 0:
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        executed = dom["exec_flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, traffic_source = traffic_from_profiles("c4" if config_name == "c5" else config_name, DOMINANT)
        # achieved: ALGORITHMIC flops (the direct convolution's) over the spans' time; `executed` = what the matrix
        # cores actually did in them (split-product launches: 3 or 6 products per multiply-add).
        roof_family = {"bound": "mfma", "kernel": DOMINANT + " family (every forward / data-gradient conv call on the im2col kernels)",
                       "achieved": round(achieved, 2), "peak": peak,
                       "unit": "TFLOP/s", "frac": round(executed / peak, 4), "algorithmic_frac": round(achieved / peak, 4),
                       "executed": round(executed, 2), "executed_frac": round(executed / peak, 4),
                       "traffic": traffic, "traffic_source": traffic_source,
                       "launches_per_step": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                       "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3)}
    # THE roofline record: the single kernel with the largest share of the instrumented step (multi-launch calls such as the
    # ring strips + fold are not "a kernel" and are skipped here; they stay visible in `kernels`).  `achieved` = flops the
    # matrix cores EXECUTED in that kernel / its span time; `peak` = the dense peak of the pipe it runs on; the
    # direct-convolution (fp32-equivalent) rate sits under `algorithmic_tflops`.
    single = [r for r in table if r["pipe"] and "multi-launch" not in r["kernel"] and r["executed_gflop_per_launch"] > 0]
    if single:
        r0 = single[0]
        # (profiles/*pmc_hbm_traffic*.json keys a kernel by name<first template argument>: "conv_halo_x3_kernel<3>")
        tr, tr_src = kernel_traffic_from_profiles("c4" if config_name == "c5" else config_name, r0["kernel"].replace(",h2>", ">"))
        # frac = matrix work the kernel EXECUTED / the dense peak of the pipe it occupies (utilisation of that pipe).  On the fp32 path
        # the kernels are split products: `products_per_mac` MFMAs of the 16-bit pipe per fp32 multiply-add (3: two f16 planes,
        # r05; 6: three bf16 planes, r02-r04), so the USEFUL fraction of that pipe is `frac_algorithmic_of_pipe` = frac /
        # products_per_mac, and `algorithmic_vs_fp32_mfma_peak` holds the useful rate against the fp32 MFMA peak (157.3 TF) that
        # fp32 arithmetic on the native instruction would be bounded by.
        roof = {"bound": "mfma", "kernel": r0["kernel"], "pipe": r0["pipe"], "achieved": r0["executed_tflops"], "peak": r0["peak_tflops"],
                "unit": "TFLOP/s", "frac": r0["frac"], "algorithmic_tflops": r0["algorithmic_tflops"],
                "products_per_mac": round(r0["executed_gflop_per_launch"] / max(r0["algorithmic_gflop_per_launch"], 1e-9), 2),
                "frac_algorithmic_of_pipe": round(r0["algorithmic_tflops"] / r0["peak_tflops"], 4),
                "algorithmic_vs_fp32_mfma_peak": round(r0["algorithmic_tflops"] / MFMA_PEAK_TFLOPS["fp32"], 4),
                "traffic": tr, "traffic_source": tr_src, "algorithmic_bytes_per_launch": int(r0["algorithmic_mbytes_per_launch"] * 1e6),
                "launches_per_step": r0["launches_per_step"], "avg_launch_us": r0["avg_launch_us"], "ms_per_step": r0["ms_per_step"],
                "share_of_step": round(r0["ms_per_step"] / (elapsed / steps * 1e3), 4),
                "algorithmic_gflop_per_launch": r0["algorithmic_gflop_per_launch"],
                "executed_gflop_per_launch": r0["executed_gflop_per_launch"], "launch_kinds": r0["kinds"]}
    roof_x3 = None
    if x3["ms"] > 0:
        # the fp32 layers that run as split products: six bf16 MFMAs per fp32 MFMA-equivalent (6 of the 9 partial products of
        # the three-way operand split; the dropped ones are below one fp32 rounding), so the roof that bounds them is the
        # dense bf16 MFMA peak and the executed rate is 6x the algorithmic one
        alg, ex = x3["flops"] / (x3["ms"] * 1e-3) / 1e12, x3["exec_flops"] / (x3["ms"] * 1e-3) / 1e12
        tr = traffic_from_profiles(config_name, X3, "split_bf16x3_family")
        roof_x3 = {"bound": "mfma", "kernel": X3, "achieved": round(alg, 2), "achieved_vs_fp32_mfma_peak": round(alg / MFMA_PEAK_TFLOPS["fp32"], 4),
                   "executed": round(ex, 2), "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                   "frac": round(ex / MFMA_PEAK_TFLOPS["bf16"], 4),
                   "traffic": tr[0], "traffic_source": tr[1],
                   "launches_per_step": x3["launches"],
                   "avg_launch_us": round(x3["ms"] * 1e3 / x3["launches"], 2),
                   "algorithmic_gflop_per_launch": round(x3["flops"] / x3["launches"] / 1e9, 3)}
    step_flops = sum(v["flops"] for v in spans.values())            # conv + linear-as-conv launches of ONE step, this rank
    step_tflops = step_flops * world / (elapsed / steps) / 1e12
    out = {
        "metric": "CelebA %dx%d training images/sec" % (image_size, image_size), "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "bf16" if precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": "%s, full iteration (dis_update + gen_update + EMA + LR step), vgg_w=%g%s" % (
                       conf["label"], args.vgg_w,
                       workload_note(ops, precision)),
                   "name": config_name, "image_size": image_size, "per_gpu_batch": per_gpu_batch,
                   "global_batch": per_gpu_batch * world, "parallelism": "dp%d" % world},
        # algorithmic (direct-convolution, fp32-equivalent) flops of the launches this step actually made (conv + linear
        # kernels; the text encoder's library GEMMs, < 0.01 %, are not counted) over the step time.  NOT a fraction of any one
        # roof on the fp32 path (split-product launches run on the 16-bit
        # pipe): the per-pipe fractions are `roofline`, `roofline_split_bf16x3` and `decode_conv_stack`.
        "whole_step_tflops": round(step_tflops, 2),
        "algorithmic_gflop_per_image_executed": round(step_flops / per_gpu_batch / 1e9, 2),
        "algorithmic_gflop_per_image_survey": round(ALGO_GFLOP_PER_IMAGE_128 * (image_size / 128.0) ** 2, 2),
        "loss_dis_all": round(float(trainer.loss_dis_all.detach()), 5),
        "loss_gen_total": round(float(trainer.loss_gen_total.detach()), 5),
        "roofline": roof,
        "roofline_family_native": roof_family,
        "roofline_split_bf16x3": roof_x3,
        "kernels": [{k: v for k, v in r.items() if k != "csv_regex"} for r in table[:12]],
        "decode_conv_stack": decode_stack,
        "kernel_spans": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                             "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in spans.items()},
    }
    if precision == "bf16":             # one pipe only: here the whole-step rate IS a fraction of a roof
        out["whole_step_frac_of_bf16_mfma_peak"] = round(step_tflops / (peak * world), 4)
    else:
        out["whole_step_fp32_equiv_tflops_vs_fp32_mfma_peak"] = round(step_tflops / (peak * world), 4)
    if getattr(trainer, "_reducers", None):
        out["data_parallel"] = {k: {"buckets": len(r.buckets), "bucket_mb": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in r.buckets],
                                    "all_reduces": r.calls, "launched_from_inside_backward": r.launched_early}
                                for k, r in trainer._reducers.items()}
    return out, (init_gen, init_dis)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c1", choices=sorted(CONFIGS),
                    help="workload = BASELINE.json configs[i] (c1 fp32 128^2 B16, the default and headline; c2 bf16 128^2 B128; "
                         "c3 fp32 128^2 B64; c4 (alias c5) fp32 256^2 B8)")
    ap.add_argument("--also", default=None,
                    help="comma list of further workloads measured after the main one in the same process and reported under "
                         "\"also\": {name: record} (default when the main workload is c1: c2 on one GPU, so that the bf16 "
                         "configuration's images/s and roofline sit in every driver-timed record; c3,c4 -- BASELINE's own "
                         "multi-GPU configurations -- on N > 1; 'none' switches it off)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default): fixed per-GPU batch; strong: fixed GLOBAL batch = the configuration's 8-GPU value "
                         "(c1 128, c2 1024, c3 512, c4 64; --global-batch overrides), split evenly over the ranks")
    ap.add_argument("--global-batch", type=int, default=None, help="global batch of --scaling strong")
    ap.add_argument("--ledger", default=None,
                    help="write the per-kind launch ledger of the instrumented step (launches, ms, algorithmic / executed flops, "
                         "shapes) and the per-kernel table to this JSON file (input of benchmarks/roofline_table.py)")
    ap.add_argument("--also-steps", type=int, default=10)
    ap.add_argument("--also-warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="CPU baseline: untimed iterations per thread count (BASELINE.md: >= 3)")
    ap.add_argument("--cpu-timed", type=int, default=5, help="CPU baseline: timed iterations per thread count (BASELINE.md: >= 5)")
    ap.add_argument("--cpu-threads", default="8,16,32,64",
                    help="CPU baseline: thread counts to sweep (r02 probe on the 256-hardware-thread GPU box, images/s at batch 4: "
                         "8: 1.10, 16: 1.40, 32: 0.80, 64: 0.33, 128: 0.11 -- this graph gets SLOWER beyond 16 threads)")
    ap.add_argument("--per-gpu-batch", type=int, default=None, help="development knob; overrides the config's per-GPU batch")
    ap.add_argument("--x3", type=int, default=None, choices=(0, 1, 2),
                    help="fp32 path: 0 = native fp32 MFMA kernels only; 1 (default) = 5x5 layers as fp32-accurate bf16x3 split "
                         "products on the bf16 MFMA; 2 = 3x3 layers too")
    ap.add_argument("--vgg-w", type=float, default=0.0,
                    help="development knob: perceptual-loss weight (the reference's shipped default is 0.1) with a RANDOMLY "
                         "initialised VGG16 (the trained weights cannot be fetched here); the contract workload is 0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    # stdout carries exactly ONE line, the JSON record: anything native libraries print there (RCCL's version banner is
    # written to the C stdout buffer and flushed at exit, i.e. AFTER the record) is diverted to stderr
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from hipdwc import ops, synth

    if args.x3 is not None:
        ops.X3 = args.x3
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dp = world == 1 and os.environ.get("DWC_FORCE_DP") == "1"      # development: drive the RCCL data-parallel path on ONE rank
    if world > 1 or force_dp:
        import torch.distributed as dist
        init = os.environ.get("DWC_BENCH_INIT")              # file:// rendezvous of the self-spawning parent
        if init and not force_dp:
            dist.init_process_group("nccl", device_id=dev, init_method=init, rank=rank, world_size=world)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            if force_dp:
                dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
            else:
                dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    out, init_sd = measure(args, args.config, args.steps, args.warmup, dev, dist, rank, world, force_dp)

    also = args.also
    if also is None:
        plain = args.config == "c1" and not force_dp and args.per_gpu_batch is None
        also = ("c2" if world == 1 else "c3,c4") if plain else "none"
    extra = {}
    for name in [n for n in also.split(",") if n and n != "none"]:
        if name not in CONFIGS:
            raise SystemExit("--also: unknown workload %r" % name)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        rec, _ = measure(args, name, args.also_steps, args.also_warmup, dev, dist, rank, world, force_dp)
        if rec is not None:
            rec.pop("kernel_spans", None)
            extra[name] = rec
    ops.set_precision(CONFIGS[args.config]["precision"])

    if rank == 0:
        if extra:
            out["also"] = extra
        if world == 1 and not args.no_cpu_baseline and not force_dp:
            import contextlib
            import io
            from solver import Solver
            # always the fp32 parity workload (128x128, batch 16): the reference's own arithmetic on the host cores
            ops.set_precision("fp32")
            cpu_cfg = synth.make_config(image_size=128)
            init_gen, init_dis = init_sd
            if CONFIGS[args.config]["image_size"] != 128:     # different architecture (D head sizes): fresh seeded weights
                torch.manual_seed(1234)
                with contextlib.redirect_stdout(io.StringIO()):
                    ref = Solver(cpu_cfg, torch.device("cpu"), None)
                init_gen, init_dis = ref.gen.state_dict(), ref.dis.state_dict()
            out["cpu_baseline"] = cpu_baseline(init_gen, init_dis, cpu_cfg, 128, 16, args.cpu_warmup, args.cpu_timed,
                                               [int(t) for t in args.cpu_threads.split(",") if t])
        os.write(record_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
