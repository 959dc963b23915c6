#!/usr/bin/env python3
"""HBM-side bytes per launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
TCC has 4 counter slots, they cost 3 + 2).  Produces profiles/rNN_pmc_hbm_traffic.json.

    cd /tmp && export TMPDIR=/tmp && cd $REPO
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python benchmarks/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_hbm_traffic_c1.json 3 225 c1 <commit>
    (arguments: fetch dir, write dir, output, iterations profiled, conv spans per iteration, bench --config, commit profiled,
     [split-bf16 spans per iteration]; benchmarks/collect_profiles.sh runs the whole set)

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter, by_first_arg=False):
    """Counter totals per kernel name with the template arguments dropped; ``by_first_arg``: per instantiation family, keyed
    "name<first template argument>" (e.g. "conv_halo_x3_kernel<5>": the filter size), the label bench.py's `roofline.kernel` uses."""
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    acc = {}
    for path in files:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                flat = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                short = flat.split("<")[0]
                if by_first_arg:
                    if "<" not in flat:
                        continue
                    short = short + "<" + flat.split("<", 1)[1].split(",")[0].split(">")[0].strip() + ">"
                if short.startswith("_ZN"):                   # templated kernels may come out mangled: _ZN12_GLOBAL__N_18in_applyI...
                    import re
                    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", short)
                    if m:
                        n0 = len(m.group(0))
                        short = short[n0:n0 + int(m.group(1))]
                ent = acc.setdefault(short, [0.0, 0])
                ent[0] += float(row["Counter_Value"])
                ent[1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    keep = ("conv_gemm_kernel", "wino_fused_kernel", "conv_gemm_batched_kernel", "conv_gemm_strips_kernel", "wino_input_kernel", "wino_output_kernel",
            "wino_dy_kernel", "wino_wgrad_reduce_kernel", "conv_wgrad_kernel", "wgrad_reduce_kernel", "in_stats_partial",
            "gemm_kernel_h", "conv_halo_kernel", "gemm_strips_kernel_h", "wgrad_kernel_h", "wgrad_reduce_kernel_h", "fold_ring_kernel_h",
            "fold_reflect_kernel_h", "splitk_reduce_kernel_h", "in_bwd_partial", "upsample2x_bwd_kernel", "ln_bwd_apply",
            "in_apply", "in_bwd_apply", "act_bwd_partial", "fold_reflect_kernel", "fold_ring_kernel", "upsample2x_fwd_kernel",
            "ln_apply", "adam_multi_kernel", "ema_multi_kernel", "lstm_step_fwd", "lstm_step_bwd",
            "conv_halo_x3_kernel", "wgrad_x3_kernel", "x3_wgrad_reduce_kernel", "wgrad_halo_kernel", "wgrad_halo_reduce_kernel",
            "conv_narrow_kernel", "conv_stem_kernel", "smallk_wgrad_kernel", "smallk_reduce_kernel",
            "conv_halo16_kernel", "fold_reflect_kernel_h8", "lstm_seq_fwd", "lstm_seq_bwd", "weight_refresh_multi_kernel",
            "adv_tail_fwd_kernel", "adv_tail_bwd_kernel", "maxpool2_fwd_kernel", "in_stats_final", "ln_bwd_partial")
    res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline",
           "note": "units KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
                   "separate passes; summarised by benchmarks/pmc_summary.py",
           "kernels": {}}
    for k in keep:
        if k not in fetch or k not in write:
            continue
        f_avg, w_avg = fetch[k][0] / fetch[k][1], write[k][0] / write[k][1]
        res["kernels"][k] = {"FETCH_SIZE_KiB_avg_per_launch": round(f_avg, 1), "launches": fetch[k][1],
                             "WRITE_SIZE_KiB_avg_per_launch": round(w_avg, 1),
                             "hbm_bytes_per_launch_corrected": int((2 * f_avg + w_avg) * 1024)}
    # per instantiation family ("name<first template argument>"): what bench.py's single-kernel `roofline.traffic` reads
    fetch_k, write_k = per_kernel(fetch_dir, "FETCH_SIZE", True), per_kernel(write_dir, "WRITE_SIZE", True)
    res["kernels_k"] = {}
    for k in sorted(fetch_k):
        if k in write_k and k.split("<")[0] in keep:
            f_avg, w_avg = fetch_k[k][0] / fetch_k[k][1], write_k[k][0] / write_k[k][1]
            res["kernels_k"][k] = {"FETCH_SIZE_KiB_avg_per_launch": round(f_avg, 1), "launches": fetch_k[k][1],
                                   "WRITE_SIZE_KiB_avg_per_launch": round(w_avg, 1),
                                   "hbm_bytes_per_launch_corrected": int((2 * f_avg + w_avg) * 1024)}
    # the forward / data-gradient GEMM family as bench.py's roofline spans see it: one span = one conv call, which for a
    # 3x3 layer is input transform + batched GEMM + output transform (+ ring strips and fold for a data gradient)
    family = ("conv_gemm_kernel", "wino_fused_kernel", "conv_gemm_batched_kernel", "conv_gemm_strips_kernel",
              "wino_input_kernel", "wino_output_kernel", "fold_ring_kernel",
              # bf16 path (bench --config c2): im2col GEMM, halo-tiled kernel, ring strips + folds, split-K reduce
              "gemm_kernel_h", "conv_halo_kernel", "conv_halo16_kernel", "conv_narrow_kernel", "conv_stem_kernel", "gemm_strips_kernel_h",
              "fold_ring_kernel_h", "fold_reflect_kernel_h", "fold_reflect_kernel_h8", "splitk_reduce_kernel_h")
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    spans_per_step = int(sys.argv[5]) if len(sys.argv) > 5 else 236
    res["config"] = sys.argv[6] if len(sys.argv) > 6 else "c1"
    res["commit"] = sys.argv[7] if len(sys.argv) > 7 else "unknown"
    total = sum((2 * fetch[k][0] + write[k][0]) * 1024 for k in family if k in fetch and k in write)
    res["conv_gemm_family"] = {"kernels": list(family), "iterations_profiled": steps, "spans_per_iteration": spans_per_step,
                               "hbm_bytes_per_span_corrected": int(total / steps / spans_per_step)}
    # the fp32 layers that run as split-bf16 products (bench.py's `roofline_split_bf16x3`): one span = one conv_halo_x3_kernel
    # launch (the ring of a data gradient is counted with the family above, its kernels are shared)
    x3_spans = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    if x3_spans and "conv_halo_x3_kernel" in fetch and "conv_halo_x3_kernel" in write:
        t3 = (2 * fetch["conv_halo_x3_kernel"][0] + write["conv_halo_x3_kernel"][0]) * 1024
        res["split_bf16x3_family"] = {"kernels": ["conv_halo_x3_kernel"], "iterations_profiled": steps, "spans_per_iteration": x3_spans,
                                      "hbm_bytes_per_span_corrected": int(t3 / steps / x3_spans)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["conv_gemm_family"], indent=1))


if __name__ == "__main__":
    main()
