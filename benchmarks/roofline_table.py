#!/usr/bin/env python3
"""Machine-readable per-kernel roofline table of one bench run.

    python benchmarks/roofline_table.py <rocprofv3 kernel_stats.csv> <bench.py --ledger json> <iterations profiled> <out.json> [commit]

Joins, per kernel, what rocprofv3 measured (calls, average duration of THAT kernel's launches over the whole run) with what
the launch ledger of the same run knows (bench.py --ledger: per launch kind -- first word of the span detail + filter size --
the launches, algorithmic and executed flops, problem shapes of ONE training iteration).  A row:

    kernel, csv_names, pipe, launches_per_iter (rocprof), avg_us (rocprof), ms_per_iter (rocprof),
    algorithmic_gflop_per_launch, executed_gflop_per_launch, algorithmic_mbytes_per_launch (input + output + weights once),
    executed_tflops = executed flops per iteration / rocprof time per iteration, peak_tflops (dense peak of the pipe the kernel
    runs on: fp32 MFMA 157.3, bf16 MFMA 2500; "bf16x3" = fp32 layers as six bf16 products per multiply-add, "f16x2" = as three f16
    products per multiply-add (r05), same 2500 peak),
    frac = executed_tflops / peak_tflops, algorithmic_tflops (direct-convolution fp32-equivalent rate)

Kernels the ledger does not know (norms, pointwise, reduces, stock torch) are listed with their rocprof time and
bound "hbm" / null flops, so that the table accounts for the whole iteration.  Multi-launch calls (Winograd) appear once per
kernel of the chain, each carrying the call's flops only on its product kernel.
"""
import csv
import json
import re
import sys

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))


def main():
    stats_csv, ledger_json, iters, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    commit = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    with open(ledger_json) as f:
        led = json.load(f)
    rows_csv = list(csv.DictReader(open(stats_csv)))
    total_ns = sum(int(r["TotalDurationNs"]) for r in rows_csv)
    used = set()
    table = []
    # ledger rows whose kernels overlap in the csv (the Winograd weight gradient runs its products on conv_wgrad_kernel, which the
    # generic weight gradients use too) are merged into one row: a kernel's time is never attributed twice
    merged = []
    for k in led["kernels"]:
        rx = k.get("csv_regex")
        names = {r["Name"] for r in rows_csv if rx and re.search(rx, r["Name"])}
        for m in merged:
            if names & m["_names"]:
                m["_names"] |= names
                m["kernel"] += " + " + k["kernel"]
                m["csv_regex"] = "(%s)|(%s)" % (m["csv_regex"], rx)
                m["kinds"] = sorted(set(m["kinds"]) | set(k["kinds"]))
                n0, n1 = m["launches_per_step"], k["launches_per_step"]
                for f in ("algorithmic_gflop_per_launch", "executed_gflop_per_launch", "algorithmic_mbytes_per_launch", "avg_launch_us"):
                    m[f] = (m[f] * n0 + k[f] * n1) / max(n0 + n1, 1)
                m["launches_per_step"] = n0 + n1
                if m.get("pipe") != k.get("pipe"):
                    m["pipe"], m["peak_tflops"] = m.get("pipe") or k.get("pipe"), m.get("peak_tflops") or k.get("peak_tflops")
                break
        else:
            e = dict(k)
            e["_names"] = names
            merged.append(e)
    for k in merged:
        rx = k.get("csv_regex")
        match = [r for r in rows_csv if r["Name"] in k["_names"]]
        # a multi-launch call's flops sit on its product kernels only
        prod = [r for r in match if not re.search(r"wino_(input|output|dy)_kernel|wino_wgrad_reduce|fold_ring|splitk_reduce", r["Name"])] or match
        ns = sum(int(r["TotalDurationNs"]) for r in match)
        calls = sum(int(r["Calls"]) for r in prod)
        used.update(r["Name"] for r in match)
        ms_iter = ns / 1e6 / iters
        n_iter = calls / iters
        peak = k.get("peak_tflops")
        ex_fl = k["executed_gflop_per_launch"] * k["launches_per_step"] * 1e9
        al_fl = k["algorithmic_gflop_per_launch"] * k["launches_per_step"] * 1e9
        ex_tf = ex_fl / (ms_iter * 1e-3) / 1e12 if ms_iter > 0 else None
        table.append({
            "kernel": k["kernel"], "csv_names": sorted({r["Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
                                                        for r in match}),
            "pipe": k["pipe"], "launch_kinds": k["kinds"], "launches_per_iter": round(n_iter, 2),
            "avg_us": round(ns / 1e3 / max(calls, 1), 2), "ms_per_iter": round(ms_iter, 3),
            "share_of_kernel_time": round(ns / total_ns, 4),
            "algorithmic_gflop_per_launch": round(al_fl / max(n_iter, 1e-9) / 1e9, 3),
            "executed_gflop_per_launch": round(ex_fl / max(n_iter, 1e-9) / 1e9, 3),
            "algorithmic_mbytes_per_launch": k["algorithmic_mbytes_per_launch"],
            "bound": "mfma" if peak else None, "peak_tflops": peak,
            "executed_tflops": None if ex_tf is None else round(ex_tf, 2),
            "algorithmic_tflops": None if ms_iter <= 0 else round(al_fl / (ms_iter * 1e-3) / 1e12, 2),
            "frac": None if (ex_tf is None or not peak) else round(ex_tf / peak, 4),
            # executed / algorithmic multiply-adds (split products: 3 = two f16 planes, 6 = three bf16 planes) and the USEFUL fraction
            "products_per_mac": None if al_fl <= 0 else round(ex_fl / al_fl, 2),
            "frac_algorithmic_of_pipe": None if (ms_iter <= 0 or not peak) else round(al_fl / (ms_iter * 1e-3) / 1e12 / peak, 4),
            "span_avg_us_hip_events": k["avg_launch_us"]})
    # HBM-bound op families (r06): bench.py's ledger carries their calls and ALGORITHMIC bytes per iteration (each operand read once, each
    # result written once: hipdwc.ops._hbm); all kernels of a family together give its effective bandwidth against the 8 TB/s spec and
    # the 6.3 TB/s a float4 copy reaches (MI355X_MICROARCH.md)
    HBM_FAMILIES = {
        "instnorm_fwd": r"in_resident_fwd|in_stats_partial|in_stats_final|in_apply",
        "instnorm_bwd": r"in_resident_bwd|in_bwd_partial|in_bwd_final|in_bwd_apply",
        "layernorm_fwd": r"ln_stats_partial|ln_stats_final|ln_apply",
        "layernorm_bwd": r"ln_bwd_partial|ln_bwd_final|ln_bwd_apply",
        "upsample2x_fwd": r"upsample2x_fwd_kernel", "upsample2x_bwd": r"upsample2x_bwd_kernel",
        "avgpool2_fwd": r"avgpool2_fwd_kernel", "avgpool2_bwd": r"avgpool2_bwd_kernel",
        "act_bwd_bias": r"act_bwd_partial|colsum_final",
        "adam_multi": r"adam_multi_kernel", "ema_multi": r"ema_multi_kernel",
    }
    for fam, ent in (led.get("hbm_ops") or {}).items():
        rx = HBM_FAMILIES.get(fam)
        match = [r for r in rows_csv if rx and re.search(rx, r["Name"]) and r["Name"] not in used]
        if not match:
            continue
        ns = sum(int(r["TotalDurationNs"]) for r in match)
        calls = sum(int(r["Calls"]) for r in match)
        used.update(r["Name"] for r in match)
        ms_iter = ns / 1e6 / iters
        gbps = ent["bytes"] / (ms_iter * 1e-3) / 1e9 if ms_iter > 0 else None
        def short(name):
            name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
            m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
            return name[len(m.group(0)):len(m.group(0)) + int(m.group(1))] if m else name
        table.append({"kernel": fam + " (" + ", ".join(sorted({short(r["Name"]) for r in match})) + ")",
                      "pipe": None, "bound": "hbm", "launches_per_iter": round(calls / iters, 2), "op_calls_per_iter": ent["calls"],
                      "avg_us": round(ns / 1e3 / calls, 2), "ms_per_iter": round(ms_iter, 3), "share_of_kernel_time": round(ns / total_ns, 4),
                      "algorithmic_mbytes_per_iter": round(ent["bytes"] / 1e6, 1), "gbps": None if gbps is None else round(gbps, 1),
                      "frac_of_hbm_spec_8TBs": None if gbps is None else round(gbps / 8000.0, 4),
                      "frac_of_hbm_achievable_6p3TBs": None if gbps is None else round(gbps / 6300.0, 4)})
    rest = {}
    for r in rows_csv:
        if r["Name"] in used:
            continue
        short = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", short)
        if m:
            short = short[len(m.group(0)):len(m.group(0)) + int(m.group(1))]
        e = rest.setdefault(short, [0, 0])
        e[0] += int(r["TotalDurationNs"])
        e[1] += int(r["Calls"])
    for name, (ns, calls) in sorted(rest.items(), key=lambda kv: -kv[1][0]):
        if ns / total_ns < 0.002:
            continue
        table.append({"kernel": name, "pipe": None, "bound": "hbm/latency", "launches_per_iter": round(calls / iters, 2),
                      "avg_us": round(ns / 1e3 / calls, 2), "ms_per_iter": round(ns / 1e6 / iters, 3),
                      "share_of_kernel_time": round(ns / total_ns, 4)})
    table.sort(key=lambda r: -r["ms_per_iter"])
    res = {"config": led["config"], "precision": led["precision"], "per_gpu_batch": led["per_gpu_batch"], "image_size": led["image_size"],
           "commit": commit, "iterations_profiled": iters, "kernel_ms_per_iter": round(total_ns / 1e6 / iters, 3),
           "peaks_tflops": {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "f16x2": 2500.0},
           "source": {"kernel_stats": stats_csv.split("/")[-1], "ledger": ledger_json.split("/")[-1]}, "kernels": table}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for r in table[:14]:
        print("%-58s %8.3f ms/iter %8.1f us  frac %s" % (r["kernel"][:58], r["ms_per_iter"], r["avg_us"], r.get("frac")))
    hb = [r for r in table if r.get("bound") == "hbm"]
    print("HBM-bound op families: %.2f ms per iteration" % sum(r["ms_per_iter"] for r in hb))
    for r in hb:
        print("%-58s %8.3f ms/iter %8.1f GB/s  %.2f of 8 TB/s" % (r["kernel"][:58], r["ms_per_iter"], r["gbps"], r["frac_of_hbm_spec_8TBs"]))


if __name__ == "__main__":
    main()
