#!/bin/bash
# SQ counters of the matrix-core kernels INSIDE a training step ("start from counters"): one counter pass over
# bench.py at c1 / c2 (counters only, no other trace domain); per kernel instantiation and grid: wave cycles, waits, MFMA-busy share.
#   gpurun --timeout 900 -- 'bash benchmarks/step_sq_counters.sh <tag> <config>'   -> gpurun_out/<tag>_step_sq_counters_<config>.json
TAG=${1:-r06}
CFG=${2:-c2}
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/psq_$CFG
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d /tmp/psq_$CFG -o q -- python3 bench.py --config $CFG --also "" --steps 2 --warmup 1 --no-cpu-baseline > /tmp/psq_$CFG.out 2> /tmp/psq_$CFG.err
python3 - <<PY > $R/gpurun_out/${TAG}_step_sq_counters_${CFG}.json 2> $R/gpurun_out/${TAG}_step_sq_counters_${CFG}.err
import csv, glob, collections, json, re
f = glob.glob('/tmp/psq_$CFG/**/*counter_collection.csv', recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    m = re.search(r'(conv_halo16_kernel|wgrad_halo_kernel|conv_halo_x3_kernel|wgrad_x3_kernel|conv_narrow_kernel|conv_narrow_persist_kernel|conv_stem_kernel|smallk_wgrad_kernel|gemm_kernel_h|conv_gemm_kernel)(<[^>]*>)?', k)
    if not m:
        continue
    key = "%s%s grid %s wg %s" % (m.group(1), m.group(2) or "", r['Grid_Size'], r['Workgroup_Size'])
    agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
out = collections.OrderedDict()
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', [0]))):
    c = {n: sum(x) / len(x) for n, x in v.items()}
    wave = c.get('SQ_WAVE_CYCLES', 0.0)
    ent = {"dispatches": len(next(iter(v.values()))), "counters_avg_per_dispatch": {n: round(x) for n, x in c.items()}}
    if wave:
        ent["of_wave_cycles"] = {n: round(c[n] / wave, 4) for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if n in c}
    if c.get('SQ_BUSY_CYCLES'):
        # MFMA-busy share of the SIMDs' cycles (profiles/README.md, r03 notes: SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES))
        ent["mfma_busy_share_of_simd_cycles"] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32.0 * c['SQ_BUSY_CYCLES']), 4)
    out[key] = ent
print(json.dumps({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -- python3 bench.py --config $CFG --also '' --steps 2 --warmup 1 --no-cpu-baseline",
                  "note": "3 iterations profiled; kernels keyed by instantiation and grid, sorted by SQ busy cycles", "kernels": out}, indent=1))
PY
head -3 /tmp/psq_$CFG.err >> $R/gpurun_out/${TAG}_step_sq_counters_${CFG}.err
