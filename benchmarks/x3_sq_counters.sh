#!/bin/bash
# SQ counters of the split-product kernels (conv_halo_x3_kernel, wgrad_x3_kernel) on the c1 layer shapes, batches 16 and 48.
# Counters only (no other trace domain); run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash benchmarks/x3_sq_counters.sh <tag>'        -> gpurun_out/<tag>_x3_sq_counters.json
TAG=${1:-r04}
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/psq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d /tmp/psq -o q -- python3 $R/benchmarks/x3_kernels_bench.py 16 48 > /tmp/psq.out 2> /tmp/psq.err
python3 - <<PY > $R/gpurun_out/${TAG}_x3_sq_counters.json 2> $R/gpurun_out/${TAG}_x3_sq_counters.err
import csv, glob, collections, json, re
f = glob.glob('/tmp/psq/**/*counter_collection.csv', recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    m = re.search(r'(conv_halo_x3_kernel|wgrad_x3_kernel)<([^>]*)>', k)
    if not m:
        continue
    key = "%s<%s> grid %s wg %s" % (m.group(1), m.group(2), r['Grid_Size'], r['Workgroup_Size'])
    agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
out = collections.OrderedDict()
for key, v in agg.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    wave = c.get('SQ_WAVE_CYCLES', 0.0)
    ent = {"dispatches": len(next(iter(v.values()))), "counters_avg_per_dispatch": {n: round(x) for n, x in c.items()}}
    if wave:
        ent["of_wave_cycles"] = {n: round(c[n] / wave, 4) for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if n in c}
        ent["mfma_busy_over_wave_cycles"] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wave, 4)
    if c.get('SQ_BUSY_CYCLES'):
        # MFMA-busy share of the SIMDs' cycles (profiles/README.md, r03 notes: SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES))
        ent["mfma_busy_share_of_simd_cycles"] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32.0 * c['SQ_BUSY_CYCLES']), 4)
    out[key] = ent
print(json.dumps({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -- python3 benchmarks/x3_kernels_bench.py 16 48", "kernels": out}, indent=1))
PY
head -5 /tmp/psq.err >> $R/gpurun_out/${TAG}_x3_sq_counters.err
cp /tmp/psq.out $R/gpurun_out/${TAG}_x3_sq_counters_bench_output.txt
