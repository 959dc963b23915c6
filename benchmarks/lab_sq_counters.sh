#!/bin/bash
# SQ counters of the halo-tiled bf16 convolution kernels before / after the r03 rewrite, same process, same operands
# (benchmarks/halo_lab.hip runs conv_halo_kernel = compiler-scheduled 32x32x16 and conv_halo16_kernel = hand-scheduled 16x16x32, 8-wave
# and two-workgroups-per-CU forms, on every shape).  Counters only (no other trace domain); run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash benchmarks/lab_sq_counters.sh'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d /tmp/psq -o q -- $R/benchmarks/bin/halo_lab 4 > /dev/null 2> /tmp/psq.err
python3 - <<'PY' > $R/gpurun_out/r03_halo_sq_counters.json 2> $R/gpurun_out/r03_halo_sq_counters.err
import csv, glob, collections, json, re
f = glob.glob('/tmp/psq/**/*counter_collection.csv', recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    m = re.search(r'(conv_halo_kernel|conv_halo16_kernel|wgrad_halo_kernel)<([^>]*)>', k)
    if not m:
        continue
    key = "%s<%s> grid %s" % (m.group(1), m.group(2), r['Grid_Size'])
    agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
out = collections.OrderedDict()
for key, v in agg.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    wave = c.get('SQ_WAVE_CYCLES', 0.0)
    busy = c.get('SQ_BUSY_CYCLES', 0.0)
    ent = {"dispatches": len(next(iter(v.values()))), "counters_avg_per_dispatch": {n: round(x) for n, x in c.items()}}
    if wave:
        ent["of_wave_cycles"] = {n: round(c[n] / wave, 4) for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if n in c}
        # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD; a wave lives on one SIMD: busy share of the SIMDs' time ~ MFMA busy / (wave cycles / waves per SIMD)
        ent["mfma_busy_over_wave_cycles"] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wave, 4)
    out[key] = ent
print(json.dumps({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -- benchmarks/bin/halo_lab 4", "kernels": out}, indent=1))
PY
head -5 /tmp/psq.err >> $R/gpurun_out/r03_halo_sq_counters.err
