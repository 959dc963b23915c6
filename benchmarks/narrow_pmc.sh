#!/bin/bash
# FETCH_SIZE and SQ counters of the image-heads kernel in isolation (benchmarks/narrow_bench.py), persistent and block-per-workgroup form.
#   gpurun --timeout 900 -- 'bash benchmarks/narrow_pmc.sh'   -> gpurun_out/r06_narrow_pmc.txt
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
cd $R
OUT=$R/gpurun_out/r06_narrow_pmc.txt
: > $OUT
for P in 1 0; do
  export DWC_NARROW_PERSIST=$P
  for SET in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    D=/tmp/npmc_${P}_$(echo $SET | cut -c1-5)
    rm -rf $D
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -o n -- python3 benchmarks/narrow_bench.py bf16 384 > /tmp/npmc.out 2> /tmp/npmc.err
    python3 - "$D" "$P" >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_narrow' not in k:
        continue
    agg[k.split('(')[0][-40:] + " grid " + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print("persist=%s %s: %s" % (sys.argv[2], k, {n: round(sum(x) / len(x)) for n, x in v.items()}), "dispatches", len(next(iter(v.values()))))
PY
  done
done
cat $OUT
