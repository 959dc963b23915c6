#!/bin/bash
# One-shot collection of the profiles/ set on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash benchmarks/collect_profiles.sh r02 <commit>'
# Writes gpurun_out/profiles_<round>/ -- copy its files into profiles/ afterwards.  Kernel statistics and the two counter
# passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters never together with other traces).
set -u
ROUND=${1:-r02}
COMMIT=${2:-unknown}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
for CFG in c1 c2; do
  S=$OUT/${CFG}_stats; F=$OUT/${CFG}_fetch; W=$OUT/${CFG}_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $S -o s -- python3 bench.py --config $CFG --also "" --steps 6 --warmup 3 --no-cpu-baseline \
      --ledger $OUT/${ROUND}_ledger_${CFG}.json > $OUT/${ROUND}_bench_${CFG}_under_rocprof.json 2> $OUT/${CFG}_stats.err
  cp $(find $S -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_bench_${CFG}_kernel_stats.csv
  # per-kernel roofline table: rocprof's average durations x the flop ledger of the same run (9 iterations profiled)
  python3 benchmarks/roofline_table.py $OUT/${ROUND}_bench_${CFG}_kernel_stats.csv $OUT/${ROUND}_ledger_${CFG}.json 9 \
      $OUT/${ROUND}_roofline_table_${CFG}.json $COMMIT
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $F -o f -- python3 bench.py --config $CFG --also "" --steps 2 --warmup 1 --no-cpu-baseline \
      > $OUT/${CFG}_fetch.json 2> $OUT/${CFG}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W -o w -- python3 bench.py --config $CFG --also "" --steps 2 --warmup 1 --no-cpu-baseline \
      > $OUT/${CFG}_write.json 2> $OUT/${CFG}_write.err
  SPANS=$(python3 -c "import json;d=json.load(open('$OUT/${CFG}_fetch.json'));print(d['roofline_family_native']['launches_per_step'])")
  X3=$(python3 -c "import json;d=json.load(open('$OUT/${CFG}_fetch.json'));r=d.get('roofline_split_bf16x3');print(r['launches_per_step'] if r else 0)")
  python3 benchmarks/pmc_summary.py $F $W $OUT/${ROUND}_pmc_hbm_traffic_${CFG}.json 3 $SPANS $CFG $COMMIT $X3
  rm -rf $S $F $W            # the raw traces exceed what gpurun merges back
done
ls -la $OUT
