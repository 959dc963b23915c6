#!/bin/bash
# A/B of one environment switch by per-kernel time (rocprofv3 --stats), not by wall clock: bash benchmarks/prof_ab.sh <config> <VAR>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; VAR=$2
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab1 -o s -- python3 bench.py --config $CFG --also "" --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> /tmp/pab1.err
cp $(find /tmp/pab1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ab_${CFG}_on.csv
export $VAR=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab0 -o s -- python3 bench.py --config $CFG --also "" --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> /tmp/pab0.err
cp $(find /tmp/pab0 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ab_${CFG}_off.csv
