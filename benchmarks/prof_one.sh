#!/bin/bash
# per-kernel time of one bench configuration: bash benchmarks/prof_one.sh <config> <tag>   -> gpurun_out/prof_<tag>.csv
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pone_$2 -o s -- python3 bench.py --config $1 --also "" --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> /tmp/pone_$2.err
cp $(find /tmp/pone_$2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_$2.csv
