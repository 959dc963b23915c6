#!/bin/bash
# per-kernel times of benchmarks/in_multipass_probe.py: bash benchmarks/in_probe.sh <tag>   -> gpurun_out/in_probe_<tag>_trace.csv
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/inp_$1 -o s -- python3 benchmarks/in_multipass_probe.py 16 > /dev/null 2> /tmp/inp_$1.err
cp $(find /tmp/inp_$1 -name "*kernel_trace.csv" | head -1) $R/gpurun_out/in_probe_$1_trace.csv
