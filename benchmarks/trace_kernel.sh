#!/bin/bash
# per-dispatch durations and grid sizes of the kernels whose name contains $2, one bench configuration ($1): gpurun_out/trace_$2.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -o t -- python3 bench.py --config $1 --also "" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/ptr.err
python3 - "$2" <<'PY' > $R/gpurun_out/trace_$2.txt
import csv, glob, sys, collections
f = glob.glob('/tmp/ptr/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r['Kernel_Name']:
        key = (r['Kernel_Name'][:60], r.get('Grid_Size') or (r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '?') + 'x' + r.get('Grid_Size_Z', '?')))
        agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s grid %10s  n %3d  avg %8.1f us  total %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
PY
