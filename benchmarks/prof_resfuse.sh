cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -o s -- python3 bench.py --config c2 --also "" --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> /tmp/pc2.err
cp $(find /tmp/pc2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03k_c2_kernel_stats.csv
DWC_RES_FUSE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2b -o s -- python3 bench.py --config c2 --also "" --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> /tmp/pc2b.err
cp $(find /tmp/pc2b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03k_c2_kernel_stats_nofuse.csv
