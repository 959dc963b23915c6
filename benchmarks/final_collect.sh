set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash benchmarks/step_sq_counters.sh r06 c1
bash benchmarks/step_sq_counters.sh r06 c2
python3 benchmarks/torch_kernel_sites.py c1 > gpurun_out/r06_torch_kernel_sites_c1.txt 2>/dev/null
python3 benchmarks/torch_op_sites.py c1 > gpurun_out/r06_torch_op_sites_c1.txt 2>/dev/null
python3 benchmarks/torch_kernel_sites.py c2 > gpurun_out/r06_torch_kernel_sites_c2.txt 2>/dev/null
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
DWC_FORCE_DP=1 python3 bench.py --gpus 1 --config c3 --also "" --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_dp_dry_c3_forced_one_rank_rccl.json 2> gpurun_out/dp3.err
DWC_FORCE_DP=1 python3 bench.py --gpus 1 --config c4 --scaling strong --also "" --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_dp_dry_c4_strong_forced_one_rank_rccl.json 2> gpurun_out/dp4.err
for B in 16 8 2; do python3 benchmarks/host_vs_gpu.py $B 128; done > gpurun_out/r06_host_vs_gpu_final.txt 2>&1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_bench_default.json"))
print("c1", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
c2=d["also"]["c2"]
print("c2", c2["value"], c2["ms_per_step"], c2["decode_conv_stack"]["forward"]["frac_of_mfma_peak"], c2["decode_conv_stack"]["backward"]["frac_of_mfma_peak"])
for f in ("r06_dp_dry_c3_forced_one_rank_rccl.json","r06_dp_dry_c4_strong_forced_one_rank_rccl.json"):
    e=json.load(open("gpurun_out/"+f)); print(f, e["value"], e["ms_per_step"])
PY
head -1 gpurun_out/r06_torch_kernel_sites_c1.txt gpurun_out/r06_torch_kernel_sites_c2.txt
cat gpurun_out/r06_host_vs_gpu_final.txt | tail -8
