#!/bin/bash
# Long runs in one process each: finite losses at the end (bench.py reads them back), no poisoned tile / expired hand-off (the
# status words are polled every step and raise), throughput steady.  gpurun --timeout 1500 -- 'bash benchmarks/soak.sh'
for spec in "c1 1200" "c2 300" "c3 60" "c4 60"; do
  set -- $spec
  python3 bench.py --config $1 --also "" --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 soak %d steps: %.1f images/s, %.2f ms/step, per-GPU batch %s, loss_dis_all %s loss_gen_total %s' % (d['steps'], d['value'], d['ms_per_step'], d['config'].get('per_gpu_batch'), d.get('loss_dis_all'), d.get('loss_gen_total')))"
done
