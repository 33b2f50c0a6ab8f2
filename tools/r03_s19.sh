#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for sw in 0 1; do
  printf "stepped64 no_replay_kernel=$sw: "; CS_NO_REPLAY_KERNEL=$sw timeout 600 python tools/quick_bench.py --n 64 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done; done
mkdir -p gpurun_out/r03_ties
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_ties/bench_$k.json 2>/dev/null; python3 -c "
import json; j=json.load(open('gpurun_out/r03_ties/bench_$k.json')); print('$k', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['diagnostics'])"
done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db gpurun_out/r03_ties/clipped_kernel_trace.txt > /dev/null; head -9 gpurun_out/r03_ties/clipped_kernel_trace.txt | cut -c1-150
