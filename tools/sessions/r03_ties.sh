#!/bin/bash
# bench lines + kernel trace of the order-dependent inputs (profiles/r03_ties/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_ties; mkdir -p $O
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${k}_blur_on.json 2>/dev/null
  timeout 1800 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames $([ $k = random8 ] && echo 8 || echo 64) > $O/bench_${k}_blur_off.json 2>/dev/null
  for b in on off; do python3 -c "
import json; j=json.load(open('$O/bench_${k}_blur_$b.json')); print('$k blur $b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"; done
done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db $O/clipped_blur_off_kernel_trace.txt > /dev/null; head -8 $O/clipped_blur_off_kernel_trace.txt | cut -c1-150
