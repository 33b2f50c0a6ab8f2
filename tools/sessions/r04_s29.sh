#!/bin/bash
# round-4 session 29: final gate after the last replay changes: every -m gpu test, smoke, the tie-path bench lines (-> profiles/r04d_ties)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s29
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s29/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s29/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
O=gpurun_out/r04d_ties; mkdir -p $O
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${k}_blur_on.json 2>/dev/null
  timeout 1800 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames 64 > $O/bench_${k}_blur_off.json 2>/dev/null
  for b in on off; do python3 -c "
import json; j=json.load(open('$O/bench_${k}_blur_$b.json')); print('$k blur $b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"; done
done 2>&1 | tee $O/summary.txt
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db $O/clipped_blur_off_kernel_trace.txt --calls k_rowwarp > /dev/null; head -8 $O/clipped_blur_off_kernel_trace.txt | cut -c1-150
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04_s29/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('gpurun_out/r04_s29/bench_default.json')); print('default', round(j['value'],1), 'fps frac', round(j['roofline']['frac'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3))"
