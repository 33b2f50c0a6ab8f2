#!/bin/bash
# round-5 session 8: gate after the k_gpuwarp bit rows / keyed table, the replay pool give-back and the new targeted tests:
# every -m gpu test, smoke, the tie-path bench lines (clipped / random8, blur off / on: does the pool give-back change noise depth?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s8; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${k}_blur_on.json 2>/dev/null
  timeout 1800 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames 64 > $O/bench_${k}_blur_off.json 2>/dev/null
  for b in on off; do python3 -c "
import json; j=json.load(open('$O/bench_${k}_blur_$b.json')); print('$k blur $b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"; done
done 2>&1 | tee $O/ties.txt
timeout 600 python bench.py --depth random8 --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames 16 > $O/bench_random8_16f.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_random8_16f.json')); print('random8 blur off 16 frames', round(j['value'],1), 'fps', j['diagnostics'])" | tee -a $O/ties.txt
