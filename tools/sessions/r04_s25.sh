#!/bin/bash
# round-4 session 25: stretch replay -- the closeness scan over long lists by a DPP wave maximum instead of a scalar pass per
# candidate (the replay kernel was scalar-unit-bound on noise depth): tie tests, polylines fuzz, noise / saturated depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "ties or saturated or replay or order or sharp or 8k" > $O/tests.log 2>&1; echo "tie tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 120 1020000 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
for k in random8 clipped; do
  timeout 900 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames $([ $k = random8 ] && echo 8 || echo 64) > $O/bench_${k}_blur_off.json 2>/dev/null
  python3 -c "
import json; j=json.load(open('$O/bench_${k}_blur_off.json')); print('$k blur off', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"
done
timeout 900 python bench.py --depth random8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_random8_blur_on.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_random8_blur_on.json')); print('random8 blur on', round(j['value'],1), 'fps')"
