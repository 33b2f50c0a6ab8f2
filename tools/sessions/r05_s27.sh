#!/bin/bash
# round-5 session 27: last gate (the 64-entry lane replay switched off): every -m gpu test, smoke, the default bench line, the saturated-depth
# bench lines, 200 s of fuzz
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s27; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 400 python tools/extended_fuzz.py 200 424242 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); print(round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; blur off', round(j['value_blur_off'],1), j['value_other_depths'], 'frac', round(j['roofline']['frac'],3), 'traffic src', j['roofline']['traffic_source'][:70])"
for b in "" "--no-blur"; do timeout 900 python bench.py --depth clipped --steps 3 --warmup 1 --no-cpu-baseline --frames 64 $b 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('clipped', '$b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['diagnostics'])"; done
