#!/bin/bash
# round-5 session 33: gate after the 160 list slots of k_polypoint: every -m gpu test, smoke, 300 s of fuzz over every technique + 150 s
# polylines-only, the default bench line, the tie-path bench lines (saturated / noise depth, blur on / off, 64 frames), kernel trace of the
# bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s33; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 500 python tools/extended_fuzz.py 300 626262 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 150 636363 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print(round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; blur off', round(j['value_blur_off'],1), j['value_other_depths'], 'frac', round(r['frac'],3), round(r['frac_node_bytes'],3), round(r['pipeline_frac'],3), 'kernel_ms', round(r['kernel_ms'],3))"
mkdir -p $O/ties
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > $O/ties/bench_${k}_blur_on.json 2>/dev/null
  timeout 1800 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames 64 > $O/ties/bench_${k}_blur_off.json 2>/dev/null
  for b in on off; do python3 -c "
import json; j=json.load(open('$O/ties/bench_${k}_blur_$b.json')); print('$k blur $b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"; done
done 2>&1 | tee $O/ties/summary.txt
for f in polylines_sharp; do for k in clipped; do for b in 0 1; do printf "%s %s blur %s (32 frames): " $f $k $b; timeout 600 python tools/quick_bench.py --n 32 --fill $f --kind $k --blur $b --iters 3 2>&1 | tail -1 | sed 's/.*: //'; done; done; done | tee $O/ties/sharp.txt
rm -rf /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 bench.py --steps 6 --warmup 2 --no-other-depths > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/kernel_trace_bench.txt > /dev/null; head -8 $O/kernel_trace_bench.txt | cut -c1-150
