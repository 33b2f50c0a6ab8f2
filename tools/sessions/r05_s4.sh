#!/bin/bash
# round-5 session 4: gate after the round's kernel changes so far (naive_interpolating bit rows + wave replay, sharp at 80 VGPRs,
# anaglyph polylines through the side-by-side form, dialect instantiations of k_fwdtile / k_polypoint, hidden techniques under the
# dialect): every -m gpu test, smoke, a fuzz slice, then throughput: default bench, the two new configs, anaglyph polylines,
# the dialect paths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s4; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python tools/extended_fuzz.py 300 616161 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps; kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'node', round(r['frac_node_bytes'],3), 'pipeline', round(r['pipeline_frac'],3), 'blur off', round(j.get('value_blur_off',0),1), 'other depths', j.get('value_other_depths'))"
for c in naive_interp sharp; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$c.json; python3 -c "
import json; j=json.load(open('$O/bench_$c.json')); r=j['roofline']; print('$c', round(j['value'],1), 'fps; kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'node', round(r['frac_node_bytes'],3), 'pipeline', round(r['pipeline_frac'],3))"; done
{
for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-28s " $f; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill $f 2>&1 | tail -1 | sed 's/.*: //'
done
printf "%-28s " "polylines_soft anaglyph"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill polylines_soft --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "polylines_sharp anaglyph"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill polylines_sharp --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "soft anaglyph clipped"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill polylines_soft --mode red-cyan-anaglyph --kind clipped 2>&1 | tail -1 | sed 's/.*: //'
for d in f64-disparity D64; do for f in none naive_interpolating inverse polylines_soft polylines_sharp; do
  printf "%-28s " "$f $d"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill $f --dialect $d 2>&1 | tail -1 | sed 's/.*: //'
done; done
} 2>&1 | tee $O/table.txt
