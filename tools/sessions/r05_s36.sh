#!/bin/bash
# round-5 session 36: the round's last gate: every -m gpu test, smoke, 200 s of fuzz, the default bench line, the bench lines of cfg 5 and
# naive_interp (k_fwdtile got a loop around its body for the second tier: no cost on the forward fills?), the forward fills by depth kind
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s36b; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 400 python tools/extended_fuzz.py 200 848484 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print(round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; blur off', round(j['value_blur_off'],1), j['value_other_depths'], 'frac', round(r['frac'],3), round(r['frac_node_bytes'],3), round(r['pipeline_frac'],3), 'kernel_ms', round(r['kernel_ms'],3))"
for c in cfg5 naive_interp; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_$c.json')); r=j['roofline']; print('$c', round(j['value'],1), 'fps frac', round(r['frac'],3), 'kernel_ms', round(r['kernel_ms'],3))"; done
for f in none naive naive_interpolating inverse; do for k in stepped clipped; do
  printf "%-22s %-8s blur 0: " $f $k; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 6 --fill $f --kind $k 2>&1 | tail -1 | sed 's/.*: //'
done; done 2>&1 | tee $O/fwd_table.txt
