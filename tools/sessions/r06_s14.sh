#!/bin/bash
# round-6 session 14: gate after the polylines work (second tier for sharp, sharp lists 6 / 9): every -m gpu test, smoke, fuzz over every
# technique + polylines only, the sharp / metric bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s14; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 300 python tools/extended_fuzz.py 150 1414 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 150 1415 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for c in sharp metric; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_$c.json')); r=j['roofline']; print('$c', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), j.get('value_other_depths'))"; done
