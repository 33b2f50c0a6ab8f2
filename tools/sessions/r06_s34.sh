#!/bin/bash
# round-6 session 34: after the closing profiles -- the ranged-row list bound restricted to polylines_sharp (the soft lean kernel back at the
# first session's 44 spilled registers): polylines / tie / width tests, polylines fuzz, soft + sharp on saturated depth, and the default
# bench line with its new D64 leg (value_dialect_d64)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s34; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or tie or replay or order or sharp or scene8 or 8k or 8192 or wide or stress" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_sharp,polylines_soft timeout 300 python tools/extended_fuzz.py 100 3401 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for f in polylines_soft polylines_sharp; do
  printf "%-16s clipped blur 0: " $f; timeout 300 python tools/quick_bench.py --n 16 --iters 5 --fill $f --kind clipped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done 2>&1 | tee $O/clipped.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'blur off', round(j.get('value_blur_off',0),1), 'other', j.get('value_other_depths'), 'd64', j.get('value_dialect_d64'))"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
