#!/bin/bash
# round-6 session 36: LAST gate of the round (soft point kernel under numba's sweep typing at five workgroups per CU; list bound for sharp only):
# every -m gpu test, smoke, fuzz over every technique (150 s) and under the three dialect settings (50 s each), the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s36; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 400 python tools/extended_fuzz.py 150 3601 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
for d in f64-disparity int64-sum D64; do CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 50 3605 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
CS_FUZZ_FILLS=polylines_soft,polylines_sharp CS_FUZZ_DIALECT=D64 timeout 200 python tools/extended_fuzz.py 50 3606 > $O/fuzz_poly_D64.log 2>&1; echo "fuzz poly D64 rc=$?"; tail -1 $O/fuzz_poly_D64.log
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'blur off', round(j.get('value_blur_off',0),1), 'other', j.get('value_other_depths'), 'd64', j.get('value_dialect_d64'))"
