#!/bin/bash
# round-4 session 21: 7680-wide polylines_sharp rows (reduced list capacity of the row kernel behind the tile kernel, lean first
# pass + whole-row export for rows whose lists overflow): the wide-row tests, the tie tests, a polylines fuzz slice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s21; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests_fullsize.log 2>&1; echo "fullsize tests rc=$?"; tail -15 $O/tests_fullsize.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "sharp or ties or saturated or replay or order" > $O/tests_ties.log 2>&1; echo "tie tests rc=$?"; tail -3 $O/tests_ties.log
CS_FUZZ_FILLS=polylines_sharp timeout 200 python tools/extended_fuzz.py 90 818000 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
for kind in stepped clipped; do printf "sharp 7680x2160 $kind n=8: "; timeout 300 python tools/quick_bench.py --n 8 --h 2160 --w 7680 --blur 0 --iters 3 --fill polylines_sharp --kind $kind 2>&1 | tail -1 | sed 's/.*: //'; done
