#!/bin/bash
# round-4 session 33: the polylines row kernel in column ranges (wide sharp rows: lists that do not fit the LDS): range tests, 8K sharp
# tests, tie tests, polylines fuzz; 8K sharp throughput (stepped / clipped) before: 77 / 7.8 frames/s
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "ranges or ties or saturated or replay or order or sharp or 8k or wide" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 200 python tools/extended_fuzz.py 90 1525000 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
CS_DBG=30 CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 200 python tools/extended_fuzz.py 90 1626000 > $O/fuzz30.log 2>&1; echo "fuzz (ranges forced) rc=$?"; tail -2 $O/fuzz30.log
for kind in stepped clipped; do printf "sharp 7680x2160 $kind n=4: "; timeout 300 python tools/quick_bench.py --n 4 --h 2160 --w 7680 --blur 0 --iters 3 --fill polylines_sharp --kind $kind 2>&1 | tail -1 | sed 's/.*: //'; done
printf "clipped 4K soft n=64: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
