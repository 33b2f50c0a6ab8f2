#!/bin/bash
# round-5 session 24: gate on the final binary (tie path: tile hints, column ranges, lane replay; pre-pass): every -m gpu test twice (fresh
# processes), smoke, 400 s of extended fuzz over every technique + 200 s polylines-only, the dialect fuzz slices
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s24; mkdir -p $O
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu_$i.log 2>&1; echo "gpu tests run $i rc=$?"; tail -2 $O/tests_gpu_$i.log; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 600 python tools/extended_fuzz.py 400 919191 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 200 929292 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for d in f64-disparity D64; do CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 100 939393 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
