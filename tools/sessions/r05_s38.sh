#!/bin/bash
# round-5 session 38: closing fuzz on the final binary: 360 s over every technique, 150 s polylines-only, 100 s forward fills, 60 s per dialect
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s38; mkdir -p $O
timeout 500 python tools/extended_fuzz.py 360 959595 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 150 969696 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
CS_FUZZ_FILLS=naive_interpolating,naive,none,inverse timeout 300 python tools/extended_fuzz.py 100 979797 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
for d in f64-disparity D64 int64-sum; do CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 60 989898 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
