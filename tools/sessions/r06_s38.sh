#!/bin/bash
# round-6 session 38: closing fuzz on the FINAL binary with fresh seeds: 330 s over every technique, 120 s polylines-only, 80 s forward + post fills,
# 60 s gpu_warp, 60 s per dialect setting (every technique) + 60 s polylines under D64
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s38; mkdir -p $O
timeout 500 python tools/extended_fuzz.py 330 3801 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 120 3802 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
CS_FUZZ_FILLS=none,naive,naive_interpolating,inverse,none_post,inverse_post timeout 300 python tools/extended_fuzz.py 80 3803 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 60 3804 > $O/fuzz_gw.log 2>&1; echo "fuzz gw rc=$?"; tail -1 $O/fuzz_gw.log
for d in f64-disparity int64-sum D64; do CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 60 3805 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
CS_FUZZ_FILLS=polylines_soft,polylines_sharp CS_FUZZ_DIALECT=D64 timeout 200 python tools/extended_fuzz.py 60 3806 > $O/fuzz_poly_D64.log 2>&1; echo "fuzz poly D64 rc=$?"; tail -1 $O/fuzz_poly_D64.log
