#!/bin/bash
# round-4 session 32: 500 s of polylines-only extended fuzz on the final replay / row kernels (node cases now include 8-bit noise depth:
# whole-row stretches), then 300 s over every technique
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s32
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 700 python tools/extended_fuzz.py 500 1323000 > gpurun_out/r04_s32/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s32/fuzz_poly.log
timeout 500 python tools/extended_fuzz.py 300 1424000 > gpurun_out/r04_s32/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s32/fuzz_all.log
