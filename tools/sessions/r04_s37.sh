#!/bin/bash
# round-4 session 37: closing fuzz on the final binary: 300 s polylines-only with the column ranges forced (CS_DBG=30), 300 s polylines-only
# default, 300 s over every technique
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s37
CS_DBG=30 CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 500 python tools/extended_fuzz.py 300 1727000 > gpurun_out/r04_s37/fuzz_ranges.log 2>&1; echo "fuzz (ranges forced) rc=$?"; tail -1 gpurun_out/r04_s37/fuzz_ranges.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 500 python tools/extended_fuzz.py 300 1828000 > gpurun_out/r04_s37/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04_s37/fuzz_poly.log
timeout 500 python tools/extended_fuzz.py 300 1929000 > gpurun_out/r04_s37/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04_s37/fuzz_all.log
