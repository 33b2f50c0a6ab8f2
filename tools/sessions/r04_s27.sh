#!/bin/bash
# round-4 session 27: 800 s of extended fuzz (every technique) on the final kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s27
timeout 1000 python tools/extended_fuzz.py 800 1121000 > gpurun_out/r04_s27/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r04_s27/fuzz.log
