#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s35
timeout 700 python tools/extended_fuzz.py 500 32000 > gpurun_out/s35/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s35/fuzz.log
timeout 400 python tools/extended_fuzz.py 240 51000 > gpurun_out/s35/fuzz2.log 2>&1; echo "fuzz2 rc=$?"; tail -2 gpurun_out/s35/fuzz2.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/s35/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s35/tests.log
