#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s37
timeout 1000 python tools/extended_fuzz.py 800 70000 > gpurun_out/s37/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s37/fuzz.log
