#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s35
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/s35/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s35/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python tools/extended_fuzz.py 300 12000 > gpurun_out/s35/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s35/fuzz.log
