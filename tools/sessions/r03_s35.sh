#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s35
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/s35/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s35/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python tools/extended_fuzz.py 200 180000 > gpurun_out/s35/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s35/fuzz.log
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline', round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
