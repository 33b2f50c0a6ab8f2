#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s20
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/s20/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/s20/tests.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])"
