#!/bin/bash
# the round-end checks in one call: every -m gpu test, smoke, the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/full/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/full/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/full/bench_default.json 2> gpurun_out/full/bench_default.err; tail -c 2500 gpurun_out/full/bench_default.json
