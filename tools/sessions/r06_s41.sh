#!/bin/bash
# round-6 session 41: every -m gpu test and smoke on the round's final binary (after the lazy blur tiles under a dialect flag: host code only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s41; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
