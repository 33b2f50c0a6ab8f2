#!/bin/bash
# round-5 session 37: after the barrier in front of bench.py's JSON line (N > 1): the sharded tests three times, then every -m gpu test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s37; mkdir -p $O
for i in 1 2 3; do timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -1; done | tee $O/sharded.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests_gpu.log
