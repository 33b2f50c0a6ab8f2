#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
for m in "node keep" "node release" "host_pageable keep" "host_pageable release"; do python tools/host_probe_r03.py $m 2>&1 | grep "keep \[\|release \["; done
timeout 600 python tools/node_host_bench.py --n 32 --iters 3 2>&1 | tail -3
