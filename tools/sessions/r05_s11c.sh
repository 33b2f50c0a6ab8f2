#!/bin/bash
# round-5 session 11c: is the sharp fixture failure of session 9 (first GPU process of a fresh box) reproducible?  40 fresh processes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s11; mkdir -p $O
for i in $(seq 1 40); do timeout 300 python -m pytest tests/test_gpu_dialect.py -x -q -m gpu -k "fixture or d64_matches" 2>&1 | tail -1; done | sort | uniq -c | tee $O/fresh_process_runs.txt
