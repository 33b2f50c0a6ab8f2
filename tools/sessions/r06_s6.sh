#!/bin/bash
# round-6 session 6: is the scalar data cache stale across kernels?  (k_gpuwarp_q reads the constants k_gpuwarp_flags wrote with s_load)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s6; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "warp or cfg4 or lazy or 8k or dropin or chunks" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=gpu_warp timeout 200 python tools/extended_fuzz.py 60 6162 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
