#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "gpu_warp or gpuwarp or warp or dropin or lazy" > gpurun_out/s38_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s38_tests.log
for i in 1 2; do
for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
  printf "%-28s 4K gpu_warp blur on " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python tools/quick_bench.py --fill gpu_warp --blur 1 --iters 5 --n 32 2>&1 | tail -1 | sed 's/.*: //'
done
done
