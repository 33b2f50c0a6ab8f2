#!/bin/bash
# k_gpuwarp at 1080p: four 512-thread workgroups per CU with the 64-register instantiation (default now) vs three with the 6-wave one (CS_PT_VARIANT=24)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for v in 24 0; do
    printf "PT_VARIANT=%-3s " $v; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --fill gpu_warp --h 1080 --w 1920 --n 128 --blur 1 --iters 5 2>&1 | tail -1 | sed 's/.*: //'
  done
done
timeout 600 python -m pytest tests -x -q -m gpu -k "gpu_warp or gpuwarp or warp" > gpurun_out/s25_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s25_tests.log
