#!/bin/bash
# round-3 session 2: phase cut-offs of k_polypoint (dev build): where does the time go?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so CS_CHUNKS=1
for rep in 1 2; do
for d in 0 31 32 33 34 35 36 37; do
  printf "dbg=%-3s " $d; CS_DBG=$d timeout 300 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
done
