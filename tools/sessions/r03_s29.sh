#!/bin/bash
# phase cut-offs of the general polylines row kernel (dev build; every row through it: CS_NO_TILE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so CS_NO_TILE=1
for d in 1 2 3 4 5 0; do
  printf "dbg=$d: "; CS_DBG=$d timeout 300 python tools/quick_bench.py --n 8 --blur 0 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done
