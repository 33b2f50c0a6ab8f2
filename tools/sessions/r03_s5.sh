#!/bin/bash
# round-3 session 5: phase cut-offs of k_polypoint at 3 workgroups per CU (latency-bound: time ~ workgroup lifetime)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so CS_CHUNKS=1
for occ in 13 0; do
for d in 0 31 32 33 34 35 37; do
  printf "pt_variant=%-2s dbg=%-3s " $occ $d; CS_PT_VARIANT=$occ CS_DBG=$d timeout 300 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
done
