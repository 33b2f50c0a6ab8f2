#!/bin/bash
# round-6 session 12: polylines_sharp on scene8 depth -- more dirty slots in the FIRST tier (PP_DCAP_SHARP 160 / 240 / 320: the sharp
# instantiation runs six workgroups per CU), with the second tier on (CS_PT_VARIANT=0) and off (49); stepped depth beside it (no regression?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s12; mkdir -p $O
for i in 1 2; do for L in comfystereo_hip cs_ppds240 cs_ppds320; do for v in 0 49; do for k in scene8 stepped; do
  printf "%-16s tier2 %-3s %-8s: " $L $([ $v = 0 ] && echo on || echo off) $k
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 4 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: //; s/errors.*//; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done; done 2>&1 | tee $O/ab.txt
for v in 0 49; do for k in scene8 stepped; do
  printf "soft tier2 %-3s %-8s: " $([ $v = 0 ] && echo on || echo off) $k
  CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 4 --fill polylines_soft --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: //; s/errors.*//; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done 2>&1 | tee -a $O/ab.txt
