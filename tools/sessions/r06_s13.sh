#!/bin/bash
# round-6 session 13: polylines_sharp -- longer per-pixel lists in the FIRST tier (KP / KS 5 / 7 -> 6 / 9, 6 / 10: 4 vector registers spilled
# at the 80-register budget), scene8 / stepped / blobs, second tier on and off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s13; mkdir -p $O
for i in 1 2; do for L in comfystereo_hip cs_ppk69 cs_ppk610; do for v in 0 49; do for k in scene8; do for b in 0 1; do
  printf "%-16s tier2 %-3s %-8s blur %s: " $L $([ $v = 0 ] && echo on || echo off) $k $b
  CS_PT_VARIANT=$v CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done; done; done 2>&1 | tee $O/ab_scene8.txt
