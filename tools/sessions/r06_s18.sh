#!/bin/bash
# round-6 session 18: polylines_sharp first tier at FIVE workgroups per CU (-DPP_SHARP_MINW=5: 88 registers, none spilled) against six
# (80 registers, 4 spilled to scratch: counted traffic 1.24 x the algorithmic bytes) with the lists 6 / 9: stepped / scene8 / blobs, 16 and 64 frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s18; mkdir -p $O
for i in 1 2 3; do for L in comfystereo_hip cs_ppsm5; do for k in stepped scene8 blobs; do for b in 0 1; do
  printf "%-16s %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 6 --fill polylines_sharp --kind $k 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done
printf "%-16s stepped 64 frames: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 4 --fill polylines_sharp --kind stepped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab.txt
