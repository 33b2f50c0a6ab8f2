#!/bin/bash
# round-6 session 20: polylines_soft with lists 5 / 7 (5 / 8) PAID FOR by a shorter pass-2 pixel list (512 / 448 entries instead of one per
# tile pixel: the LDS request stays under the cliff of the seventh workgroup): stepped / scene8 / blobs / clipped, 16 and 64 frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s20; mkdir -p $O
for i in 1 2; do for L in comfystereo_hip cs_ppsk57c cs_ppsk58c; do for k in stepped scene8 blobs clipped; do for b in 0 1; do
  printf "%-16s %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 6 --fill polylines_soft --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done
printf "%-16s stepped 64 frames: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 4 --fill polylines_soft --kind stepped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
printf "%-16s radial 64 frames blur 1: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 64 --blur 1 --iters 4 --fill polylines_soft --kind radial 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab.txt
