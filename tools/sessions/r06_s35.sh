#!/bin/bash
# round-6 session 35: the soft point kernel under numba's sweep typing at 5 / 6 workgroups per CU (95 registers fit 5 without a spill) against 4:
# D64 polylines_soft, 16 x 4K, stepped / scene8 depth, blur off / on, two alternations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s35; mkdir -p $O
for i in 1 2; do for L in comfystereo_hip cs_sw5 cs_sw6; do for k in stepped scene8; do for b in 0 1; do
  printf "%-16s soft %-8s blur %s D64: " $L $k $b; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --fill polylines_soft --kind $k --blur $b --dialect D64 --iters 6 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done; done; done 2>&1 | tee $O/ab.txt
