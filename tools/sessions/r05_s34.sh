#!/bin/bash
# round-5 session 34: every technique on saturated and on blobs depth (16 x 4K SBS, blur off and on): is any tile kernel handing most of
# its rows to a row kernel there, as k_polypoint did before the 160 list slots?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s34; mkdir -p $O
for k in clipped blobs stepped; do for b in 0 1; do for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-8s blur %s %-22s " $k $b $f; timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill $f --kind $k 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee $O/table_by_depth.txt
