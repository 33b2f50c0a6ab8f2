#!/bin/bash
# round-3 session 4: occupancy what-if of k_polypoint (LDS padded so that 3..6 workgroups fit a CU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CS_CHUNKS=1
for rep in 1 2; do
for v in 0 16 15 14 13; do
  printf "pt_variant=%-3s " $v; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
done
