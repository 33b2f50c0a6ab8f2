#!/bin/bash
# README table: 4K side by side, depth blur off, device tensors, frames/s per technique
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-20s " $f; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill $f 2>&1 | tail -1 | sed 's/.*: //'
done
printf "%-20s " "polylines_soft anaglyph"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill polylines_soft --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
printf "%-20s " "gpu_warp 1080p"; timeout 300 python tools/quick_bench.py --n 32 --h 1080 --w 1920 --blur 0 --iters 5 --fill gpu_warp 2>&1 | tail -1 | sed 's/.*: //'
