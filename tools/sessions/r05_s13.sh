#!/bin/bash
# round-5 session 13: the depth-blur pre-pass: k_blur_fused's phases on the metric workload (dev build, CS_DBG 21..24 cut
# the tile function short after: mask windows / depth tile / weights / vertical box), 16 frames of 4K, blur on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s13; mkdir -p $O
C=comfystereo_amd
for d in 0 21 22 23 24; do
  rm -rf /tmp/pp
  CS_DBG=$d CS_LIB_PATH=$PWD/$C/libcomfystereo_hip_dev.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --fill polylines_soft --kind stepped --blur 1 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_dbg$d.txt > /dev/null
  printf "dbg=%s " $d; grep -E "k_blur_fused|k_gray_edges|k_blur_classify" $O/trace_dbg$d.txt | awk '{printf "%s %s us | ", substr($0,1,16), $(NF-1)} END {print ""}'
done 2>&1 | tee $O/blur_phases.txt
