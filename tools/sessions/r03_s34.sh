#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for dbg in 0 25; do
  rm -rf /tmp/pt
  CS_DBG=$dbg CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 64 --blur 1 --iters 3 --fill none > /tmp/run.log 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  echo "dbg=$dbg k_blur_fused $(grep k_blur_fused /tmp/t.txt | awk '{print $(NF-1)}') us"
done
