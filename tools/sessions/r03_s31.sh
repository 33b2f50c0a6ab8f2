#!/bin/bash
# k_gpuwarp phase cut-offs (dev build), 1080p, 128 frames, blur off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for dbg in 0 51 52 53 54; do
  rm -rf /tmp/pt
  CS_DBG=$dbg CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 128 --blur 0 --iters 3 --fill gpu_warp --h 1080 --w 1920 > /tmp/run.log 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  echo "dbg=$dbg $(grep 'k_gpuwarp<' /tmp/t.txt | awk '{print $(NF-1)}') us"
done
