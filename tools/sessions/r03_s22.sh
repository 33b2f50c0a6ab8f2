#!/bin/bash
# k_blur_fused phase cut-offs (dev build): where do the 19 us per tile go?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s22
for dbg in 0 21 22 23 24; do
  rm -rf /tmp/pt
  CS_DBG=$dbg CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 32 --blur 1 --iters 3 --fill none > /tmp/run.log 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  echo "dbg=$dbg $(grep k_blur_fused /tmp/t.txt | awk '{print $(NF-1)}') us"
done | tee gpurun_out/s22/phases.txt
