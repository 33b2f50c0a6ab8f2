#!/bin/bash
# k_gray_edges knobs: rows per strip (32 default, 64, 16), 3 waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in libcomfystereo_hip libcs_rb64 libcs_rb16 libcs_w3; do
  rm -rf /tmp/pt
  CS_LIB_PATH=$PWD/comfystereo_amd/$L.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 64 --blur 1 --iters 4 --fill none > /tmp/run.log 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  echo "$L $(grep k_gray_edges /tmp/t.txt | awk '{print $(NF-1)}') us"
done
