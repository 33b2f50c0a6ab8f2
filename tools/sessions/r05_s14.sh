#!/bin/bash
# round-5 session 14 (run twice: before and after the software pipeline over tiles): k_blur_fused<FAST>: every GPU
# test, then the phases again (dev build) and the release kernel's time on the metric workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s14b; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
C=comfystereo_amd
for d in 0 21 22 23 24; do
  rm -rf /tmp/pp
  CS_DBG=$d CS_LIB_PATH=$PWD/$C/libcomfystereo_hip_dev.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --fill polylines_soft --kind stepped --blur 1 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_dbg$d.txt > /dev/null
  printf "dbg=%s " $d; grep -E "k_blur_fused|k_gray_edges|k_blur_classify" $O/trace_dbg$d.txt | awk '{printf "%s %s us | ", substr($0,1,16), $(NF-1)} END {print ""}'
done 2>&1 | tee $O/blur_phases.txt
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 3 > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_release_64.txt > /dev/null; head -12 $O/trace_release_64.txt | cut -c1-150
tail -1 /tmp/run.log
