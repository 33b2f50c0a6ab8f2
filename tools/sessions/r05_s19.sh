#!/bin/bash
# round-5 session 19: k_blur_fused skips window rows without edge bits (per eye); blur tests + kernel time on the metric workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s19; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "blur or lazy or node or golden or chunk" > $O/tests_blur.log 2>&1; echo "blur tests rc=$?"; tail -2 $O/tests_blur.log
for kind in stepped blobs radial; do
  rm -rf /tmp/pp
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind $kind --blur 1 --iters 4 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$kind.txt > /dev/null
  printf "%-10s " $kind; grep -E "k_blur_fused|k_gray_edges|k_blur_classify|k_polypoint" $O/trace_$kind.txt | awk '{printf "%s %s us | ", substr($0,1,22), $(NF-1)} END {print ""}'
done 2>&1 | tee $O/prepass.txt
