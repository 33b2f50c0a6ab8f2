#!/bin/bash
# round-5 session 17: k_blur_classify (candidate mask + one round trip per candidate block), nontemporal gray stores (k_gray_edges default,
# k_gray A/B); blur tests; blur on / off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s17; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "blur or lazy or node or golden or chunk" > $O/tests_blur.log 2>&1; echo "blur tests rc=$?"; tail -2 $O/tests_blur.log
C=comfystereo_amd
for rep in 1 2; do
for L in libcomfystereo_hip.so libcs_gePLAIN.so; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 4 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$L.txt > /dev/null
  printf "%-24s " $L; grep -E "k_blur_fused|k_gray_edges|k_blur_classify" $O/trace_$L.txt | awk '{printf "%s %s us | ", substr($0,1,22), $(NF-1)} END {print ""}'
done
for L in libcomfystereo_hip.so libcs_grayNT.so; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 0 --iters 4 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace0_$L.txt > /dev/null
  printf "%-24s " $L; grep -E "k_gray|k_polypoint" $O/trace0_$L.txt | awk '{printf "%s %s us | ", substr($0,1,22), $(NF-1)} END {print ""}'
done
done 2>&1 | tee $O/prepass_ab.txt
