#!/bin/bash
# round-5 session 15: pre-pass, continued: k_blur_fused with packed FMAs on a row-pair depth tile, k_blur_classify with the batched summary
# pre-test; every GPU test; kernel times on the metric workload (64 frames); A/B of k_gray_edges strip heights / waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s15; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
C=comfystereo_amd
for L in libcomfystereo_hip.so libcs_geRB64.so libcs_geRB16.so libcs_geW3.so; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 4 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$L.txt > /dev/null
  printf "%-24s " $L; grep -E "k_blur_fused|k_gray_edges|k_blur_classify" $O/trace_$L.txt | awk '{printf "%s %s us | ", substr($0,1,22), $(NF-1)} END {print ""}'
done 2>&1 | tee $O/prepass_ab.txt
head -14 $O/trace_libcomfystereo_hip.so.txt | cut -c1-150
LIBS="$C/libcomfystereo_hip.so $C/libcs_geRB64.so" bash tools/abn.sh --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 5 2>&1 | tee $O/ab_fps.txt
