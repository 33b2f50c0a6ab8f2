#!/bin/bash
# k_blur_fused: tile statistics folded by a kernel of their own: parity, kernel times, A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_gray_edges.py tests/test_gpu_lazy_blur.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/s33_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s33_tests.log
for L in libcs_base libcomfystereo_hip; do
  rm -rf /tmp/pt
  CS_LIB_PATH=$PWD/comfystereo_amd/$L.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 64 --blur 1 --iters 4 --fill none > /tmp/run.log 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  echo "$L $(grep 'k_blur_fused\|k_blur_tile_stats' /tmp/t.txt | awk '{print $1, $(NF-1)}' | tr '\n' ' ')"
done
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 64 --blur 1 --iters 10
