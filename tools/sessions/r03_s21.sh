#!/bin/bash
# k_blur_fused with software-pipelined tiles: parity, A/B against HEAD, trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s21
timeout 900 python -m pytest tests/test_gpu_gray_edges.py tests/test_gpu_lazy_blur.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s21/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/s21/tests.log
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 64 --blur 1 --iters 10
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 64 --blur 1 --iters 5 > /tmp/run.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/s21/trace.txt > /dev/null; head -8 gpurun_out/s21/trace.txt
