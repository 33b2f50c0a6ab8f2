#!/bin/bash
# both-eyes workgroups of k_polypoint: A/B against HEAD first (cheap), parity second
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s16
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 0 --iters 10
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 1 --iters 10 --kind blobs
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_blur.py -x -q -m gpu > gpurun_out/s16/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s16/tests.log
timeout 400 python tools/extended_fuzz.py 240 11000 > gpurun_out/s16/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s16/fuzz.log
