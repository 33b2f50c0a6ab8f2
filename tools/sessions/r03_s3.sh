#!/bin/bash
# round-3 session 3: general search with hoisted segment data -- parity, then A/B against the previous kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_blur.py -x -q -m gpu > gpurun_out/s3/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s3/tests.log
timeout 400 python tools/extended_fuzz.py 200 7000 > gpurun_out/s3/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s3/fuzz.log
export CS_CHUNKS=1
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 0 --iters 10
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 1 --iters 10 --kind blobs
