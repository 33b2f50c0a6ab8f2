#!/bin/bash
# k_polypoint with 8-byte point records and four slots per lane (tiles of 768): A/B against HEAD, parity, fuzz
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s26
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 0 --iters 10
for i in 1 2; do printf "new, 256x3 (PT_VARIANT=3)   "; CS_PT_VARIANT=3 timeout 200 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'; done
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 1 --iters 10 --kind blobs
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 32 --blur 1 --iters 10 --h 1080 --w 1920 --div 3.5
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_blur.py tests/test_gpu_chunks.py -x -q -m gpu > gpurun_out/s26/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s26/tests.log
timeout 400 python tools/extended_fuzz.py 240 11000 > gpurun_out/s26/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s26/fuzz.log
