#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s13
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py tests/test_gpu_chunks.py tests/test_gpu_sharded.py -x -q -m gpu -k "hybrid or node or asd or apply or cfg3 or chunk or host or two_ranks" > gpurun_out/s13/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s13/tests.log
timeout 300 python tools/extended_fuzz.py 120 9000 > gpurun_out/s13/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s13/fuzz.log
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so" tools/abn.sh --n 16 --blur 1 --iters 10 --fill hybrid_edge
