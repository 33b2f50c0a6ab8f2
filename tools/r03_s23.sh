#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s23
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_lazy_blur.py tests/test_gpu_chunks.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/s23/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s23/tests.log
timeout 300 python tools/extended_fuzz.py 120 19000 > gpurun_out/s23/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/s23/fuzz.log
for rep in 1 2 3; do
for sw in 0 1; do
  printf "stepped64 no_replay_kernel=$sw: "; CS_NO_REPLAY_KERNEL=$sw timeout 600 python tools/quick_bench.py --n 64 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done; done
printf "clipped blur=0 n=64: "; timeout 600 python tools/quick_bench.py --n 64 --blur 0 --iters 2 --kind clipped 2>&1 | tail -1 | sed 's/.*: //'
