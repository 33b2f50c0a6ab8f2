#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s18
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_blur.py tests/test_gpu_dropin.py tests/test_gpu_dialect.py -x -q -m gpu > gpurun_out/s18/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s18/tests.log
timeout 500 python tools/extended_fuzz.py 300 13000 > gpurun_out/s18/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s18/fuzz.log
for k in clipped random8 stepped; do
  for sw in 0 1; do
  printf "$k no_replay_kernel=$sw: "; CS_NO_REPLAY_KERNEL=$sw timeout 600 python tools/quick_bench.py --n 8 --blur 0 --iters 2 --kind $k 2>&1 | tail -1 | sed 's/.*: //'
  done
done
printf "clipped n=32: "; timeout 600 python tools/quick_bench.py --n 32 --blur 1 --iters 2 --kind clipped 2>&1 | tail -1 | sed 's/.*: //'
