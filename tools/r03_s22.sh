#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s22
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s22/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s22/tests.log
timeout 400 python tools/extended_fuzz.py 200 17000 > gpurun_out/s22/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/s22/fuzz.log
CS_DBG=14 python tools/quick_bench.py --n 8 --blur 0 --iters 1 --kind clipped 2>&1 | tail -4 | head -1
for n in 8 32 64; do
  printf "clipped blur=0 n=$n: "; timeout 600 python tools/quick_bench.py --n $n --blur 0 --iters 2 --kind clipped 2>&1 | tail -1 | sed 's/.*: //'
done
