#!/bin/bash
# k_polypoint geometries with the 8-byte records: 256 x 4 (default) against 256 x 5 (PT_VARIANT=7) and 256 x 3 (=3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for v in 0 7 3; do
    printf "4K blur0 PT_VARIANT=%-2s " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
  done
done
for v in 0 7; do printf "4K blobs blur1 PT_VARIANT=%-2s " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 32 --blur 1 --iters 10 --kind blobs 2>&1 | tail -1 | sed 's/.*: //'; done
for v in 0 7; do printf "1080p PT_VARIANT=%-2s " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 32 --blur 1 --iters 10 --h 1080 --w 1920 --div 3.5 2>&1 | tail -1 | sed 's/.*: //'; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s27_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s27_tests.log
