#!/bin/bash
# round-4 session 22: why are 7680-wide sharp frames slow?  row statistics (CS_DBG=14) and kernel trace; soft for comparison
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s22; mkdir -p $O
for f in polylines_sharp polylines_soft; do
  CS_DBG=14 timeout 300 python tools/quick_bench.py --n 4 --h 2160 --w 7680 --blur 0 --iters 2 --fill $f --kind stepped 2>&1 | grep -v amdgpu.ids | tail -4
  rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 4 --h 2160 --w 7680 --blur 0 --iters 2 --fill $f --kind stepped > /dev/null 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db $O/${f}_trace.txt > /dev/null; head -8 $O/${f}_trace.txt | cut -c1-150
done
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "8k_sharp" 2>&1 | tail -5
