#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in clipped random8; do
  timeout 600 python tools/quick_bench.py --n 8 --blur 0 --iters 2 --kind $k 2>&1 | tail -3
done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 8 --blur 0 --iters 2 --kind clipped > /dev/null 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db gpurun_out/clipped_trace.txt > /dev/null; head -8 gpurun_out/clipped_trace.txt | cut -c1-150
