#!/bin/bash
# Run on the GPU box: kernel trace (rocprofv3 --kernel-trace --stats) + the bench line of one BASELINE.json config
#   tools/gpu_profile_cfg.sh <tag> <cfgN>      ->  gpurun_out/<tag>_<cfgN>/{kernel_trace.txt, bench.json}
set -u
TAG=$1; CFG=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_$CFG; mkdir -p $OUT
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o p -- python3 bench.py --config $CFG --no-cpu-baseline > $OUT/trace.log 2>&1
db=$(find /tmp/prof_c -name '*.db' | head -1)
python3 tools/prof_summary.py $db $OUT/kernel_trace.txt > /dev/null
grep '^{' $OUT/trace.log > $OUT/bench_under_profiler.json; rm -f $OUT/trace.log; rm -rf /tmp/prof_c
timeout 300 python3 bench.py --config $CFG --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench.json
head -8 $OUT/kernel_trace.txt | cut -c1-150
