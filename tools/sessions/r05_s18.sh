#!/bin/bash
# round-5 session 18: gate after the pre-pass work (k_blur_fused<FAST>, k_blur_classify, nontemporal gray stores): every GPU test, the fuzz,
# the bench line, kernel trace of the bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s18; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 400 python tools/extended_fuzz.py 240 515151 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cat $O/bench_default.json | cut -c1-900
rm -rf /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 bench.py --steps 6 --warmup 2 --no-other-depths > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/kernel_trace_bench.txt > /dev/null; head -16 $O/kernel_trace_bench.txt | cut -c1-150
