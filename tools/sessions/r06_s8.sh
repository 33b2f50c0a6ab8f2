#!/bin/bash
# round-6 session 8: gate for k_gpuwarp_q (final form of the round's first item): every -m gpu test, smoke, 150 s of fuzz over every
# technique + 120 s gpu_warp only, bench lines of cfg 4 and the metric, kernel trace of cfg 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s8; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 300 python tools/extended_fuzz.py 150 8181 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 120 8282 > $O/fuzz_gw.log 2>&1; echo "fuzz gw rc=$?"; tail -1 $O/fuzz_gw.log
timeout 600 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_cfg4.json')); r=j['roofline']; print('cfg4', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'node', round(r['frac_node_bytes'],3))"
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3))"
rm -rf /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/kernel_trace_cfg4.txt > /dev/null; head -8 $O/kernel_trace_cfg4.txt | cut -c1-150
