#!/bin/bash
# hybrid_edge: splat kernel writing the node outputs itself; parity, then cfg3 A/B against HEAD~ (libcs_base), then trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s18
timeout 900 python -m pytest tests/test_gpu_hybrid_fused.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "hybrid or cfg3 or fused" > gpurun_out/s18/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s18/tests.log
for i in 1 2 3; do
  for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
    printf "%-28s " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms')"
  done
done
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/run.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/s18/trace.txt > /dev/null; head -12 gpurun_out/s18/trace.txt
