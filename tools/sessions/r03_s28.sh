#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "hybrid or cfg3 or fused or lazy" > gpurun_out/s28_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s28_tests.log
for i in 1 2 3; do
  for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
    printf "%-28s " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
