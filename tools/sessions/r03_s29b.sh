#!/bin/bash
# k_fwdtile: 3 against 4 source slots per lane
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for L in comfystereo_amd/libcomfystereo_hip.so comfystereo_amd/libcs_ft4.so; do
    printf "%-28s cfg5 " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
for fill in none naive naive_interpolating inverse; do
  for L in comfystereo_amd/libcomfystereo_hip.so comfystereo_amd/libcs_ft4.so; do
    printf "%-28s %-20s " "$(basename $L)" $fill; CS_LIB_PATH=$PWD/$L timeout 200 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill $fill 2>&1 | tail -1 | sed 's/.*: //'
  done
done
