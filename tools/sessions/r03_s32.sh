#!/bin/bash
# k_gpuwarp: DPP prefix maximum instead of the gap walk -- parity, A/B at two divergences and at 4K
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "gpu_warp or gpuwarp or warp or dropin or sharded" > gpurun_out/s32_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s32_tests.log
for args in "--h 1080 --w 1920 --n 128 --div 8" "--h 1080 --w 1920 --n 128 --div 3.5" "--n 16 --div 8"; do
  for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
    printf "%-26s %-36s " "$(basename $L)" "$args"; CS_LIB_PATH=$PWD/$L timeout 300 python tools/quick_bench.py --fill gpu_warp --blur 1 --iters 5 $args 2>&1 | tail -1 | sed 's/.*: //'
  done
done
for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
  printf "%-28s cfg4 " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python bench.py --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done
