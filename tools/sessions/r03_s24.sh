#!/bin/bash
# (1) the D32 tie path after the dialect moved into kernels of its own: random8 / clipped against the build before D64
# (2) hybrid splat with parked pixels: parity + cfg3 A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s24
timeout 900 python -m pytest tests/test_gpu_dialect.py tests/test_gpu_hybrid_fused.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s24/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/s24/tests.log
for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
  for kind in random8 clipped; do
    printf "%-28s %-8s " "$(basename $L)" $kind; CS_LIB_PATH=$PWD/$L timeout 300 python tools/quick_bench.py --n 4 --blur 0 --iters 2 --kind $kind 2>&1 | tail -1 | sed 's/.*: //'
  done
done
for i in 1 2 3; do
  for L in comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so; do
    printf "%-28s " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 300 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
