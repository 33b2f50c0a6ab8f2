#!/bin/bash
# round-4 session 3: eye-group order as the default of the three tile kernels (parity: the whole GPU suite), group sizes 8 << 3..7
# rows A/B, cfg3 / cfg2 bench lines, the host pipeline after its rewrite (result allocation off the critical path, prewarm,
# streaming stores, gpu_warp's mask as bytes and depth maps as one channel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s3/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04_s3/tests.log
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_eg3.so $C/libcs_eg5.so $C/libcs_eg6.so $C/libcs_eg7.so" tools/abn.sh --n 32 --blur 0 --iters 20 2>&1 | tee gpurun_out/r04_s3/ab.txt
for c in cfg3 cfg2 metric; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_s3/bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r04_s3/bench_$c.json')); print('$c', round(d['value'],1), 'fps; kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"; done
timeout 600 python tools/node_host_bench.py --n 32 --iters 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s3/host_4k.txt
timeout 600 python tools/node_host_bench.py --n 32 --iters 3 --prewarm 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s3/host_4k_cold.txt
timeout 600 python tools/node_host_bench.py --n 24 --iters 3 --fill "GPU Warp (Fast)" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s3/host_4k_gpuwarp.txt
