#!/bin/bash
# round-4 session 36: compiler scheduling strategies on the two heaviest kernels (a lottery ticket): k_polypoint with
# -amdgpu-sched-strategy=max-ilp / max-memory-clause, -O2, -amdgpu-schedule-metric-bias=100; k_gpuwarp with max-ilp / max-memory-clause
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s36; mkdir -p $O
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_ppilp.so $C/libcs_ppmem.so $C/libcs_ppo2.so $C/libcs_ppbias.so" tools/abn.sh --n 32 --blur 0 --iters 10 --fill polylines_soft 2>&1 | tee $O/ab_polypoint.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwilp.so $C/libcs_gwmem.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee $O/ab_gpuwarp.txt
