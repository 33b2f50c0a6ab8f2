#!/bin/bash
# round-4 session 19: k_gpuwarp, continued: node-layout instantiation + per-eye re-read + per-eye tile selectors (default build) against
# the round's first build (libcs_gwhead) and against re-reads at every phase (libcs_gwph); 256-thread workgroups at 1080p (CS_PT_VARIANT=26)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s19; mkdir -p $O
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_dropin.py -x -q -m gpu -k "warp or cfg4 or gpu_warp or Warp" > $O/tests.log 2>&1; echo "warp tests rc=$?"; tail -3 $O/tests.log
LIBS="$C/libcs_gwhead.so $C/libcomfystereo_hip.so $C/libcs_gwph.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p.txt
for i in 1 2; do printf "256 threads: "; CS_PT_VARIANT=26 timeout 200 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'; done 2>&1 | tee -a $O/ab_1080p.txt
LIBS="$C/libcs_gwhead.so $C/libcomfystereo_hip.so $C/libcs_gwph.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee $O/ab_4k.txt
