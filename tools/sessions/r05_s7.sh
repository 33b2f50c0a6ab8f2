#!/bin/bash
# round-5 session 7: which part of the k_gpuwarp pass-1 clean-up made the kernel slower?  The three parts alone and two together
# against the kernel without them (the default build): GW_P1_SPLIT (separate loops for lazy / plain depth loads), GW_P1_OFF32 (32-bit
# offsets for the depth-map stores), GW_P1_RFL (statistics words through readfirstlane)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s7; mkdir -p $O
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwSPLIT.so $C/libcs_gwOFF32.so $C/libcs_gwRFL.so $C/libcs_gwSO.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 0 --iters 10 2>&1 | tee $O/ab_1080p.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwSPLIT.so $C/libcs_gwOFF32.so $C/libcs_gwRFL.so $C/libcs_gwSO.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind blobs --div 8 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p_blur.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwSPLIT.so $C/libcs_gwOFF32.so $C/libcs_gwRFL.so $C/libcs_gwSO.so" tools/abn.sh --n 16 --fill gpu_warp --blur 0 --iters 10 2>&1 | tee $O/ab_4k.txt
