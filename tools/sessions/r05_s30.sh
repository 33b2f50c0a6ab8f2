#!/bin/bash
# round-5 session 30: how many list slots fit seven workgroups per CU?  200 (default) against 128 / 184 / 216 on the headline workload, then
# saturated depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s30; mkdir -p $O
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_dcap128.so $C/libcs_dcap184.so $C/libcs_dcap216.so" bash tools/abn.sh --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 5 2>&1 | tee $O/ab_headline.txt
for b in 1 0; do for L in libcomfystereo_hip.so libcs_dcap128.so libcs_dcap184.so libcs_dcap216.so; do
  printf "%-22s clipped blur $b: " $L; CS_LIB_PATH=$PWD/$C/$L timeout 600 python tools/quick_bench.py --n 32 --fill polylines_soft --kind clipped --blur $b --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done; done 2>&1 | tee $O/ab_clipped.txt
