#!/bin/bash
# round-4 session 18: k_gpuwarp's scalar-register spills (80 SGPRs at 8 waves per SIMD: ~290 v_writelane / v_readlane in the kernel):
# instantiations for the node layout (no generic strides: libcs_gwnl), + width / LDS base re-read per eye (default build),
# + tile selectors computed per eye (libcs_gwsel), against the build before (libcs_gwhead): warp tests, A/B at 1080p and 4K
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s18; mkdir -p $O
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_dropin.py -x -q -m gpu -k "warp or cfg4" > $O/tests.log 2>&1; echo "warp tests rc=$?"; tail -3 $O/tests.log
CS_LIB_PATH=$PWD/$C/libcs_gwsel.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "warp or cfg4" > $O/tests_sel.log 2>&1; echo "gwsel warp tests rc=$?"; tail -3 $O/tests_sel.log
LIBS="$C/libcs_gwhead.so $C/libcs_gwnl.so $C/libcomfystereo_hip.so $C/libcs_gwsel.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p.txt
LIBS="$C/libcs_gwhead.so $C/libcs_gwnl.so $C/libcomfystereo_hip.so $C/libcs_gwsel.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee $O/ab_4k.txt
for L in $C/libcs_gwhead.so $C/libcomfystereo_hip.so $C/libcs_gwsel.so; do
  rm -rf /tmp/pt; CS_LIB_PATH=$PWD/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 5 > /dev/null 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db /tmp/kt.txt > /dev/null; printf "%-26s " $(basename $L); grep k_gpuwarp /tmp/kt.txt | cut -c1-140
done 2>&1 | tee $O/kernel_1080p.txt
