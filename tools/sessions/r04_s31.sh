#!/bin/bash
# round-4 session 31: geometry of the lean row kernel: 1024 threads x 8 waves per SIMD (64 VGPRs, 48 spilled) against 768 x 6
# (80 VGPRs, 22 spilled) and 512 x 4 (116 VGPRs, none spilled); two workgroups per CU in every case.  Saturated depth, tie tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s31; mkdir -p $O
C=comfystereo_amd
for L in $C/libcs_lean768.so $C/libcs_lean512.so; do
  CS_LIB_PATH=$PWD/$L timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "ties or saturated or replay or order or 8k" 2>&1 | tail -1
done
LIBS="$C/libcomfystereo_hip.so $C/libcs_lean768.so $C/libcs_lean512.so" tools/abn.sh --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tee $O/ab_clipped.txt
for L in $C/libcomfystereo_hip.so $C/libcs_lean768.so $C/libcs_lean512.so; do
  rm -rf /tmp/pt; CS_LIB_PATH=$PWD/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 2 > /dev/null 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db /tmp/kt.txt > /dev/null; printf "%-26s " $(basename $L); grep "k_rowwarp<3, false, true>" /tmp/kt.txt | cut -c1-140
done 2>&1 | tee $O/kernel.txt
