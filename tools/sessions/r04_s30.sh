#!/bin/bash
# round-4 session 30: lean row kernel with the width / LDS base re-read per eye (scalar spills 336 -> 278, vector 77 -> 48): tie tests,
# polylines fuzz, saturated depth A/B against the build without it (libcs_rwoff)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s30; mkdir -p $O
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "ties or saturated or replay or order or sharp or 8k" > $O/tests.log 2>&1; echo "tie tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 200 python tools/extended_fuzz.py 90 1222000 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
LIBS="$C/libcs_rwoff.so $C/libcomfystereo_hip.so" tools/abn.sh --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tee $O/ab_clipped.txt
for L in $C/libcs_rwoff.so $C/libcomfystereo_hip.so; do
  rm -rf /tmp/pt; CS_LIB_PATH=$PWD/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 2 > /dev/null 2>&1
  db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db /tmp/kt.txt > /dev/null; printf "%-26s " $(basename $L); grep "k_rowwarp<3, false, true>" /tmp/kt.txt | cut -c1-140
done 2>&1 | tee $O/kernel.txt
