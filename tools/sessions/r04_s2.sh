#!/bin/bash
# round-4 session 2: k_polypoint experiments, A/B in one session (32 x 4K, blur off): +100 packed adds / +100 dependent adds per
# wave (the model, continued); the two eyes of a row group adjacent in dispatch order (PP_EYE_GROUP 0 / 2 / 4: L2 reuse of the image
# row, with the FETCH_SIZE / WRITE_SIZE counters); the one-before point peeled off the dense pass (PP_PEEL) + its parity
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s2
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_padP100.so $C/libcs_padD100.so $C/libcs_eg0.so $C/libcs_eg2.so $C/libcs_eg4.so $C/libcs_peel.so" \
  tools/abn.sh --n 32 --blur 0 --iters 20 2>&1 | tee gpurun_out/r04_s2/ab.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_eg0.so $C/libcs_eg2.so $C/libcs_eg4.so" PMC=k_polypoint bash -c 'LIBS2="$LIBS"; cd "$GRAFT_REPO_ROOT"; for L in $LIBS2; do for grp in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pp; CS_LIB_PATH=$PWD/$L timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --blur 0 --iters 5 > /tmp/run.log 2>&1; db=$(find /tmp/pp -name "*.db" | head -1); printf "%-28s %s " "$(basename $L)" $grp; [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "k_polypoint" | awk "{print \$(NF-4), \$(NF-2), \$(NF-1), \$NF}"; done; done' 2>&1 | tee gpurun_out/r04_s2/pmc.txt
CS_LIB_PATH=$PWD/$C/libcs_peel.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "poly or soft or metric or cfg2 or ties or fuzz" > gpurun_out/r04_s2/peel_tests.log 2>&1; echo "peel tests rc=$?"; tail -3 gpurun_out/r04_s2/peel_tests.log
CS_FUZZ_FILLS=polylines_soft CS_LIB_PATH=$PWD/$C/libcs_peel.so timeout 300 python tools/extended_fuzz.py 150 424242 > gpurun_out/r04_s2/peel_fuzz.log 2>&1; echo "peel fuzz rc=$?"; tail -2 gpurun_out/r04_s2/peel_fuzz.log
