#!/bin/bash
# round-4 session 5: (1) k_gpuwarp with XCD-contiguous rows (GW_XCD_ROWS: the neighbour row of the 4-corner blend from L2) A/B at
# 1080p and 4K, with FETCH_SIZE; (2) forward_warp_gpu's keyword parameters (general instantiation): parity tests; (3) kernel trace of
# the saturated-depth tie path after the replay-pool change (487 frames/s in session 4 against 650-756 in round 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s5
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -x -q -m gpu -k "warp or cfg4" > gpurun_out/r04_s5/tests.log 2>&1; echo "warp tests rc=$?"; tail -3 gpurun_out/r04_s5/tests.log
CS_LIB_PATH=$PWD/$C/libcs_gwx.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "warp or cfg4" > gpurun_out/r04_s5/tests_gwx.log 2>&1; echo "gwx warp tests rc=$?"; tail -3 gpurun_out/r04_s5/tests_gwx.log
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s5/ab_1080p.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s5/ab_4k.txt
for L in $C/libcomfystereo_hip.so $C/libcs_gwx.so; do for grp in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pp; CS_LIB_PATH=$PWD/$L timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 3 > /tmp/run.log 2>&1; db=$(find /tmp/pp -name "*.db" | head -1); printf "%-28s %s " "$(basename $L)" $grp; [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "k_gpuwarp" | awk '{print $(NF-4), $(NF-2), $(NF-1), $NF}'; done; done 2>&1 | tee gpurun_out/r04_s5/pmc.txt
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r04_s5/clipped.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/r04_s5/clipped_kernel_trace.txt > /dev/null; head -12 gpurun_out/r04_s5/clipped_kernel_trace.txt | cut -c1-150
