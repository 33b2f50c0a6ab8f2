#!/bin/bash
# round-5 session 6: k_gpuwarp -- gap scan on bit rows (libcs_gwbits.so) + pass-1 clean-up (the default build) against the round-4 kernel
# no `>= 0` test per round): warp tests + fuzz, A/B against the previous kernel (libcs_gwold.so) at 1080p and 4K, instruction
# counters per phase cut-off (dev build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s6; mkdir -p $O
C=comfystereo_amd
timeout 1200 python -m pytest tests -x -q -m gpu -k "warp or gpu_warp or cfg4 or node or golden or forward" > $O/tests_warp.log 2>&1; echo "warp tests rc=$?"; tail -3 $O/tests_warp.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 150 717171 > $O/fuzz_warp.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_warp.log
LIBS="$C/libcs_gwold.so $C/libcs_gwbits.so $C/libcomfystereo_hip.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 0 --iters 10 2>&1 | tee $O/ab_1080p.txt
LIBS="$C/libcs_gwold.so $C/libcs_gwbits.so $C/libcomfystereo_hip.so" tools/abn.sh --n 16 --fill gpu_warp --blur 0 --iters 10 2>&1 | tee $O/ab_4k.txt
LIBS="$C/libcs_gwold.so $C/libcs_gwbits.so $C/libcomfystereo_hip.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind blobs --div 8 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p_blobs.txt
for L in cs_gwold cs_gwbits comfystereo_hip; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/lib$L.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 0 --iters 10 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$L.txt > /dev/null
  printf "%-20s " $L; grep "k_gpuwarp" $O/trace_$L.txt | awk '{print $(NF-3), $(NF-2), $(NF-1)}'
done 2>&1 | tee $O/kernel_times.txt
bash tools/gpu_pmc_gw_phases.sh 2>&1 | tee $O/phases.txt
