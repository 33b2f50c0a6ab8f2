#!/bin/bash
# round-5 session 32: 160 / 176 list slots: k_polypoint's time on the headline workload (rocprofv3) and what they do for saturated depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s32; mkdir -p $O
C=comfystereo_amd
for rep in 1 2; do for L in libcs_dcap128.so libcs_dcap160.so libcs_dcap176.so libcomfystereo_hip.so; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 6 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  printf "%-24s " $L; grep -E "k_polypoint" /tmp/t.txt | awk '{printf "%s us\n", $(NF-1)}'
done; done 2>&1 | tee $O/kernel_ab.txt
for b in 1 0; do for L in libcs_dcap128.so libcs_dcap160.so libcs_dcap176.so libcomfystereo_hip.so; do
  printf "%-22s clipped blur $b: " $L; CS_LIB_PATH=$PWD/$C/$L timeout 600 python tools/quick_bench.py --n 32 --fill polylines_soft --kind clipped --blur $b --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done; done 2>&1 | tee $O/ab_clipped.txt
for L in libcs_dcap128.so libcs_dcap160.so libcs_dcap176.so libcomfystereo_hip.so; do printf "%-22s sharp clipped blur 1: " $L; CS_LIB_PATH=$PWD/$C/$L timeout 600 python tools/quick_bench.py --n 32 --fill polylines_sharp --kind clipped --blur 1 --iters 3 2>&1 | tail -1 | sed 's/.*: //'; done 2>&1 | tee -a $O/ab_clipped.txt
