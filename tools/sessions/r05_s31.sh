#!/bin/bash
# round-5 session 31: k_polypoint's own time with 200 / 128 / 160 list slots on the headline workload (rocprofv3, alternating, three rounds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s31; mkdir -p $O
C=comfystereo_amd
for rep in 1 2 3; do for L in libcomfystereo_hip.so libcs_dcap128.so libcs_dcap160.so; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 6 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/t.txt > /dev/null
  printf "%-24s " $L; grep -E "k_polypoint" /tmp/t.txt | awk '{printf "%s us\n", $(NF-1)}'
done; done 2>&1 | tee $O/kernel_ab.txt
