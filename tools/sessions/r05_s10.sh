#!/bin/bash
# round-5 session 10: (1) debug: polylines_sharp under the float64 chain in the tile kernel against the reference fixture; (2) the
# polylines row kernel's phases on saturated depth (dev build, CS_DBG 1..5 cut-offs of technique_polylines: where do the 24 ms go?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s10; mkdir -p $O
timeout 300 python tools/sessions/r05_s10_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/debug_sharp_dia.txt
C=comfystereo_amd
for d in 0 1 2 3 4 5; do
  rm -rf /tmp/pp
  CS_DBG=$d CS_LIB_PATH=$PWD/$C/libcomfystereo_hip_dev.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --fill polylines_soft --kind clipped --blur 0 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_dbg$d.txt > /dev/null
  printf "dbg=%s " $d; grep -E "k_rowwarp|k_poly_replay|k_polypoint" $O/trace_dbg$d.txt | awk '{printf "%s %s us | ", substr($0,1,40), $(NF-1)} END {print ""}'
done 2>&1 | tee $O/rowkernel_phases.txt
