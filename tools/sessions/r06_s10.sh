#!/bin/bash
# round-6 session 10: why are the polylines techniques slow on scene8 depth (sharp 1 060 against 4 050 frames/s on stepped depth, soft 4 200
# against 5 380)?  rows flagged / replayed (plan.stats()), the hazard reasons (dev build: stats word 12, dbg 15 event counts), kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s10; mkdir -p $O
for f in polylines_soft polylines_sharp; do for k in scene8 stepped; do
  echo "== $f $k (release)"; timeout 300 python tools/quick_bench.py --n 8 --blur 0 --iters 3 --fill $f --kind $k 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400
done; done 2>&1 | tee $O/stats.txt
export CS_LIB_PATH=$GRAFT_REPO_ROOT/comfystereo_amd/libcomfystereo_hip_dev.so
for f in polylines_soft polylines_sharp; do
  echo "== $f scene8 (dev, dbg 15: reason bits + events per reason class)"; CS_DBG=15 timeout 300 python tools/quick_bench.py --n 8 --blur 0 --iters 1 --fill $f --kind scene8 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600
done 2>&1 | tee $O/reasons.txt
unset CS_LIB_PATH
for f in polylines_sharp polylines_soft; do
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --blur 0 --iters 4 --fill $f --kind scene8 > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$f.txt > /dev/null; head -9 $O/trace_$f.txt | cut -c1-150
done
