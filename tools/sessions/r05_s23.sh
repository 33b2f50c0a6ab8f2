#!/bin/bash
# round-5 session 23: phases of the lean row kernel on saturated depth after the column-range restriction (dev build, CS_DBG 1..5), whole rows
# (CS_PT_VARIANT=44) beside it; what the lane replay kernel hands on to the wave kernel (CS_DBG=14 counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s23; mkdir -p $O
C=comfystereo_amd
for v in 0 44; do for d in 0 1 2 3 4 5; do
  rm -rf /tmp/pp
  CS_PT_VARIANT=$v CS_DBG=$d CS_LIB_PATH=$PWD/$C/libcomfystereo_hip_dev.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --fill polylines_soft --kind clipped --blur 0 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_v${v}_dbg$d.txt > /dev/null
  printf "variant %s dbg=%s " $v $d; grep -E "k_rowwarp<3, false, true>" $O/trace_v${v}_dbg$d.txt | awk '{printf "%s us\n", $(NF-1)}'
done; done 2>&1 | tee $O/rowkernel_phases.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/replay_counters.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch, synth
from comfystereo_amd import engine, _native
_native.debug_set("dbg", 14)
n, h, w = 8, 2160, 3840
img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).cuda()
for kind in ("clipped", "blobs"):
    depth = torch.from_numpy(synth.depth_batch(kind, n, h, w, channels=3)).cuda()
    p = engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    plan.run(img, depth); torch.cuda.synchronize()
    st = plan.stats()
    print(kind, "stats words 9..15 summed over frames:", [int(st[:, k].sum()) for k in range(9, 16)])
PY
