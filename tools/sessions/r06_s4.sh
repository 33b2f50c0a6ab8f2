#!/bin/bash
# round-6 session 4: instruction counters of k_gpuwarp_q per phase cut-off (dev build: CS_DBG 52 = after stage + pairs, 53 = after the
# column pass, 54 = after the rightmost reduction, 0 = whole kernel), 1080p radial, 16 frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06_s4}; mkdir -p $O
export CS_LIB_PATH=$GRAFT_REPO_ROOT/comfystereo_amd/libcomfystereo_hip_dev.so
for d in 52 53 54 0; do
  rm -rf /tmp/pp
  CS_DBG=$d timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --iters 2 > $O/run$d.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  python3 tools/prof_summary.py $db $O/dbg$d.txt --pmc > /dev/null
  python3 - "$d" $O/dbg$d.txt <<'PY'
import sys
d, path = sys.argv[1], sys.argv[2]
v = {}
for ln in open(path):
    if "k_gpuwarp_q<" in ln:
        parts = ln.split()
        v[parts[-5]] = float(parts[-1])
w = v.get("SQ_WAVES", 1)
print(f"dbg={d:>2}: per wave VALU {v.get('SQ_INSTS_VALU',0)/w:7.1f}  VALU-busy quad-cycles {v.get('SQ_ACTIVE_INST_VALU',0)/w:7.1f}  SALU {v.get('SQ_INSTS_SALU',0)/w:6.1f}  "
      f"LDS {v.get('SQ_INSTS_LDS',0)/w:5.1f}  wave life {v.get('SQ_WAVE_CYCLES',0)/w*4:8.0f} cyc")
PY
done 2>&1 | tee $O/phases.txt
