#!/bin/bash
# development aid: instruction counters of the point-owner polylines kernel per phase cut-off (needs the -DCS_DEV build:
# make -C comfystereo_amd/csrc dev).   tools/gpu_pmc_phases.sh [quick_bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/phases
export CS_LIB_PATH=$GRAFT_REPO_ROOT/comfystereo_amd/libcomfystereo_hip_dev.so
for d in ${PHASES:-31 32 33 34 35 0}; do
  rm -rf /tmp/pp
  CS_DBG=$d timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 8 --iters 2 "$@" > gpurun_out/phases/run$d.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  python3 tools/prof_summary.py $db gpurun_out/phases/dbg$d.txt --pmc > /dev/null
  python3 - "$d" gpurun_out/phases/dbg$d.txt <<'PY'
import sys
d, path = sys.argv[1], sys.argv[2]
v = {}
for ln in open(path):
    if "polypoint" in ln or "polytile" in ln:
        parts = ln.split()
        v[parts[-5]] = float(parts[-1])
w = v.get("SQ_WAVES", 1)
print(f"dbg={d:>2}: per wave VALU {v.get('SQ_INSTS_VALU',0)/w:7.1f}  VALU-busy quad-cycles {v.get('SQ_ACTIVE_INST_VALU',0)/w:7.1f}  SALU {v.get('SQ_INSTS_SALU',0)/w:6.1f}  "
      f"LDS {v.get('SQ_INSTS_LDS',0)/w:5.1f}  wave life {v.get('SQ_WAVE_CYCLES',0)/w*4:8.0f} cyc  wait_any {v.get('SQ_WAIT_ANY',0)/w*4:7.0f}  wait_inst {v.get('SQ_WAIT_INST_ANY',0)/w*4:7.0f}")
PY
done
