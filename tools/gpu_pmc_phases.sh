#!/bin/bash
# development aid: instruction counters of the polylines tile kernel per phase cutoff (CS_DBG)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/phases
for d in 11 12 13 0; do
  rm -rf /tmp/pp
  CS_DBG=$d CS_PT_VARIANT=${CS_PT_VARIANT:-3} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 8 --iters 2 > /dev/null 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  python3 tools/prof_summary.py $db gpurun_out/phases/dbg$d.txt --pmc | grep polytile | awk -v d=$d '{print "dbg=" d, $3, $5, $6}'
done
