#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of a bench.py command, summaries only
# (the rocpd databases are large; only the text summaries are kept under gpurun_out/).
#   tools/gpu_profile.sh <tag> <bench args...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python3 bench.py $* --no-cpu-baseline --no-other-depths"
run() { # name, extra rocprof args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 "$@" -d /tmp/prof_$name -o p -- $B > $OUT/$name.log 2>&1
  local db=$(find /tmp/prof_$name -name '*.db' | head -1)
  if [ "$name" = trace ]; then python3 tools/prof_summary.py $db $OUT/kernel_trace.txt > /dev/null
  else python3 tools/prof_summary.py $db $OUT/$name.txt --pmc > /dev/null; fi
  grep '^{' $OUT/$name.log > $OUT/$name.json; rm -f $OUT/$name.log; rm -rf /tmp/prof_$name
}
run trace --kernel-trace --stats
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run pmc_sq2 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
ls -la $OUT
