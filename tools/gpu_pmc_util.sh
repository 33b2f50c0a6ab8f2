#!/bin/bash
# development aid: utilisation counters of the polylines tile kernel (one --pmc pass per group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/util
i=0
for grp in "VALUBusy VALUUtilization SALUBusy" "MemUnitBusy MemUnitStalled WriteUnitStalled" "LDSBankConflict FetchSize WriteSize" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 8 --iters 2 "$@" > gpurun_out/util/run$i.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  python3 tools/prof_summary.py $db gpurun_out/util/g$i.txt --pmc | grep -E "polytile|kernel" | head -5
  i=$((i+1))
done
