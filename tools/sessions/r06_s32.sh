#!/bin/bash
# round-6 session 32: diagnostics of the stretch form under numba's sweep typing (CS_DBG=14: row-eyes done in stretches / stretches / whole-row
# fallbacks / give-up reasons) on clipped and scene8 depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s32; mkdir -p $O
for f in polylines_soft polylines_sharp; do for k in clipped scene8; do
  echo "== $f $k"; CS_DBG=14 timeout 300 python tools/quick_bench.py --n 4 --fill $f --kind $k --dialect D64 --iters 1 2>&1 | grep "chain px\|seq-fallback\|fps"
done; done 2>&1 | tee $O/diag.txt
