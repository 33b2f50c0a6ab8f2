#!/bin/bash
# round-5 session 11: where the dialect instantiations still differ (sharp fixture case, the dialect fuzz's first mismatch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s11; mkdir -p $O
timeout 600 python tools/sessions/r05_s11_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/debug_sharp.txt
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_dialect.py -x -q -m gpu > $O/tests_dialect_$i.log 2>&1; echo "dialect tests run $i rc=$?"; tail -3 $O/tests_dialect_$i.log; done
for d in f64-disparity D64; do
  CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 120 818181 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -2 $O/fuzz_$d.log
done
