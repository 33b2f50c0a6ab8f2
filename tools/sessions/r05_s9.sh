#!/bin/bash
# round-5 session 9: the dialect instantiations of k_hybrid_splat_tile and k_polypoint<SHARP> (float64 disparity chain at tile speed):
# the dialect tests, a fuzz slice under each dialect setting, then every -m gpu test; throughput of the dialect paths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dialect.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests_dialect.log 2>&1; echo "dialect tests rc=$?"; tail -4 $O/tests_dialect.log
for d in f64-disparity D64 int64-sum; do
  CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 150 818181 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -2 $O/fuzz_$d.log
done
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
{
for d in f64-disparity D64; do for f in polylines_soft polylines_sharp hybrid_edge; do
  printf "%-28s " "$f $d"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill $f --dialect $d 2>&1 | tail -1 | sed 's/.*: //'
done; done
printf "%-28s " "hybrid_edge D32"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill hybrid_edge 2>&1 | tail -1 | sed 's/.*: //'
} 2>&1 | tee $O/table_dialect.txt
