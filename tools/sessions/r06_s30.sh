#!/bin/bash
# round-6 session 30: second tier of the point kernel under numba's sweep typing (k_polypoint_listed<..., SW>, both techniques): dialect tests,
# polylines fuzz under int64-sum / D64, D64 speed on stepped / scene8 / clipped depth (s22: stepped 3 606 / 2 705, scene8 2 435 / 1 336)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dialect.py -x -q > $O/tests_dialect.log 2>&1; echo "dialect tests rc=$?"; tail -3 $O/tests_dialect.log
for d in int64-sum D64; do CS_FUZZ_FILLS=polylines_soft,polylines_sharp CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 90 3005 > $O/fuzz_poly_$d.log 2>&1; echo "fuzz poly $d rc=$?"; tail -1 $O/fuzz_poly_$d.log; done
for f in polylines_soft polylines_sharp; do for k in stepped scene8 clipped; do for b in 0 1; do
  printf "%-16s %-8s blur %s D64: " $f $k $b; timeout 300 python tools/quick_bench.py --n 16 --fill $f --kind $k --blur $b --dialect D64 --iters 4 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done 2>&1 | tee $O/d64.txt
