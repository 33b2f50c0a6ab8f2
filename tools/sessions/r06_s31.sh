#!/bin/bash
# round-6 session 31: order-dependent rows under numba s sweep typing replayed in stretches (poly_stretch64, a lane per stretch) instead of whole rows by one lane: dialect tests,
# dialect tests, polylines + hybrid_edge_plus fuzz under int64-sum / D64, D64 speed on stepped / scene8 / clipped depth (s30: clipped 13.7 / 10.0, scene8 with the blur 874 / 359 frames/s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s31; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dialect.py -x -q > $O/tests_dialect.log 2>&1; echo "dialect tests rc=$?"; tail -3 $O/tests_dialect.log
for d in int64-sum D64; do CS_FUZZ_FILLS=polylines_soft,polylines_sharp,hybrid_edge_plus CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 90 3005 > $O/fuzz_poly_$d.log 2>&1; echo "fuzz poly $d rc=$?"; tail -1 $O/fuzz_poly_$d.log; done
for f in polylines_soft polylines_sharp; do for k in stepped scene8 clipped; do for b in 0 1; do
  printf "%-16s %-8s blur %s D64: " $f $k $b; timeout 300 python tools/quick_bench.py --n 16 --fill $f --kind $k --blur $b --dialect D64 --iters 4 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done 2>&1 | tee $O/d64.txt
