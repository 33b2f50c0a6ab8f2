#!/bin/bash
# round-6 session 23: tile hints from the second tier for polylines_sharp (the lean row kernel on the flagged tiles' columns), the new
# point-kernel test of numba's sweep; polylines / scene8 / tie tests, polylines fuzz; sharp on clipped (saturated) / scene8 / stepped depth
# against the previous binary (libcs_base.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s23; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dialect.py tests/test_gpu_scene8.py -x -q > $O/tests_a.log 2>&1; echo "dialect + scene8 tests rc=$?"; tail -3 $O/tests_a.log
timeout 1200 python -m pytest tests -x -q -m gpu -k "poly or tie or replay or order or sharp" > $O/tests_b.log 2>&1; echo "polylines tests rc=$?"; tail -3 $O/tests_b.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 120 2301 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for i in 1 2; do for L in cs_base comfystereo_hip; do for k in clipped scene8 stepped; do for b in 0 1; do
  printf "%-16s sharp %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 5 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done; done 2>&1 | tee $O/ab_sharp.txt
