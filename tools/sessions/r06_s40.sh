#!/bin/bash
# round-6 session 40: lazy depth-blur tiles for the polylines techniques under a dialect flag: dialect tests, node-level dialect fuzz (blur on in 80 %
# of the cases) for polylines, the metric's workload under D64 against the gated binary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dialect.py tests/test_gpu_lazy_blur.py -x -q > $O/tests.log 2>&1; echo "dialect + lazy tests rc=$?"; tail -3 $O/tests.log
for d in f64-disparity D64; do CS_FUZZ_FILLS=polylines_soft,polylines_sharp CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 70 4001 > $O/fuzz_poly_$d.log 2>&1; echo "fuzz poly $d rc=$?"; tail -1 $O/fuzz_poly_$d.log; done
for L in cs_gated comfystereo_hip cs_gated comfystereo_hip; do
  printf "%-16s soft stepped 64 frames blur 1 D64: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --dialect D64 --iters 4 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done 2>&1 | tee $O/ab.txt
