#!/bin/bash
# round-6 session 22: k_gpuwarp_q with the three-instruction pre-test (pair mirrored where `safe` is negative: one unsigned compare):
# gpu_warp tests, 90 s of gpu_warp fuzz, A/B of cfg 4 (256 x 1080p) and 4K against the previous binary (libcs_base.so), three alternations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s22; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "warp or cfg4 or gpu_warp" > $O/tests_gw.log 2>&1; echo "gpu_warp tests rc=$?"; tail -2 $O/tests_gw.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 90 2201 > $O/fuzz_gw.log 2>&1; echo "fuzz gw rc=$?"; tail -1 $O/fuzz_gw.log
for i in 1 2 3; do for L in cs_base comfystereo_hip; do
  printf "%-16s 1080p x 128: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind stepped --iters 8 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
  printf "%-16s 4K x 32:      " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 32 --fill gpu_warp --kind stepped --iters 8 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab.txt
# numba's typing of the polylines sweep in the point kernel (k_polypoint<..., DIA, SW>): dialect tests, polylines fuzz under int64-sum and D64,
# speed of 16 x 4K against the previous binary (the general row kernel)
timeout 900 python -m pytest tests/test_gpu_dialect.py -x -q > $O/tests_dialect.log 2>&1; echo "dialect tests rc=$?"; tail -3 $O/tests_dialect.log
for d in int64-sum D64; do CS_FUZZ_FILLS=polylines_soft,polylines_sharp CS_FUZZ_DIALECT=$d timeout 300 python tools/extended_fuzz.py 100 2205 > $O/fuzz_poly_$d.log 2>&1; echo "fuzz poly $d rc=$?"; tail -2 $O/fuzz_poly_$d.log; done
for L in cs_base comfystereo_hip; do for f in polylines_soft polylines_sharp; do for k in stepped scene8; do
  printf "%-16s %-16s %-8s D64: " $L $f $k; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --fill $f --kind $k --dialect D64 --iters 4 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done; done 2>&1 | tee $O/d64.txt
