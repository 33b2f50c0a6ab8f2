#!/bin/bash
# round-4 session 14: sliding replay windows of 512 positions / 1024 columns with a one-ballot coverage test, 5 waves per SIMD;
# lean row kernel only where two rows share a CU; noise depth with a pool that holds every row (--tie-pool-mb): whole suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s14
C=comfystereo_amd
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s14/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s14/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 200 505000 > gpurun_out/r04_s14/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s14/fuzz.log
LIBS="$C/libcomfystereo_hip.so $C/libcs_rp4.so" tools/abn.sh --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tee gpurun_out/r04_s14/ab_clipped.txt
printf "clipped blur off polylines_sharp n=32: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 32 --iters 3 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
printf "blobs blur off n=32: "; timeout 300 python tools/quick_bench.py --kind blobs --blur 0 --n 32 --iters 5 2>&1 | tail -1 | sed 's/.*: //'
printf "random8 blur on n=8: "; timeout 300 python tools/quick_bench.py --kind random8 --blur 1 --n 8 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
printf "random8 blur off n=4, pool for every row: "; CS_DBG=14 timeout 600 python tools/quick_bench.py --kind random8 --blur 0 --n 4 --iters 1 --tie-pool-mb 1000 2>&1 | tail -4
printf "random8 blur off n=8 polylines_sharp, pool for every row: "; timeout 600 python tools/quick_bench.py --kind random8 --blur 0 --n 4 --iters 1 --tie-pool-mb 1500 --fill polylines_sharp 2>&1 | tail -2
timeout 900 python bench.py --depth random8 --no-blur --no-cpu-baseline --frames 16 --steps 2 --warmup 1 2>/dev/null | tail -1 | tee gpurun_out/r04_s14/bench_random8_blur_off.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench random8 blur off (16 frames)', round(d['value'],2), 'fps', d['diagnostics'])"
