#!/bin/bash
# round-4 session 8: the tie path with two plain atomic adds instead of the compare-and-swap loop (saturated depth, blur off)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s8
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "ties or saturated or replay or order or clipped" > gpurun_out/r04_s8/tests.log 2>&1; echo "tie tests rc=$?"; tail -3 gpurun_out/r04_s8/tests.log
for rep in 1 2; do
(cd tools/_r3 && timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-3 tree: clipped blur off', round(d['value'],1), 'fps')")
timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | tee gpurun_out/r04_s8/bench_clipped_blur_off.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current:      clipped blur off', round(d['value'],1), 'fps')"
done
CS_DBG=14 timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 16 --iters 3 2>&1 | tail -4
timeout 600 python bench.py --depth clipped --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | tee gpurun_out/r04_s8/bench_clipped_blur_on.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current:      clipped blur on', round(d['value'],1), 'fps')"
timeout 900 python bench.py --depth random8 --no-blur --no-cpu-baseline --frames 8 --steps 2 --warmup 1 2>/dev/null | tail -1 | tee gpurun_out/r04_s8/bench_random8_blur_off.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current:      random8 blur off (8 frames)', round(d['value'],2), 'fps')"
