#!/bin/bash
# round-6 session 2: k_gpuwarp_q (four contiguous columns per lane) -- the gpu_warp tests, a gpu_warp fuzz, then A/B against
# k_gpuwarp (CS_PT_VARIANT=27) at 1080p (cfg 4's shape, 128 frames) and 4K, blur off and on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s2; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "warp or node or cfg4 or lazy or 8k or dropin or chunks or fuzz or stress" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
CS_FUZZ_FILLS=gpu_warp timeout 200 python tools/extended_fuzz.py 90 6161 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
for i in 1 2 3; do
  for v in 0 27; do
    printf "1080p blur0 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 0 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
    printf "1080p blur1 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 1 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
    printf "4K    blur0 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill gpu_warp 2>&1 | tail -1 | sed 's/.*: //'
  done
done 2>&1 | tee $O/ab.txt
timeout 600 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_cfg4.json')); r=j['roofline']; print('cfg4', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3))"
