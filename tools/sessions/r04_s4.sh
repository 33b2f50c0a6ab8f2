#!/bin/bash
# round-4 session 4: polylines_sharp on the point-owner kernel (k_polypoint<SHARP>): whole GPU suite (+ the new stress tests),
# polylines-only extended fuzz, throughput against the first-generation kernel (CS_PT_VARIANT=1), the tie path after the
# replay-pool change (saturated depth, blur off)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s4
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s4/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r04_s4/tests.log
CS_FUZZ_FILLS=polylines_sharp,polylines_soft timeout 400 python tools/extended_fuzz.py 240 808000 > gpurun_out/r04_s4/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s4/fuzz.log
for rep in 1 2; do
for v in 0 1; do
  for b in 0 1; do
    printf "sharp PT_VARIANT=%s blur=%s: " $v $b; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 32 --blur $b --iters 10 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
  done
done
done
printf "soft blur=0: "; timeout 300 python tools/quick_bench.py --n 32 --blur 0 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
printf "sharp blobs blur=1: "; timeout 300 python tools/quick_bench.py --n 32 --blur 1 --iters 10 --fill polylines_sharp --kind blobs 2>&1 | tail -3
timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_s4/bench_clipped_blur_off.json; python -c "
import json; d=json.load(open('gpurun_out/r04_s4/bench_clipped_blur_off.json')); print('clipped blur off', round(d['value'],1), 'fps', d['diagnostics'])"
