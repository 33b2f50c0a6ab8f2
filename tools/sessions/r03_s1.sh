#!/bin/bash
# round-3 session 1: chunked pre-pass overlap -- parity, then A/B of the chunk count under quick_bench and bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests/test_gpu_chunks.py -x -q -m gpu > gpurun_out/s1/chunks_test.log 2>&1; echo "chunk tests rc=$?"; tail -3 gpurun_out/s1/chunks_test.log
for rep in 1 2; do
for c in 1 2 4 8 16; do
  printf "metric64 chunks=%-2s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 64 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
done
for c in 1 4; do
  printf "cfg5 chunks=%-2s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 64 --blur 1 --iters 10 --fill none --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
  printf "cfg3 chunks=%-2s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge 2>&1 | tail -1 | sed 's/.*: //'
  printf "cfg4 chunks=%-2s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 256 --h 1080 --w 1920 --div 4.5 --kind radial --blur 1 --iters 5 --fill gpu_warp 2>&1 | tail -1 | sed 's/.*: //'
  printf "cfg2 chunks=%-2s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 32 --h 1080 --w 1920 --div 3.5 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err; tail -c 1500 gpurun_out/s1/bench_default.json
