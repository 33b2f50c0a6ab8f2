#!/bin/bash
# round-4 session 17: where the time of 8-bit noise depth goes (blur off / on): kernel traces of the tie path, replay counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s17; mkdir -p $O
for b in off on; do
  fl=$([ $b = off ] && echo "--no-blur --frames 8" || echo "--frames 16")
  rm -rf /tmp/pt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth random8 $fl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$b.json 2>/dev/null
  db=$(find /tmp/pt -name '*.db' | head -1); python3 tools/prof_summary.py $db $O/random8_blur_${b}_kernel_trace.txt > /dev/null
  head -14 $O/random8_blur_${b}_kernel_trace.txt | cut -c1-150
done
CS_DBG=14 timeout 600 python tools/quick_bench.py --n 8 --blur 0 --iters 2 --fill polylines_soft --kind random8 --tie-pool-mb 1024 2>&1 | grep -v amdgpu.ids | tail -6
