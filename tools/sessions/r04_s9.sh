#!/bin/bash
# round-4 session 9: (1) hybrid_edge: forward tiles without the counting sort + branch-free exp: parity (hybrid tests, fuzz), cfg3
# A/B against the always-sort build (dev switch), kernel trace; (2) tie path: retry pass with 256 workgroups instead of 32,
# per-dispatch durations of k_rowwarp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s9
timeout 1500 python -m pytest tests -x -q -m gpu -k "hybrid or cfg3 or fused or node or fuzz or stress or ties or saturated" > gpurun_out/r04_s9/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s9/tests.log
CS_FUZZ_FILLS=hybrid_edge,hybrid_edge_plus timeout 300 python tools/extended_fuzz.py 150 909000 > gpurun_out/r04_s9/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s9/fuzz.log
for rep in 1 2 3; do
  timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 forward tiles:', round(d['value'],1), 'fps, kernel_ms', round(d['roofline']['kernel_ms'],3))"
  (cd tools/_r3 && timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 round-3 tree:  ', round(d['value'],1), 'fps, kernel_ms', round(d['roofline']['kernel_ms'],3))")
done
printf "hybrid radial 4K: "; timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge --kind radial 2>&1 | tail -1 | sed 's/.*: //'
(cd tools/_r3 && printf "hybrid radial 4K (round-3 tree): " && timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge --kind radial 2>&1 | tail -1 | sed 's/.*: //')
printf "hybrid blobs 4K: "; timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge --kind blobs 2>&1 | tail -1 | sed 's/.*: //'
(cd tools/_r3 && printf "hybrid blobs 4K (round-3 tree): " && timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge --kind blobs 2>&1 | tail -1 | sed 's/.*: //')
for v in 0 32; do
  printf "clipped blur off, retry pass PT_VARIANT=$v: "; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r04_s9/clipped.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/r04_s9/clipped_kernel_trace.txt --calls k_rowwarp > /dev/null; head -30 gpurun_out/r04_s9/clipped_kernel_trace.txt | cut -c1-150
