#!/bin/bash
# round-4 session 12: the lean row kernel takes its export scratch from the dead colour row (rows with many layers had no room in the segment-list tail): parity, saturated / noise depth
# (44 KB windows, 3 waves per CU) instead of being replayed in the row kernel: parity (tie tests, fuzz), saturated / noise depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s12
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or ties or saturated or replay or order or fuzz or stress or metric or cfg2 or wide or 8k" > gpurun_out/r04_s12/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s12/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 250 404000 > gpurun_out/r04_s12/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s12/fuzz.log
for rep in 1 2; do
  printf "clipped blur off n=64: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done
printf "clipped blur off polylines_sharp n=32: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 32 --iters 3 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
printf "blobs blur off n=32: "; timeout 300 python tools/quick_bench.py --kind blobs --blur 0 --n 32 --iters 5 2>&1 | tail -1 | sed 's/.*: //'
printf "random8 blur on n=8: "; timeout 300 python tools/quick_bench.py --kind random8 --blur 1 --n 8 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
printf "random8 blur off n=4: "; timeout 600 python tools/quick_bench.py --kind random8 --blur 0 --n 4 --iters 1 2>&1 | tail -2
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r04_s12/clipped.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/r04_s12/clipped_kernel_trace.txt --calls k_rowwarp > /dev/null; head -24 gpurun_out/r04_s12/clipped_kernel_trace.txt | cut -c1-150
