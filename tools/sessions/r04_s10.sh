#!/bin/bash
# round-4 session 10: the lean first pass of the polylines row kernel over flagged rows (64 registers, two rows per CU; rows it
# cannot export go to the retry pass): parity on the tie tests + fuzz with saturated depth, A/B against the full kernel (PT_VARIANT 43)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s10
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or ties or saturated or replay or order or fuzz or stress or metric or cfg2 or wide or 8k" > gpurun_out/r04_s10/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s10/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 150 303000 > gpurun_out/r04_s10/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s10/fuzz.log
for rep in 1 2; do
for v in 0 43; do
  printf "clipped blur off n=64, PT_VARIANT=$v: "; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done
done
for v in 0 43; do
  printf "clipped blur off polylines_sharp n=32, PT_VARIANT=$v: "; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 32 --iters 3 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
  printf "blobs blur off n=32, PT_VARIANT=$v: "; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --kind blobs --blur 0 --n 32 --iters 5 2>&1 | tail -1 | sed 's/.*: //'
  printf "random8 blur on n=8, PT_VARIANT=$v: "; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --kind random8 --blur 1 --n 8 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 bench.py --depth clipped --no-blur --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r04_s10/clipped.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/r04_s10/clipped_kernel_trace.txt --calls k_rowwarp > /dev/null; head -24 gpurun_out/r04_s10/clipped_kernel_trace.txt | cut -c1-150
