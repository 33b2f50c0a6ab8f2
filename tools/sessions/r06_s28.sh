#!/bin/bash
# round-6 session 28: polylines_sharp, first pass over flagged rows in the LEAN row kernel at 4K (list capacity that lets two rows share a CU,
# rows evaluated in two column ranges) instead of the full kernel: polylines / scene8 / tie / 8K tests, polylines fuzz, A/B on clipped /
# scene8 / stepped / random8 depth against the round's first binary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s28; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or tie or replay or order or sharp or scene8 or 8k or 8192 or wide" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_sharp timeout 300 python tools/extended_fuzz.py 120 2801 > $O/fuzz_sharp.log 2>&1; echo "fuzz sharp rc=$?"; tail -1 $O/fuzz_sharp.log
for i in 1 2; do for L in cs_base comfystereo_hip; do for k in clipped scene8 stepped; do for b in 0 1; do
  printf "%-16s sharp %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 5 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done
printf "%-16s sharp random8 4 frames: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 4 --iters 3 --fill polylines_sharp --kind random8 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab_sharp.txt
bash tools/gpu_trace.sh r06_s28/sharp_clipped tools/quick_bench.py --n 16 --fill polylines_sharp --kind clipped --iters 4 | cut -c1-160
