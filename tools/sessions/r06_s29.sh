#!/bin/bash
# round-6 session 29: rows evaluated in column ranges keep the row's longest per-pixel list, so that their stretches can go to the LANE replay
# (s28: sharp on saturated depth went to the wave replay, 710 -> 676 frames/s): polylines / tie tests, sharp fuzz, A/B as in s28
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s29; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or tie or replay or order or sharp or scene8 or 8k or 8192 or wide" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_sharp,polylines_soft timeout 300 python tools/extended_fuzz.py 120 2901 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for i in 1 2; do for L in cs_base comfystereo_hip; do for k in clipped scene8; do for b in 0 1; do
  printf "%-16s sharp %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 5 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done
printf "%-16s sharp random8 4 frames: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 4 --iters 3 --fill polylines_sharp --kind random8 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
printf "%-16s soft clipped blur 0: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --iters 5 --fill polylines_soft --kind clipped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab_sharp.txt
bash tools/gpu_trace.sh r06_s29/sharp_clipped tools/quick_bench.py --n 16 --fill polylines_sharp --kind clipped --iters 4 | cut -c1-160
