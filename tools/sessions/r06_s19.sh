#!/bin/bash
# round-6 session 19: polylines_soft with longer per-pixel lists in the first tier (KP / KS 4 / 6 and 5 / 7 against 4 / 5: + 320 / + 960 bytes
# of LDS -- does the seventh workgroup per CU still fit? -- and 1 - 2 spilled registers): stepped / scene8 / blobs / clipped, 16 and 64 frames;
# then the bench line's CPU legs (OpenMP rows by thread count)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s19; mkdir -p $O
for i in 1 2; do for L in comfystereo_hip cs_ppsk46 cs_ppsk57s; do for k in stepped scene8 blobs clipped; do for b in 0 1; do
  printf "%-16s %-8s blur %s: " $L $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 6 --fill polylines_soft --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done
printf "%-16s stepped 64 frames: " $L; CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 4 --fill polylines_soft --kind stepped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee $O/ab.txt
timeout 900 python bench.py --steps 5 > $O/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); print(round(j['value'],1), 'fps'); print(json.dumps(j['cpu_baseline'], indent=1))"
