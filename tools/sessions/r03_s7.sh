#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
PY
for rep in 1 2; do
for c in 1 4 104 108 102; do
  printf "metric64 chunks=%-3s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 64 --blur 1 --iters 10 2>&1 | tail -1 | sed 's/.*: //'
done
done
for c in 1 104; do
  printf "cfg5 chunks=%-3s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 64 --blur 1 --iters 10 --fill none --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
  printf "cfg3 chunks=%-3s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 16 --blur 1 --iters 10 --fill hybrid_edge 2>&1 | tail -1 | sed 's/.*: //'
  printf "cfg4 chunks=%-3s " $c; CS_CHUNKS=$c timeout 300 python tools/quick_bench.py --n 256 --h 1080 --w 1920 --div 4.5 --kind radial --blur 1 --iters 5 --fill gpu_warp 2>&1 | tail -1 | sed 's/.*: //'
done
