#!/bin/bash
# first-call frames/s of the node with host tensors, one fresh process per measurement: three shapes x {cold, opt-in warm-up}
#  -> stdout (copied to profiles/r05_host.txt by the session script); then the steady state of 32 x 4K under
# the default 8 GB pinned cap (pageable results) and with the cap lifted (pinned results)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "32 2160 3840" "8 2160 3840" "96 1080 1920"; do
  set -- $shape
  for pw in 0 1; do
    echo "== $1 frames $3x$2, prewarm=$pw"
    timeout 300 python tools/node_host_bench.py --n $1 --h $2 --w $3 --prewarm $pw --first-only 1 2>&1 | grep -v -E "Warning|amdgpu.ids"
  done
done
echo "== 8 frames 4K cold, pageable results (cap 0)"
timeout 300 python tools/node_host_bench.py --n 8 --prewarm 0 --first-only 1 --pin-cap-gb 0 2>&1 | grep -v -E "Warning|amdgpu.ids"
echo "== steady state, 32 frames 4K, default cap of 8 GB (pageable results)"
timeout 300 python tools/node_host_bench.py --n 32 --prewarm 0 --iters 3 2>&1 | grep -v -E "Warning|amdgpu.ids"
echo "== steady state, 32 frames 4K, cap 64 GB (pinned results: round 4's form)"
timeout 300 python tools/node_host_bench.py --n 32 --prewarm 0 --iters 3 --pin-cap-gb 64 2>&1 | grep -v -E "Warning|amdgpu.ids"
