#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for blur in 0 1; do for n in 8 32; do for sw in 0 1; do
  printf "clipped blur=$blur n=$n no_replay_kernel=$sw: "; CS_NO_REPLAY_KERNEL=$sw timeout 600 python tools/quick_bench.py --n $n --blur $blur --iters 2 --kind clipped 2>&1 | tail -1 | sed 's/.*: //'
done; done; done
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 32 --blur 0 --iters 1 --kind clipped > /tmp/qb.log 2>&1
db=$(find /tmp/pt -name '*.db' | head -1)
python3 - $db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
for name, s, e in rows[-8:]:
    print(f"{name[:40]:40s} {(e-s)/1e3:10.1f} us")
PY
