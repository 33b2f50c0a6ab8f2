#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 tools/quick_bench.py --n 32 --blur 1 --iters 2 --kind clipped > /tmp/qb.log 2>&1
tail -2 /tmp/qb.log
db=$(find /tmp/pt -name '*.db' | head -1)
python3 - $db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:12])
try:
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
except Exception as e:
    print("err", e); rows = []
for name, s, e in rows:
    if 'rowwarp' in name or 'poly_replay' in name or 'polypoint' in name or 'collect' in name:
        print(f"{name[:40]:40s} {(e-s)/1e3:10.1f} us")
PY
