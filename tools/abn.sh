#!/bin/bash
# development aid (on the GPU box): alternate N builds of the library under tools/quick_bench.py, optionally with the HBM
# traffic counters of one kernel.   LIBS="a.so b.so c.so" [PMC=k_polypoint] tools/abn.sh [quick_bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for L in $LIBS; do
    printf "%-28s " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 200 python tools/quick_bench.py "$@" 2>&1 | tail -1 | sed 's/.*: //'
  done
done
if [ -n "$PMC" ]; then
  for L in $LIBS; do
    for grp in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pp
      CS_LIB_PATH=$PWD/$L timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py "$@" > /tmp/run.log 2>&1
      db=$(find /tmp/pp -name '*.db' | head -1)
      printf "%-28s " "$(basename $L)"
      [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "$PMC" | awk '{print $(NF-4), $(NF-2), $(NF-1), $NF}'
    done
  done
fi
