#!/bin/bash
# development aid: hardware counters of the polylines tile kernel (one --pmc pass per group, each under a timeout)
#   tools/gpu_pmc_util.sh "<group1 counters>;<group2 counters>;..." <quick_bench args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/util
IFS=';' read -ra GROUPS_ <<< "$1"; shift
i=0
for grp in "${GROUPS_[@]}"; do
  rm -rf /tmp/pp
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 8 --iters 2 "$@" > gpurun_out/util/run$i.log 2>&1
  echo "group $i ($grp): rc=$?"
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db gpurun_out/util/g$i.txt --pmc | grep -E "polytile" | sed 's/.*PolyTileArgs)//'
  i=$((i+1))
done
