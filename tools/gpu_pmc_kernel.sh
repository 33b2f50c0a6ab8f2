#!/bin/bash
# development aid: hardware counters of one kernel under tools/quick_bench.py, one --pmc pass per group
#   tools/gpu_pmc_kernel.sh <tag> <kernel name pattern> "<group1>;<group2>;..." <quick_bench args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; PAT=$2; IFS=';' read -ra GROUPS_ <<< "$3"; shift 3
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for grp in "${GROUPS_[@]}"; do
  rm -rf /tmp/pp
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py "$@" > $OUT/run$i.log 2>&1
  echo "group $i ($grp): rc=$?"
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $OUT/g$i.txt --pmc | grep -E "$PAT" | awk '{print "   ", $(NF-2), $(NF-1), $NF}'
  i=$((i+1))
done
