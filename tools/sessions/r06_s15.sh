#!/bin/bash
# round-6 session 15: second tier as a PLAIN launch over the list (no persistent loop around the inlined tile function: 96 / 85 registers,
# 2 / 0 spilled, five workgroups per CU): polylines / scene8 / tie tests, polylines fuzz, then scene8 / stepped / blobs / clipped for
# sharp (tier on = default / off = 49) and soft (default = off / forced = 50): what does the tier cost when the list is empty (stepped)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s15; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -k "polylines or scene8 or tie or replay or lean or saturated or stretch or order or anaglyph or fullsize or sharded" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 100 1515 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for i in 1 2; do for f in polylines_sharp polylines_soft; do for v in 0 49 50; do for k in scene8 stepped blobs; do for b in 0 1; do
  printf "%-16s variant %2d %-8s blur %s: " $f $v $k $b
  CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill $f --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done; done; done 2>&1 | tee $O/ab.txt
for f in polylines_sharp; do
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --blur 0 --iters 4 --fill $f --kind scene8 > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$f.txt > /dev/null; head -9 $O/trace_$f.txt | cut -c1-150
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 64 --blur 0 --iters 3 --fill $f --kind stepped > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_${f}_stepped64.txt > /dev/null; head -6 $O/trace_${f}_stepped64.txt | cut -c1-150
done
