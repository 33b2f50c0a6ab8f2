#!/bin/bash
# round-6 session 11: second tier of k_polypoint (k_polypoint_listed: flagged rows once more with 512 dirty slots and longer lists):
# polylines / scene8 / tie tests, polylines fuzz, then scene8 / stepped / clipped / blobs, soft and sharp, A/B against one tier (CS_PT_VARIANT=49)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06_s11}; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -k "polylines or scene8 or tie or replay or lean or saturated or stretch or order or anaglyph or fullsize or parity or lazy or 8k" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 120 1111 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for v in 0 49; do for k in scene8 stepped clipped blobs; do for f in polylines_soft polylines_sharp; do for b in 0 1; do
  printf "variant %2d %-8s %-16s blur %s: " $v $k $f $b; CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill $f --kind $k 2>&1 | tail -1 | sed 's/.*: //'
done; done; done; done 2>&1 | tee $O/ab.txt
for f in polylines_sharp polylines_soft; do
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --blur 0 --iters 4 --fill $f --kind scene8 > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$f.txt > /dev/null; head -9 $O/trace_$f.txt | cut -c1-150
done
