#!/bin/bash
# round-5 session 26: the lane replay with 64-entry lists (second instantiation of k_poly_replay_lanes: noise depth): tie / polylines tests,
# polylines fuzz (node cases with noise depth), then noise and saturated depth A/B against CS_PT_VARIANT=46 (not tried), kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s26; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or parity or lean or saturated or stretch or fuzz or anaglyph or sharp or order" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 240 818282 > $O/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_poly.log
for v in 0 46; do
  printf "variant %-3s random8 blur 0 (8 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 8 --fill polylines_soft --kind random8 --blur 0 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s random8 blur 1 (16 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 16 --fill polylines_soft --kind random8 --blur 1 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s clipped blur 0 (32 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 32 --fill polylines_soft --kind clipped --blur 0 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s sharp random8 blur 0 (4 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 4 --fill polylines_sharp --kind random8 --blur 0 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
done 2>&1 | tee $O/ab.txt
rm -rf /tmp/pp
CS_DBG=14 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 8 --fill polylines_soft --kind random8 --blur 0 --iters 2 > /tmp/run.log 2>&1
db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_random8.txt > /dev/null; head -8 $O/trace_random8.txt | tail -6 | cut -c1-140
