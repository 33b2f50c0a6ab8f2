#!/bin/bash
# round-5 session 22: the tie path -- lean row kernel on the flagged tiles' column ranges without the redundant depth-map rewrite, and the
# stretches replayed by a lane each (k_poly_replay_lanes) before the wave kernel: tie / polylines tests, polylines fuzz, then A/B:
# default / whole rows (CS_PT_VARIANT=44) / wave replay only (45) on saturated and noise depth, kernel traces
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s22c; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or parity or lean or saturated or stretch or fuzz or anaglyph or sharp or order" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 240 717171 > $O/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_poly.log
for v in 0 44 45; do for k in clipped; do for b in 0 1; do
  printf "variant %-3s %-8s blur $b: " $v $k; CS_PT_VARIANT=$v timeout 600 python tools/quick_bench.py --n 32 --fill polylines_soft --kind $k --blur $b --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee $O/ab.txt
for v in 0 45; do
  printf "variant %-3s random8 blur 1 (16 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 16 --fill polylines_soft --kind random8 --blur 1 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s sharp stepped blur 0: " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 32 --fill polylines_sharp --kind stepped --blur 0 --iters 5 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s random8 blur 0 (8 frames): " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 8 --fill polylines_soft --kind random8 --blur 0 --iters 2 2>&1 | tail -1 | sed 's/.*: //'
  printf "variant %-3s sharp clipped blur 0: " $v; CS_PT_VARIANT=$v timeout 900 python tools/quick_bench.py --n 16 --fill polylines_sharp --kind clipped --blur 0 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done 2>&1 | tee -a $O/ab.txt
for v in 0 45; do
  rm -rf /tmp/pp
  CS_PT_VARIANT=$v CS_DBG=14 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --fill polylines_soft --kind clipped --blur 0 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_v$v.txt > /dev/null; echo "variant $v"; head -8 $O/trace_v$v.txt | tail -6 | cut -c1-140
done
