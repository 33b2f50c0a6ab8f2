#!/bin/bash
# round-5 session 21: the lean row kernel on the flagged tiles' column ranges (tile hints of k_polypoint): tie / polylines tests, fuzz with
# saturated / noise depth in the node cases, then the tie-path bench lines A/B against whole rows (CS_PT_VARIANT=44)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s21; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or parity or lean or saturated or stretch or fuzz or anaglyph or sharp" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 240 616161 > $O/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_poly.log
for v in 0 44; do for k in clipped blobs; do
  printf "variant %-3s %-8s blur 0: " $v $k; CS_PT_VARIANT=$v timeout 600 python tools/quick_bench.py --n 32 --fill polylines_soft --kind $k --blur 0 --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done; done 2>&1 | tee $O/ab.txt
for v in 0 44; do
  rm -rf /tmp/pp
  CS_PT_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --fill polylines_soft --kind clipped --blur 0 --iters 3 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1); [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_v$v.txt > /dev/null; echo "variant $v"; head -7 $O/trace_v$v.txt | tail -5 | cut -c1-140
done
