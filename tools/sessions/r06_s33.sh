#!/bin/bash
# round-6 session 33: gate on the FINAL binary of the round's second session: every -m gpu test, smoke, fuzz over every technique (240 s),
# polylines (100 s), forward + post fills (60 s), gpu_warp (60 s), 50 s per dialect setting; D64 speed lines (stepped / scene8 / clipped)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s33; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 400 python tools/extended_fuzz.py 240 3301 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 100 3302 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
CS_FUZZ_FILLS=none,naive,naive_interpolating,inverse,none_post,inverse_post timeout 300 python tools/extended_fuzz.py 60 3303 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 60 3304 > $O/fuzz_gw.log 2>&1; echo "fuzz gw rc=$?"; tail -1 $O/fuzz_gw.log
for d in f64-disparity int64-sum D64; do CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 50 3305 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
for f in polylines_soft polylines_sharp; do for k in stepped scene8 clipped; do for b in 0 1; do
  printf "%-16s %-8s blur %s D64: " $f $k $b; timeout 300 python tools/quick_bench.py --n 16 --fill $f --kind $k --blur $b --dialect D64 --iters 4 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done 2>&1 | tee $O/d64.txt
