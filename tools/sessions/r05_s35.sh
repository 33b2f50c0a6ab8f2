#!/bin/bash
# round-5 session 35: naive_interpolating, second tier (flagged rows through k_fwdtile once more with a window of 2 halo + 16): the tests that
# name the technique + the new one, a fuzz slice of the forward fills, then saturated / blobs / stepped depth A/B against one tier (CS_PT_VARIANT=47)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s35; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "naive or fwd or forward or fuzz or golden or node or dialect or second_tier or cfg5 or lazy" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=naive_interpolating,naive,none timeout 400 python tools/extended_fuzz.py 200 737373 > $O/fuzz_fwd.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_fwd.log
CS_FUZZ_DIALECT=D64 CS_FUZZ_FILLS=naive_interpolating,naive timeout 300 python tools/extended_fuzz.py 80 747474 > $O/fuzz_fwd_d64.log 2>&1; echo "fuzz D64 rc=$?"; tail -1 $O/fuzz_fwd_d64.log
for v in 0 47; do for k in clipped blobs stepped; do for b in 0 1; do
  printf "variant %-3s %-8s blur %s: " $v $k $b; CS_PT_VARIANT=$v timeout 600 python tools/quick_bench.py --n 32 --fill naive_interpolating --kind $k --blur $b --iters 4 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee $O/ab.txt
