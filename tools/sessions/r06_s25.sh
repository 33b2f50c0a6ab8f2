#!/bin/bash
# round-6 session 25: polylines_sharp at 8 192 columns (list-capacity threshold 2 048), none_post / inverse_post with their post-fill arrays over
# dead LDS (11 578 / 9 004 columns): the width tests, the hidden-technique tests, fuzz of the post fills and of sharp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "wide or refused or 8192 or 8k" > $O/tests_a.log 2>&1; echo "width tests rc=$?"; tail -3 $O/tests_a.log
timeout 900 python -m pytest tests -x -q -m gpu -k "hidden or post or dropin or parity" > $O/tests_b.log 2>&1; echo "hidden / parity tests rc=$?"; tail -3 $O/tests_b.log
CS_FUZZ_FILLS=none_post,inverse_post,polylines_sharp timeout 300 python tools/extended_fuzz.py 100 2501 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
for d in f64-disparity D64; do CS_FUZZ_FILLS=none_post,inverse_post CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 40 2502 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
