#!/bin/bash
# round-6 session 26: naive / naive_interpolating with their scratch arrays over the dead normalised depth (11 578 / 9 536 columns): width tests,
# every -m gpu test, smoke, fuzz of the forward fills (D32 + the two dialect settings)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "wide or refused or 8192 or 8k" > $O/tests_a.log 2>&1; echo "width tests rc=$?"; tail -3 $O/tests_a.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
CS_FUZZ_FILLS=none,naive,naive_interpolating,inverse,none_post,inverse_post timeout 300 python tools/extended_fuzz.py 100 2601 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
for d in int64-sum D64; do CS_FUZZ_FILLS=naive,naive_interpolating CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 40 2602 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
