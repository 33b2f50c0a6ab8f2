#!/bin/bash
# round-4 session 16: the replay kernel's whole-window fast path (tie tests, fuzz, saturated depth), then the closing profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s16
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s16/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s16/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 150 707000 > gpurun_out/r04_s16/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s16/fuzz.log
for rep in 1 2; do printf "clipped blur off n=64: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'; done
bash tools/sessions/r04_profiles.sh r04a
