#!/bin/bash
# round-4 session 26: gate + closing profiles of the final kernels (tag r04c): every -m gpu test, smoke, then tools/sessions/r04_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s26
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s26/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s26/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/sessions/r04_profiles.sh r04c
