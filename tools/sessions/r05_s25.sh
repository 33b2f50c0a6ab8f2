#!/bin/bash
# round-5 session 25: triage of the D64 dialect fuzz's one mismatch (session 24); then the D64 fuzz again with the exact exponents
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s25; mkdir -p $O
timeout 600 python tools/sessions/r05_s25_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/triage.txt
CS_FUZZ_DIALECT=D64 timeout 300 python tools/extended_fuzz.py 150 939393 > $O/fuzz_D64.log 2>&1; echo "fuzz D64 rc=$?"; tail -1 $O/fuzz_D64.log
