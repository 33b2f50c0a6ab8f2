#!/bin/bash
# round-5 session 12: does any result depend on what the workspace / output buffers held before the call?  (poisoned allocator blocks)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s12; mkdir -p $O
for p in 0 255 85 1; do timeout 600 python tools/sessions/r05_s12_debug.py $p 2>&1 | grep -v amdgpu.ids | tee -a $O/poison.txt; done
