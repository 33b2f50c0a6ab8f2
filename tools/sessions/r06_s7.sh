#!/bin/bash
# round-6 session 7: what did k_gpuwarp_flags write?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/sessions/r06_s7d_debug.py 2>&1 | grep -v amdgpu.ids
