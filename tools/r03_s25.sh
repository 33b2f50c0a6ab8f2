#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in host node host_prog; do python tools/host_probe_r03.py $m 2>&1 | grep -v "^$" | head -14; done
