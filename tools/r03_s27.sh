#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/host_probe_r03.py host keep 2>&1 | grep -v "^$" | head -16
