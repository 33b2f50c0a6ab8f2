#!/bin/bash
# round-5 session 20: the tile hints of k_polypoint (which tiles of a flagged row-eye raised the hazard): tie / polylines tests, then the
# distribution on saturated, stepped, blobs and noise depth (4K, 8 frames)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s20; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or parity" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 600 python tools/sessions/r05_s20_hints.py 2>&1 | grep -v amdgpu.ids | tee $O/hints.txt
