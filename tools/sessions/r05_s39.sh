#!/bin/bash
# round-5 session 39: after the empty-list guard of the lane replay: the tie / polylines tests, a short polylines fuzz, smoke
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or lean or saturated or stretch or order" 2>&1 | tail -2
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 200 python tools/extended_fuzz.py 60 121212 2>&1 | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
