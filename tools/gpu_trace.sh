#!/bin/bash
# kernel trace of an arbitrary python command on the GPU box -> gpurun_out/<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats -d /tmp/tr -o p -- python3 "$@" > /dev/null 2>&1
python3 tools/prof_summary.py $(find /tmp/tr -name '*.db' | head -1) gpurun_out/$TAG.txt | head -12
