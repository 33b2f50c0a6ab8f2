#!/bin/bash
# round-6 session 17: gate for the bounded second tier (every -m gpu test, smoke, fuzz), then the host path A/B: this round's
# host_pipeline.py against round 5's (a temporary copy: git show a858e00:comfystereo_amd/host_pipeline.py > comfystereo_amd/host_pipeline_r05.py, removed
# afterwards), alternating, three times each (32 x 4K polylines_soft, pinned cap 64 GB, 5 iterations): 185-192 against 180-186 frames/s
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s17; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 300 python tools/extended_fuzz.py 150 1717 > $O/fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 120 1718 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for i in 1 2 3; do
echo "--- round 6"; timeout 600 python tools/node_host_bench.py --n 32 --iters 5 --prewarm 0 --pin-cap-gb 64 2>&1 | grep "host tensors\|out=" | cut -c1-120
echo "--- round 5 host_pipeline.py"; timeout 600 python -c "
import sys, runpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import comfystereo_amd.host_pipeline_r05 as old
sys.modules['comfystereo_amd.host_pipeline'] = old
import comfystereo_amd; comfystereo_amd.host_pipeline = old
sys.argv = ['node_host_bench.py', '--n', '32', '--iters', '5', '--prewarm', '0', '--pin-cap-gb', '64']
runpy.run_path('tools/node_host_bench.py', run_name='__main__')" 2>&1 | grep "host tensors\|out=" | cut -c1-120
done 2>&1 | tee $O/host_ab.txt
