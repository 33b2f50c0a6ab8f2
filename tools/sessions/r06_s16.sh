#!/bin/bash
# round-6 session 16: the second tier with a bounded grid (first quarter of the rows; k_polypoint_carry hands the rest on): polylines /
# scene8 / tie tests + fuzz; sharp with first-tier lists 6 / 9 (default) against 5 / 7 (libcs_ppk57.so), tier on / off, scene8 / stepped /
# blobs / clipped, 16 and 64 frames; then the host path: this round's host_pipeline.py against round 5's (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s16; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -k "polylines or scene8 or tie or replay or lean or saturated or stretch or order or anaglyph or fullsize" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 100 1616 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
for i in 1 2; do for L in comfystereo_hip cs_ppk57; do for v in 0 49; do for k in scene8 stepped blobs clipped; do for b in 0 1; do
  printf "%-16s tier2 %-3s %-8s blur %s: " $L $([ $v = 0 ] && echo on || echo off) $k $b
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill polylines_sharp --kind $k 2>&1 | grep "tile-redo\|fps" | sed 's/.*tile-redo rows: \[\([0-9]*\),.*/rows(frame 0) \1/; s/.*ms\/batch, //' | tr '\n' ' '; echo
done; done; done; done; done 2>&1 | tee $O/ab.txt
for L in comfystereo_hip cs_ppk57; do for v in 0 49; do
  printf "%-16s tier2 %-3s stepped 64 frames: " $L $([ $v = 0 ] && echo on || echo off)
  CS_LIB_PATH=$PWD/comfystereo_amd/lib$L.so CS_PT_VARIANT=$v timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 4 --fill polylines_sharp --kind stepped 2>&1 | tail -1 | sed 's/.*ms\/batch, //'
done; done 2>&1 | tee -a $O/ab.txt
echo "--- host path, round 6"; timeout 600 python tools/node_host_bench.py --n 32 --iters 3 --prewarm 0 --pin-cap-gb 64 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/host_r06.txt
echo "--- host path, round 5's host_pipeline.py"; timeout 600 python -c "
import sys, runpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import comfystereo_amd.host_pipeline_r05 as old
sys.modules['comfystereo_amd.host_pipeline'] = old
import comfystereo_amd; comfystereo_amd.host_pipeline = old
sys.argv = ['node_host_bench.py', '--n', '32', '--iters', '3', '--prewarm', '0', '--pin-cap-gb', '64']
runpy.run_path('tools/node_host_bench.py', run_name='__main__')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/host_r05.txt
echo "--- host path, round 6 again"; timeout 600 python tools/node_host_bench.py --n 32 --iters 3 --prewarm 0 --pin-cap-gb 64 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/host_r06.txt
