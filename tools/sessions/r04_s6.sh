#!/bin/bash
# round-4 session 6: (1) where did the saturated-depth tie path lose 36 ms per 64 frames?  the round-3 tree (git worktree tools/_r3)
# against the current one on the same box + the replay kernel's own counters (CS_DBG=14); (2) k_gpuwarp with XCD-contiguous rows
# (libcs_gwx, rebuilt) A/B + FETCH_SIZE; (3) the whole GPU suite (scipy depth blur, forward_warp_gpu's keyword parameters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s6
C=comfystereo_amd
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_s6/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04_s6/tests.log
for rep in 1 2; do
  (cd tools/_r3 && timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-3 tree: clipped blur off', round(d['value'],1), 'fps', d['diagnostics'])")
  timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current:      clipped blur off', round(d['value'],1), 'fps', d['diagnostics'])"
done
CS_DBG=14 timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 8 --iters 3 2>&1 | tail -4
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s6/ab_1080p.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s6/ab_4k.txt
for L in $C/libcomfystereo_hip.so $C/libcs_gwx.so; do for grp in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pp; CS_LIB_PATH=$PWD/$L timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 3 > /tmp/run.log 2>&1; db=$(find /tmp/pp -name "*.db" | head -1); printf "%-28s %s " "$(basename $L)" $grp; [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "k_gpuwarp" | awk '{print $(NF-4), $(NF-2), $(NF-1), $NF}'; done; done 2>&1 | tee gpurun_out/r04_s6/pmc.txt
