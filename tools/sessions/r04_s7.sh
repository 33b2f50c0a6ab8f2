#!/bin/bash
# round-4 session 7: (1) the tie path: does the replay pool run out (CS_PT_VARIANT=31: four times the budget; CS_DBG=14 counters)?
# whole-workgroup window copies; against the round-3 tree; (2) k_gpuwarp with XCD-contiguous rows (libcs_gwx, rebuilt last)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s7
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "ties or saturated or replay or order" > gpurun_out/r04_s7/tests.log 2>&1; echo "tie tests rc=$?"; tail -3 gpurun_out/r04_s7/tests.log
for v in 0 31; do
  CS_PT_VARIANT=$v CS_DBG=14 timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 16 --iters 3 2>&1 | tail -4 | sed "s/^/PT_VARIANT=$v: /"
done
(cd tools/_r3 && timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-3 tree: clipped blur off', round(d['value'],1), 'fps')")
timeout 600 python bench.py --depth clipped --no-blur --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('current:      clipped blur off', round(d['value'],1), 'fps')"
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s7/ab_1080p.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_gwx.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s7/ab_4k.txt
for L in $C/libcomfystereo_hip.so $C/libcs_gwx.so; do for grp in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pp; CS_LIB_PATH=$PWD/$L timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 3 > /tmp/run.log 2>&1; db=$(find /tmp/pp -name "*.db" | head -1); printf "%-28s %s " "$(basename $L)" $grp; [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "k_gpuwarp" | awk '{print $(NF-4), $(NF-2), $(NF-1), $NF}'; done; done 2>&1 | tee gpurun_out/r04_s7/pmc.txt
CS_LIB_PATH=$PWD/$C/libcs_gwx.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -x -q -m gpu -k "warp or cfg4" > gpurun_out/r04_s7/tests_gwx.log 2>&1; echo "gwx warp tests rc=$?"; tail -3 gpurun_out/r04_s7/tests_gwx.log
