#!/bin/bash
# round-4 session 15: (1) k_polypoint with 8 workgroups per CU (64 VGPRs: 4 spilled since the 8-byte point records) A/B;
# (2) replay kernel without the pointer indirection of its sliding windows; overflowed rows (depth noise) exported as one whole-row
# stretch, with a pool that holds every row: parity subset + fuzz, saturated / noise depth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s15
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_pp8.so" tools/abn.sh --n 32 --blur 0 --iters 20 2>&1 | tee gpurun_out/r04_s15/ab_pp8.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_pp8.so" tools/abn.sh --n 64 --blur 1 --iters 10 2>&1 | tee gpurun_out/r04_s15/ab_pp8_blur.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_pp8.so" tools/abn.sh --n 32 --blur 1 --iters 10 --kind blobs 2>&1 | tee gpurun_out/r04_s15/ab_pp8_blobs.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_pp8.so" tools/abn.sh --n 32 --blur 0 --iters 10 --fill polylines_sharp 2>&1 | tee gpurun_out/r04_s15/ab_pp8_sharp.txt
CS_LIB_PATH=$PWD/$C/libcs_pp8.so timeout 900 python -m pytest tests -x -q -m gpu -k "poly or metric or cfg2 or fuzz" > gpurun_out/r04_s15/tests_pp8.log 2>&1; echo "pp8 tests rc=$?"; tail -2 gpurun_out/r04_s15/tests_pp8.log
timeout 1500 python -m pytest tests -x -q -m gpu -k "poly or ties or saturated or replay or order or fuzz or stress or metric or cfg2 or wide or 8k" > gpurun_out/r04_s15/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_s15/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 200 606000 > gpurun_out/r04_s15/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_s15/fuzz.log
for rep in 1 2; do printf "clipped blur off n=64: "; timeout 300 python tools/quick_bench.py --kind clipped --blur 0 --n 64 --iters 3 2>&1 | tail -1 | sed 's/.*: //'; done
printf "random8 blur off n=4, pool for every row: "; CS_DBG=14 timeout 600 python tools/quick_bench.py --kind random8 --blur 0 --n 4 --iters 1 --tie-pool-mb 1000 2>&1 | tail -4
printf "random8 blur off n=4, default pool: "; timeout 600 python tools/quick_bench.py --kind random8 --blur 0 --n 4 --iters 1 2>&1 | tail -1
timeout 900 python bench.py --depth random8 --no-blur --no-cpu-baseline --frames 16 --steps 2 --warmup 1 2>/dev/null | tail -1 | tee gpurun_out/r04_s15/bench_random8_blur_off.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench random8 blur off (16 frames)', round(d['value'],2), 'fps', d['diagnostics'])"
