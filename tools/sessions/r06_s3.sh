#!/bin/bash
# round-6 session 3: k_gpuwarp_q with column-strided global accesses (stage + sampling) and the quad layout for the column pass only:
# gpu_warp tests, fuzz, A/B against k_gpuwarp (CS_PT_VARIANT=27), VALU per wave of both
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r06_s3}; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "warp or cfg4 or lazy or 8k or dropin or chunks" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=gpu_warp timeout 200 python tools/extended_fuzz.py 60 6162 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
for i in 1 2; do
  for v in 0 27; do
    printf "1080p blur0 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 0 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
    printf "1080p blur1 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 1 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
    printf "4K    blur0 variant %2d: " $v; CS_PT_VARIANT=$v timeout 200 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill gpu_warp 2>&1 | tail -1 | sed 's/.*: //'
  done
done 2>&1 | tee $O/ab.txt
for v in 0 27; do
  rm -rf /tmp/pp
  CS_PT_VARIANT=$v timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --iters 2 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/pmc_v$v.txt --pmc > /dev/null && grep "k_gpuwarp" $O/pmc_v$v.txt | awk '{print $(NF-4), $NF}' | tr '\n' ' '; echo " (variant $v)"
done
