#!/bin/bash
# round-5 session 2: (1) parity of the round's host-side changes (library-owned tile counter, lazy second pipeline slot + pinned
# cap, bench.py --gather none) and of the hybrid_edge weight-table experiment (libcs_wtab: -DHYB_WTAB); (2) cfg3 A/B: weight table
# against the computed exp; (3) first-call frames/s of the node for three shapes, cold / opt-in warm-up (profiles/r05_host.txt);
# (4) profiles of their own for naive_interpolating and polylines_sharp (kernel trace + four PMC passes, 64 x 4K SBS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s2; mkdir -p $O
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_chunks.py tests/test_gpu_dropin.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests_host.log 2>&1; echo "host-side tests rc=$?"; tail -3 $O/tests_host.log
CS_LIB_PATH=$PWD/$C/libcs_wtab.so timeout 900 python -m pytest tests -x -q -m gpu -k "hybrid or cfg3" > $O/tests_wtab.log 2>&1; echo "wtab tests rc=$?"; tail -3 $O/tests_wtab.log
LIBS="$C/libcomfystereo_hip.so $C/libcs_wtab.so" tools/abn.sh --n 16 --fill hybrid_edge --blur 1 --iters 10 2>&1 | tee $O/ab_wtab.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_wtab.so" tools/abn.sh --n 16 --fill hybrid_edge --blur 1 --iters 10 --kind radial 2>&1 | tee -a $O/ab_wtab.txt
for L in comfystereo_hip cs_wtab; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/lib$L.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 16 --fill hybrid_edge --blur 1 --iters 10 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_hyb_$L.txt > /dev/null
  printf "%-20s " $L; grep k_hybrid_splat_tile $O/trace_hyb_$L.txt | awk '{print $(NF-3), $(NF-2), $(NF-1)}'
done 2>&1 | tee $O/kernel_times_wtab.txt
bash tools/host_first_call.sh 2>&1 | tee $O/host_first_call.txt
bash tools/gpu_profile.sh r05a_naive_interp --config naive_interp > /dev/null 2>&1
bash tools/gpu_profile.sh r05a_sharp --config sharp > /dev/null 2>&1
for c in naive_interp sharp; do
  timeout 300 python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05a_$c/bench.json
  head -8 gpurun_out/r05a_$c/kernel_trace.txt | cut -c1-150
  grep -E "k_fwdtile|k_polypoint" gpurun_out/r05a_$c/pmc_sq1.txt gpurun_out/r05a_$c/pmc_sq2.txt gpurun_out/r05a_$c/pmc_fetch.txt gpurun_out/r05a_$c/pmc_write.txt | awk '{print $(NF-2), $(NF-1), $NF}'
done
