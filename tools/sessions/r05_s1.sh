#!/bin/bash
# round-5 session 1: the memory skeleton of k_polypoint.  (1) tools/ubench/store_patterns: the tile geometry and
# byte counts of the kernel with no arithmetic -- today's lane-per-source 12-byte stores against full-line dwordx4 stores, plain and
# nontemporal, write-only / read-only; (2) the production kernel cut down to its loads, LDS records and stores (libcs_skel1: staging
# arithmetic kept; skel2: none; skel3: loads only; skel4: no depth-map stores) and with nontemporal output stores (libcs_nt),
# alternated with the production build; kernel times from rocprofv3; (3) write-request granularity (TCC_EA0_WRREQ / _64B);
# (4) the default bench line of this box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s1; mkdir -p $O
C=comfystereo_amd
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "WRREQ|WRITE_SIZE|FETCH_SIZE|TCC_EA0" $O/counters.txt | head -60 > $O/counters_tcc.txt
( cd tools/ubench && timeout 300 ./store_patterns 16 ) 2>&1 | tee $O/store_patterns.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
LIBS="$C/libcomfystereo_hip.so $C/libcs_skel1.so $C/libcs_skel2.so $C/libcs_skel3.so $C/libcs_skel4.so $C/libcs_nt.so" \
  tools/abn.sh --n 32 --blur 0 --iters 20 2>&1 | tee $O/ab_blur0.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_nt.so" tools/abn.sh --n 32 --blur 1 --iters 20 2>&1 | tee $O/ab_blur1.txt
for L in comfystereo_hip cs_skel1 cs_skel2 cs_skel3 cs_skel4 cs_nt; do
  rm -rf /tmp/pp
  CS_LIB_PATH=$PWD/$C/lib$L.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --blur 0 --iters 10 > /tmp/run.log 2>&1
  db=$(find /tmp/pp -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db $O/trace_$L.txt > /dev/null
  printf "%-20s " $L; grep k_polypoint $O/trace_$L.txt | awk '{print $(NF-3), $(NF-2), $(NF-1)}'
done 2>&1 | tee $O/kernel_times.txt
for L in comfystereo_hip cs_skel2 cs_nt; do
  for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE" "FETCH_SIZE"; do
    rm -rf /tmp/pp
    CS_LIB_PATH=$PWD/$C/lib$L.so timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o p -- python3 tools/quick_bench.py --n 32 --blur 0 --iters 3 > /tmp/run.log 2>&1
    db=$(find /tmp/pp -name '*.db' | head -1)
    [ -n "$db" ] && python3 tools/prof_summary.py $db /tmp/g.txt --pmc | grep -E "k_polypoint" | sed "s/^/$L /"
  done
done 2>&1 | tee $O/pmc_wrreq.txt
