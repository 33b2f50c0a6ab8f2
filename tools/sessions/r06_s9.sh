#!/bin/bash
# round-6 session 9: scene8 depth (8-bit gradient + flat ellipses): the new GPU tests, the by-depth table with a scene8 column
# (16 x 4K SBS, blur off / on, every technique: stepped vs scene8), the bench line with value_other_depths.scene8; the host path of
# gpu_warp after the threaded copy of its float32 output (default pinned cap: pageable results)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scene8.py tests/test_gpu_dropin.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for k in stepped scene8; do for b in 0 1; do for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-8s blur %s %-22s " $k $b $f; timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill $f --kind $k 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee $O/table_by_depth.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; blur off', round(j['value_blur_off'],1), j['value_other_depths'], 'frac', round(r['frac'],3), 'own', round(r['frac_own_bytes'],3), 'binding', r['binding_roof'], 'valu', r['valu'] and round(r['valu']['frac_of_issue_floor'],3))"
timeout 300 python tools/node_host_bench.py --n 24 --iters 3 --prewarm 0 --fill "GPU Warp (Fast)" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/host_gpuwarp.txt
