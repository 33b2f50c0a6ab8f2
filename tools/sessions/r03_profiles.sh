#!/bin/bash
# round 3 profiles: default bench (kernel trace + PMC passes), the other BASELINE configs (kernel trace + bench line), host bench
TAG=${1:-r03b}
bash tools/gpu_profile.sh ${TAG}_blur_on > /dev/null 2>&1
for c in cfg2 cfg3 cfg4 cfg5; do bash tools/gpu_profile_cfg.sh $TAG $c > /dev/null 2>&1; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2>/dev/null
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 > gpurun_out/${TAG}_host_4k.txt 2>&1
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 --h 1080 --w 1920 > gpurun_out/${TAG}_host_1080p.txt 2>&1
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 --fill "GPU Warp (Fast)" --h 1080 --w 1920 > gpurun_out/${TAG}_host_1080p_gpuwarp.txt 2>&1
head -12 gpurun_out/${TAG}_blur_on/kernel_trace.txt | cut -c1-140
for c in cfg2 cfg3 cfg4 cfg5; do python3 -c "
import json,sys
j=json.load(open('gpurun_out/${TAG}_$c/bench.json')); print('$c', round(j['value'],1), 'fps frac', round(j['roofline']['frac'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3))"; done
tail -3 gpurun_out/${TAG}_host_4k.txt; tail -2 gpurun_out/${TAG}_host_1080p_gpuwarp.txt
