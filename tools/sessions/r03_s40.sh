#!/bin/bash
# last profiles of round 3 (tag r03g): cfg3 and cfg4 after the lazy tiles (kernel trace + bench line + PMC passes)
for c in cfg3 cfg4; do bash tools/gpu_profile_cfg.sh r03g $c > /dev/null 2>&1; bash tools/gpu_profile.sh r03g_${c}_pmc --config $c > /dev/null 2>&1; done
cd "$GRAFT_REPO_ROOT"
for c in cfg3 cfg4; do python3 -c "
import json
j=json.load(open('gpurun_out/r03g_$c/bench.json')); print('$c', round(j['value'],1), 'fps frac', round(j['roofline']['frac'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3), 'pipeline', round(j['roofline']['pipeline_frac'],3))"; head -9 gpurun_out/r03g_$c/kernel_trace.txt | cut -c1-130; done
