#!/bin/bash
# round-3 closing profiles (tag r03f; r03d: the state before the 4-slot geometries): default bench (kernel trace + PMC passes), BASELINE configs 2-5 (kernel trace + bench line),
# PMC passes of cfg3 (its kernels changed), host bench, tie bench lines
bash tools/sessions/r03_profiles.sh r03f
for c in cfg3; do bash tools/gpu_profile.sh r03f_${c}_pmc --config $c > /dev/null 2>&1; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03f_ties
for d in clipped random8; do
  timeout 600 python bench.py --depth $d --no-blur --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03f_ties/bench_${d}_blur_off.json 2>/dev/null
  timeout 600 python bench.py --depth $d --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03f_ties/bench_${d}_blur_on.json 2>/dev/null
done
for f in gpurun_out/r03f_ties/*.json; do python3 -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['value'],1), 'fps')"; done
grep -h "FETCH_SIZE\|WRITE_SIZE" gpurun_out/r03f_cfg3_pmc/pmc_fetch.txt gpurun_out/r03f_cfg3_pmc/pmc_write.txt | cut -c1-60,95-150 | head -8
