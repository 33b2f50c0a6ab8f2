#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate) of the BASELINE configs 2-5 -> gpurun_out/r03c_cfgN_pmc/
for c in cfg2 cfg3 cfg4 cfg5; do bash tools/gpu_profile.sh r03c_${c}_pmc --config $c > /dev/null 2>&1; done
cd "$GRAFT_REPO_ROOT"
for c in cfg2 cfg3 cfg4 cfg5; do echo "== $c"; grep -h "FETCH_SIZE\|WRITE_SIZE" gpurun_out/r03c_${c}_pmc/pmc_fetch.txt gpurun_out/r03c_${c}_pmc/pmc_write.txt | cut -c1-60,95-150 | head -12; done
