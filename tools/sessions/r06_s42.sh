#!/bin/bash
# round-6 session 42: the default bench line on the round's final binary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_s42
timeout 170 python bench.py > gpurun_out/r06_s42/bench_default.json 2>/dev/null; echo "bench rc=$?"
python3 -c "
import json; j=json.load(open('gpurun_out/r06_s42/bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'd64', j.get('value_dialect_d64'))"
