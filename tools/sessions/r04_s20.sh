#!/bin/bash
# round-4 session 20: gate after the k_gpuwarp node-layout instantiations: every -m gpu test, smoke, cfg4 / default bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s20; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_cfg4.json')); print('cfg4', round(j['value'],1), 'fps frac', round(j['roofline']['frac'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3))"
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); print('default', round(j['value'],1), 'fps frac', round(j['roofline']['frac'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3))"
