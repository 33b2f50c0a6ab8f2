#!/bin/bash
# round-6 session 1: gate after the ADVICE r5 fixes (single-CAS pool refund + tiny-pool test, cs_max_width_params / ABI 4, pinned
# accounting of the host pipeline): every -m gpu test, smoke, the default bench line, cfg 4 baseline (k_gpuwarp: the round's first target)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s1; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; j=json.load(open('$O/bench_default.json')); r=j['roofline']; print(round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms; blur off', round(j['value_blur_off'],1), j['value_other_depths'], 'frac', round(r['frac'],3), round(r['frac_node_bytes'],3), round(r['pipeline_frac'],3), 'kernel_ms', round(r['kernel_ms'],3))"
timeout 600 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_cfg4.json')); r=j['roofline']; print('cfg4', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3))"
timeout 300 python tools/node_host_bench.py --n 24 --iters 3 --prewarm 0 --fill "GPU Warp (Fast)" 2>&1 | grep -v amdgpu.ids | tail -4
