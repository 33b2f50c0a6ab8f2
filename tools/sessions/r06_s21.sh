#!/bin/bash
# round-6 session 21: closing runs on the FINAL binary: 360 s of fuzz over every technique, 150 s polylines-only, 100 s forward fills, 90 s gpu_warp,
# 60 s per dialect setting; the sharp config again (kernel trace + PMC passes, after the bounded second tier); bench lines of the metric, cfg 4, sharp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s21; mkdir -p $O
timeout 500 python tools/extended_fuzz.py 360 2121 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 300 python tools/extended_fuzz.py 150 2122 > $O/fuzz_poly.log 2>&1; echo "fuzz poly rc=$?"; tail -1 $O/fuzz_poly.log
CS_FUZZ_FILLS=none,naive,naive_interpolating,inverse timeout 300 python tools/extended_fuzz.py 100 2123 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
CS_FUZZ_FILLS=gpu_warp timeout 300 python tools/extended_fuzz.py 90 2124 > $O/fuzz_gw.log 2>&1; echo "fuzz gw rc=$?"; tail -1 $O/fuzz_gw.log
for d in f64-disparity int64-sum D64; do CS_FUZZ_DIALECT=$d timeout 200 python tools/extended_fuzz.py 60 2125 > $O/fuzz_$d.log 2>&1; echo "fuzz $d rc=$?"; tail -1 $O/fuzz_$d.log; done
bash tools/gpu_profile_cfg.sh r06b sharp > /dev/null 2>&1; bash tools/gpu_profile.sh r06b_sharp_pmc --config sharp > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
head -6 gpurun_out/r06b_sharp/kernel_trace.txt | cut -c1-150
for c in metric cfg4 sharp; do timeout 900 python bench.py --config $c > $O/bench_$c.json 2>/dev/null; python3 -c "
import json; j=json.load(open('$O/bench_$c.json')); r=j['roofline']; print('$c', round(j['value'],1), 'fps', round(j['ms_per_step'],2), 'ms kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'own', round(r['frac_own_bytes'],3), 'pipeline', round(r['pipeline_frac'],3), r['binding_roof'], j.get('value_other_depths'))"; done
