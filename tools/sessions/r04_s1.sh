#!/bin/bash
# round-4 session 1: (1) what one more instruction costs in k_polypoint: the production kernel padded with +100 / +200 scalar
# adds, +100 / +200 vector adds, +100 conversions per wave (libcs_pad*.so: -DPP_PAD_S/V/C), alternated with the unpadded build;
# (2) the default bench line of this box (baseline of the round)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_s1
LIBS="comfystereo_amd/libcomfystereo_hip.so comfystereo_amd/libcs_padS100.so comfystereo_amd/libcs_padV100.so comfystereo_amd/libcs_padC100.so comfystereo_amd/libcs_padS200.so comfystereo_amd/libcs_padV200.so" \
  tools/abn.sh --n 32 --blur 0 --iters 20 2>&1 | tee gpurun_out/r04_s1/model.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_s1/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_s1/bench.json')); print('headline', round(d['value'],1), 'fps; kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
