#!/bin/bash
# round-5 session 28: why does k_polypoint flag rows of saturated depth WITH the blur on (no stretch comes of them)?  hazard reason bits
# (dev build: PP_HAZARD codes OR-ed into stats word 12) and the tile-hint distribution, blur on / off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s28b; mkdir -p $O
CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_dev.so timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/hazard_reasons.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch, synth
from comfystereo_amd import engine, _native
_native.debug_set('dbg', 15)
n, h, w = 8, 2160, 3840
img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).cuda()
for kind in ("clipped", "blobs"):
    depth = torch.from_numpy(synth.depth_batch(kind, n, h, w, channels=3)).cuda()
    for blur in (False, True):
        p = engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0, 20.0, 2.0, 6, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        plan.run(img, depth); torch.cuda.synchronize()
        st = plan.stats()
        bits = 0
        for v in st[:, 12]: bits |= int(v)
        print(kind, "blur", blur, "hazard reason bits", hex(bits), "per frame", [hex(int(v)) for v in st[:, 12]], "events capacity / tie / other", int(st[:, 13].sum()), int(st[:, 14].sum()), int(st[:, 15].sum()), "rows redone", int(st[:, 11].sum()), "rows with replayed pixels", int(st[:, 10].sum()))
PY
