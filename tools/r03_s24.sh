#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, synth
from comfystereo_amd.GenerateStereo import StereoImageNode
n,h,w=32,2160,3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3))
node = StereoImageNode()
args = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True, 2.0, 6, 12)
r=[]
out=None
for i in range(8):
    t0=time.perf_counter(); out = node.generate(img, dep, *args); torch.cuda.synchronize(); r.append(round(n/(time.perf_counter()-t0),1))
print("node.generate, results kept (first call = cold):", r)
PY
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 2>&1 | tail -3
