#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, synth
from comfystereo_amd import host_pipeline
from comfystereo_amd.GenerateStereo import StereoImageNode
n,h,w=32,2160,3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3))
hargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "polylines_soft", 20.0, 20.0, True, 2.0, 6, 12)
nargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True, 2.0, 6, 12)
node = StereoImageNode()
def t(f, keep):
    held = None
    r = []
    for _ in range(4):
        t0=time.perf_counter(); out = f(); torch.cuda.synchronize(); r.append(round(n/(time.perf_counter()-t0),1))
        if keep: held = out
        else: del out
    return r
host_pipeline.generate_host(img, dep, *hargs)
print("generate_host, results released each time:", t(lambda: host_pipeline.generate_host(img, dep, *hargs), False))
print("generate_host, previous results kept alive:", t(lambda: host_pipeline.generate_host(img, dep, *hargs), True))
print("node.generate, released:", t(lambda: node.generate(img, dep, *nargs), False))
print("node.generate, kept:", t(lambda: node.generate(img, dep, *nargs), True))
print("generate_host with a progress callback, released:", t(lambda: host_pipeline.generate_host(img, dep, *hargs, progress=lambda k: None), False))
import subprocess; print(subprocess.run("free -g | head -2", shell=True, capture_output=True, text=True).stdout)
PY
