"""k_gpuwarp_q against k_gpuwarp: the row state (normalised depth, D, source map) through the dev build's dbg 60 (development aid)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native
np.set_printoptions(precision=5, linewidth=220, suppress=True)
dev = torch.device("cuda")
n, h, w = 1, 32, 64
img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).to(dev)
dnp = synth.depth_batch("blobs", n, h, w, channels=3)
depth = torch.from_numpy(dnp).to(dev)
p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
_native.debug_set("dbg", 60)
outs = {}
for v in (27, 0):
    _native.debug_set("pt_variant", v)
    outs[v] = [t.clone().cpu().numpy() for t in engine.Plan(p, dev).run(img, depth)]
_native.debug_set("pt_variant", 0); _native.debug_set("dbg", 0)
print("depth min/max", dnp.min(), dnp.max())
for y in (0, 5, 17):
    for c, name in enumerate(("ndn", "D", "sm")):
        print("row", y, name, "old", outs[27][0][0, y, 20:26, c], "new", outs[0][0][0, y, 20:26, c], "depth", dnp[0, y, 20:23, 0])
