"""k_gpuwarp_q against k_gpuwarp (pt_variant 27), one small shape: raw values of a bad row (development aid, round 6)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native
np.set_printoptions(precision=4, linewidth=200, suppress=True)
dev = torch.device("cuda")
n, h, w = 1, 32, 64
img_np = synth.image_f32(n, h, w, seed=5)
# image whose pixel value encodes its row: channel 0 = row / 100, channel 1 = column / 100
for y in range(h):
    img_np[0, y, :, 0] = y / 100.0
    img_np[0, y, :, 1] = np.arange(w) / 100.0
img = torch.from_numpy(img_np).to(dev)
depth = torch.from_numpy(synth.depth_batch("blobs", n, h, w, channels=3)).to(dev)
p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
outs = {}
for v in (27, 0):
    _native.debug_set("pt_variant", v)
    outs[v] = [t.clone().cpu().numpy() for t in engine.Plan(p, dev).run(img, depth)]
_native.debug_set("pt_variant", 0)
for y in (0, 1, 2, 5, 17, 31):
    print("row", y, "old ch0 (row code) L:", outs[27][0][0, y, 20:26, 0], "new:", outs[0][0][0, y, 20:26, 0])
    print("        old ch1 (col code) L:", outs[27][0][0, y, 20:26, 1], "new:", outs[0][0][0, y, 20:26, 1])
    print("        old R ch0:", outs[27][0][0, y, w + 20:w + 26, 0], "new:", outs[0][0][0, y, w + 20:w + 26, 0])
