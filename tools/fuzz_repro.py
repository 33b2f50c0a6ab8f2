"""Re-run ONE node case of tools/extended_fuzz.py by its seed (development aid): python tools/fuzz_repro.py SEED [REPEATS]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native
from oracle import node_oracle

FILLS = ["none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp", "inverse", "hybrid_edge"]
seed0 = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for key in sys.argv[3:]:
    _native.debug_set(key, 1)
seed = seed0
ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}
modes = ["left-right", "right-left", "top-bottom", "bottom-top", "red-cyan-anaglyph"]
rng = np.random.default_rng(seed); seed += 1
n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 70)), int(rng.choice([64, 200, 516, 1028, 1540]))
img = synth.image_f32(n, h, w, seed=seed)
depth = synth.depth_batch(str(rng.choice(["blobs", "stepped", "radial", "noisy_ramp", "clipped", "clipped"])), n, h, w, channels=3)
fill = str(rng.choice([f for f in FILLS if f in ui] + ["gpu_warp"]))
args = (float(rng.choice([2.0, 5.0, 8.0, 12.0])), float(rng.choice([0.0, 0.5, -1.0])), str(rng.choice(modes)),
        float(rng.choice([0.0, 0.3, -0.5])), float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([1.0, 2.0, 1.4])))
blur = (float(rng.choice([20.0, 5.0, 33.0])), float(rng.choice([20.0, 3.0])), bool(rng.random() < 0.8))
kw = dict(depth_blur_falloff=float(rng.choice([2.0, 1.0, 0.5, 3.0, 1.7])), depth_blur_vert_smooth=int(rng.integers(0, 8)),
          batch_size=int(rng.integers(1, 4)))
print("case", seed0, fill, (n, h, w), args, blur, kw)
want = node_oracle.generate(img, depth, *args, ui[fill], *blur, **kw)
for r in range(reps):
    got = [t.cpu().numpy() for t in engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), *args, fill, *blur, **kw)]
    res = []
    for k, (g, w_) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w_)
        res.append((k, len(bad), bad[:2].tolist()))
    print("run", r, res)
