"""The D64 dialect fuzz's mismatch of session 24 (node case 941490: hybrid_edge_plus, exponent 1.4): how many pixels, which techniques,
and does exponent 2.0 / the float64 chain alone show it?  (development aid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from oracle import node_oracle, oracle
from comfystereo_amd import engine
from test_gpu_fuzz import FILLS
seed = 941490 + 1   # (the fuzz prints seed - 1 after incrementing)
rng = np.random.default_rng(941490)
n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 70)), int(rng.choice([64, 200, 516, 1028, 1540]))
img = synth.image_f32(n, h, w, seed=seed)
depth = synth.depth_batch(str(rng.choice(["blobs", "stepped", "radial", "noisy_ramp", "clipped", "clipped", "random8"])), n, h, w, channels=3)
print("case", n, h, w)
ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}
for dialect in ("D64", "f64-disparity", "D32"):
    for fill in ("hybrid_edge_plus", "hybrid_edge", "polylines_soft"):
        if fill not in ui: continue
        for e in (1.4, 2.0):
            oracle.set_dialect(dialect); engine.DIALECT = dialect
            args = (8.0, 0.5, "left-right", -0.5, 0.0, e)
            want = node_oracle.generate(img, depth, *args, ui[fill], 20.0, 3.0, True, depth_blur_falloff=3.0, depth_blur_vert_smooth=4, batch_size=1)
            got = [t.cpu().numpy() for t in engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), *args, fill, 20.0, 3.0, True,
                                                           depth_blur_falloff=3.0, depth_blur_vert_smooth=4, batch_size=1)]
            bad = np.argwhere(got[0] != want[0])
            print(dialect, fill, "e", e, "mismatching values:", len(bad), bad[:4].tolist(), [ (float(got[0][tuple(b)]), float(want[0][tuple(b)])) for b in bad[:3]])
oracle.set_dialect("D32"); engine.DIALECT = "D32"
