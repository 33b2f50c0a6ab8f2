"""k_gpuwarp_q against k_gpuwarp (pt_variant 27) on small frames: where do they differ?  (development aid, round 6)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native

dev = torch.device("cuda")
for (n, h, w, blur) in ((2, 32, 64, False), (3, 64, 96, False), (1, 46, 1540, True), (2, 28, 1540, True), (3, 8, 516, True)):
    img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).to(dev)
    depth = torch.from_numpy(synth.depth_batch("blobs", n, h, w, channels=3)).to(dev)
    p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0, 20.0, 2.0, 6, 12)
    outs = {}
    for v in (27, 0):
        _native.debug_set("pt_variant", v)
        plan = engine.Plan(p, dev)
        outs[v] = [t.clone().cpu().numpy() for t in plan.run(img, depth)]
    _native.debug_set("pt_variant", 0)
    for k, name in enumerate(("stereo", "dl", "dr", "mask")):
        a, b = outs[27][k], outs[0][k]
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        d = d.reshape(d.shape[0], d.shape[1], -1).max(axis=2)   # per frame, row
        bad = np.argwhere(d > 1e-6)
        print((n, h, w, blur), name, "max", d.max(), "bad rows", len(bad), bad[:12].tolist())
    # which row of the old kernel's stereoscope does each new row resemble?  (left eye half, right eye half)
    a, b = outs[27][0], outs[0][0]
    if a.shape[1] == h and a.shape[2] == 2 * w:
        for f in range(min(n, 1)):
            for y in (0, 1, 2, h // 2, h - 1):
                for half, sl in (("L", slice(0, w)), ("R", slice(w, 2 * w))):
                    d = np.abs(a[f, :, sl].astype(np.float64) - b[f, y:y + 1, sl].astype(np.float64)).reshape(h, -1)
                    frac_equal = (d <= 1e-6).mean(axis=1)
                    print("   frame", f, "new row", y, half, "best old row", int(frac_equal.argmax()), "equal fraction", round(float(frac_equal.max()), 3),
                          "| same row equal fraction", round(float(frac_equal[y]), 3))
