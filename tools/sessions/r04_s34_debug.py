"""round-4 session 34 (development aid): statistics and mismatches of noise rows through the ranged row kernel."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synth
from oracle import node_oracle
from comfystereo_amd import engine, _native

h, w = 24, 1540
img = synth.image_f32(1, h, w, seed=12)
depth = synth.depth_batch("random8", 1, h, w, channels=3)
for fill, ui in (("polylines_soft", "Fill - Polylines Soft"), ("polylines_sharp", "Fill - Polylines Sharp")):
    want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.0, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
    for dbg in (0, 31, 30):
        _native.debug_set("dbg", dbg)
        p = engine.make_params(1, h, w, h, w, 3, fill, "left-right", 6.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        got = [t.cpu().numpy() for t in plan.run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
        st = plan.stats()
        bad = int((got[0] != want[0]).any(axis=-1).sum())
        print(fill, "dbg", dbg, "flagged", int(st[:, 11].sum()), "fallback", int(st[:, 10].sum()), "err", int(st[:, 9].sum()), "mismatching px", bad, "stats row", st[0].tolist())
        _native.debug_set("dbg", 0)
