"""round-4 session 23: where do 7680-wide polylines_sharp rows with ties differ from the oracle? (development aid)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synth
from oracle import node_oracle
from comfystereo_amd import engine, _native

def run(kind, w, h=5, div=4.0, switches=()):
    for k, v in switches: _native.debug_set(k, v)
    img = synth.image_f32(1, h, w, seed=21)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    want = node_oracle.generate(img, depth, div, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Sharp", 20.0, 20.0, False, batch_size=12)
    p = engine.make_params(1, h, w, h, w, 3, "polylines_sharp", "left-right", div, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    got = [t.cpu().numpy() for t in plan.run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
    st = plan.stats()
    bad = np.argwhere((got[0] != want[0]).any(axis=-1))
    print(kind, w, switches, "flagged", int(st[:, 11].sum()), "seq", int(st[:, 10].sum()), "err", int(st[:, 9].sum()), "mismatching pixels", len(bad))
    if len(bad):
        rows = sorted(set(int(b[1]) for b in bad))
        for r in rows[:3]:
            cols = bad[bad[:, 1] == r][:, 2]
            print("  row", r, "cols", cols.min(), "..", cols.max(), "count", len(cols), "first", cols[:12].tolist())
    for k, v in switches: _native.debug_set(k, 0)

if __name__ == "__main__":
    kind, w = sys.argv[1], int(sys.argv[2])
    sw = tuple((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[3:])
    run(kind, w, switches=sw)
