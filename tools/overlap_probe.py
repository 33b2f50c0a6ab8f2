"""Development probe: does running two half-batches on two HIP streams (pre-pass of one overlapping the warp of the other)
beat one full batch on one stream?   python tools/overlap_probe.py [--blur 1]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--blur", type=int, default=1)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
h, w = 2160, 3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).to(dev).expand(a.n, -1, -1, -1).contiguous()
depth = torch.from_numpy(synth.depth_batch("stepped", a.n, h, w, channels=3)).to(dev)


def params(n):
    return engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, bool(a.blur), 20.0, 20.0, 2.0, 6, 12)


full = engine.Plan(params(a.n), dev)
full.run(img, depth); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    full.run(img, depth)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / a.iters
k = a.n // a.parts
parts = [engine.Plan(params(k), dev) for _ in range(a.parts)]
streams = [torch.cuda.Stream(dev) for _ in range(a.parts)]
for i, (pl, s) in enumerate(zip(parts, streams)):
    with torch.cuda.stream(s):
        pl.run(img[i * k:(i + 1) * k], depth[i * k:(i + 1) * k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    for i, (pl, s) in enumerate(zip(parts, streams)):
        with torch.cuda.stream(s):
            pl.run(img[i * k:(i + 1) * k], depth[i * k:(i + 1) * k])
torch.cuda.synchronize()
t_par = (time.perf_counter() - t0) / a.iters
print(f"n={a.n} blur={a.blur}: one stream {t_full*1e3:.2f} ms ({a.n/t_full:.0f} fps); {a.parts} streams x {k} frames {t_par*1e3:.2f} ms ({a.n/t_par:.0f} fps)")
