"""Census of the polylines geometry on the bench workload (development aid behind the k_polytile design):
per output pixel, how many polyline points fall into it, how many pixels lie under a reversed (folded) segment, how long
the disocclusion bridges are, and how the classes distribute over 512-pixel tiles."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from oracle import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--kind", default="stepped")
ap.add_argument("--div", type=float, default=8.0)
ap.add_argument("--blur", type=int, default=1)
ap.add_argument("--rows", type=int, default=2160)
ap.add_argument("--tile", type=int, default=512)
a = ap.parse_args()

d = synth.DEPTHS[a.kind](a.h, a.w) * np.float32(255)
if a.blur:
    dl, dr = oracle.blur(d, 20, 20, 2.0, 6)
else:
    dl = dr = d
w = a.w
for eye, (dep, sign) in enumerate(((dl, 1.0), (dr, -1.0))):
    dep = dep[: a.rows]
    nd = (dep - dep.min()) / (dep.max() - dep.min()) - np.float32(0.5)
    div32 = np.float32(sign * a.div / 100.0 * w)
    cd = np.sign(nd).astype(np.float32) * (np.abs(nd) ** 2).astype(np.float32) * div32
    x = (np.arange(w, dtype=np.float32) + np.float32(0.5))[None, :] + cd
    fx = np.floor(x).astype(np.int64)
    rows = x.shape[0]
    npts = np.zeros((rows, w), np.int32)
    for r in range(rows):
        ok = (fx[r] >= 0) & (fx[r] < w)
        np.add.at(npts[r], fx[r][ok], 1)
    rev = x[:, 1:] <= x[:, :-1]
    # pixels under a reversed segment (closed extent)
    dirty = np.zeros((rows, w + 1), np.int32)
    rr, cc = np.nonzero(rev)
    lo = np.clip(np.floor(x[rr, cc + 1]).astype(np.int64), 0, w - 1)
    hi = np.clip(np.floor(x[rr, cc]).astype(np.int64), 0, w - 1)
    np.add.at(dirty, (rr, lo), 1)
    np.add.at(dirty, (rr, hi + 1), -1)
    dirty = np.cumsum(dirty, axis=1)[:, :w] > 0
    tot = rows * w
    print(f"eye {eye}: points per pixel histogram (fraction of pixels):",
          {k: round(float((npts == k).sum()) / tot, 4) for k in range(0, 6)}, ">=6:", round(float((npts >= 6).sum()) / tot, 5))
    print(f"  reversed segments: {rev.sum() / tot:.4f} of segments; dirty pixels: {dirty.sum() / tot:.4f}")
    clean = ~dirty
    print("  clean pixels by points:", {k: round(float(((npts == k) & clean).sum()) / tot, 4) for k in range(0, 5)})
    # bridges: runs of 0-point clean pixels
    z = (npts == 0) & clean
    runs = []
    for r in range(0, rows, max(rows // 200, 1)):
        zz = np.diff(np.concatenate(([0], z[r].astype(np.int8), [0])))
        s, e = np.nonzero(zz == 1)[0], np.nonzero(zz == -1)[0]
        runs.extend((e - s).tolist())
    runs = np.array(runs) if runs else np.zeros(1)
    print("  bridge run lengths (sampled rows): mean %.2f, p50 %d, p90 %d, p99 %d, max %d" %
          (runs.mean(), np.percentile(runs, 50), np.percentile(runs, 90), np.percentile(runs, 99), runs.max()))
    T = a.tile
    nt = (w + T - 1) // T
    dt = np.array([[dirty[r, t * T:(t + 1) * T].sum() for t in range(nt)] for r in range(0, rows, max(rows // 300, 1))])
    print(f"  tiles with a dirty pixel: {float((dt > 0).mean()):.3f}; dirty pixels per such tile: mean {dt[dt > 0].mean():.1f}, "
          f"p90 {np.percentile(dt[dt > 0], 90):.0f}, max {dt.max()}")
    print(f"  max |shift| {np.abs(cd).max():.1f} px")
