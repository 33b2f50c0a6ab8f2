"""Development aid: the point-owner polylines kernel (cs_polypoint.hip) against the CPU oracle and against the first
generation (cs_polytile.hip, cs_debug_set(PT_VARIANT, 9)) on a spread of inputs; mismatches are listed with the geometry
of their pixel (points in the pixel, reversed segments over it) so that a failing path can be named.  GPU only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import _native, engine
from oracle import node_oracle, oracle


def geometry(depth, div, sep, e, conv, w):
    nd = (depth - depth.min()) / (depth.max() - depth.min()) - np.float32(conv) if depth.max() > depth.min() else np.zeros_like(depth) - np.float32(conv)
    div32, sep32 = np.float32(div / 100.0 * w), np.float32(sep / 100.0 * w)
    pw = np.array([[oracle.lib().oracle_powf(float(abs(v)), float(np.float32(e))) for v in r] for r in nd], dtype=np.float32) if e not in (1.0, 2.0) else (np.abs(nd) if e == 1.0 else (np.abs(nd) * np.abs(nd)).astype(np.float32))
    cd = (np.sign(nd) + (nd == 0)).astype(np.float32) * pw * div32
    x = ((np.arange(w, dtype=np.float32) + np.float32(0.5))[None, :] + cd) + sep32
    return x


def describe(x_row, col):
    fx = np.floor(x_row)
    pts = np.nonzero(fx == col)[0]
    rev = [(j, float(x_row[j]), float(x_row[j + 1])) for j in range(len(x_row) - 1)
           if x_row[j + 1] <= x_row[j] and np.floor(x_row[j + 1]) <= col <= np.floor(x_row[j])]
    return f"points {pts.tolist()} x={[float(x_row[p]) for p in pts[:4]]} reversed-over {rev[:3]}"


bad_total = 0
cases = []
for kind in ("stepped", "radial", "blobs", "noisy_ramp", "random8"):
    for (h, w, div, sep, e, conv) in ((24, 1500, 7.0, 0.3, 2.0, 0.5), (16, 700, -6.0, -0.4, 1.0, 0.3), (12, 2600, 8.0, 0.0, 2.0, 0.5),
                                      (10, 333, 5.0, 1.1, 1.3, 0.7), (8, 96, 9.0, 0.0, 2.0, 0.0)):
        cases.append((kind, h, w, div, sep, e, conv))
for kind, h, w, div, sep, e, conv in cases:
    img = synth.image_u8(h, w, seed=h + w)
    depth = synth.DEPTHS[kind](h, w) * np.float32(255)
    want = oracle.apply_stereo_divergence(img, depth, div, sep, e, "polylines_soft", conv)
    ti, td = torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda()
    _native.debug_set("pt_variant", 0)
    got = engine.apply_stereo_divergence(ti, td, div, sep, e, "polylines_soft", conv).cpu().numpy()
    _native.debug_set("pt_variant", 9)
    old = engine.apply_stereo_divergence(ti, td, div, sep, e, "polylines_soft", conv).cpu().numpy()
    _native.debug_set("pt_variant", 0)
    bad = np.nonzero((got != want).any(-1))
    badold = int((old != want).any(-1).sum())
    tag = f"{kind:10s} {h}x{w} div {div} sep {sep} e {e} conv {conv}"
    if len(bad[0]) == 0 and badold == 0:
        print("OK  ", tag)
        continue
    bad_total += len(bad[0])
    print("FAIL", tag, f": {len(bad[0])} pixels differ (first generation: {badold})")
    x = geometry(depth, div, sep, e, conv, w)
    for r, c in list(zip(*bad))[:6]:
        print(f"     row {r} col {c}: got {got[r, c].tolist()} want {want[r, c].tolist()} | {describe(x[r], c)}")
# node path (both eyes, SBS, float outputs, mask, depth maps)
for kind, blur in (("stepped", True), ("blobs", False), ("radial", True)):
    n, h, w = 2, 64, 1400
    img = synth.image_f32(n, h, w, seed=4)
    dep = synth.depth_batch(kind, n, h, w, channels=3)
    args = (7.0, 0.2, "left-right", 0.1, 0.5, 2.0)
    got = engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), *args, "polylines_soft", 20.0, 20.0, blur,
                          depth_blur_falloff=2.0, depth_blur_vert_smooth=3, batch_size=12)
    want = node_oracle.generate(img, dep, *args, "Fill - Polylines Soft", 20.0, 20.0, blur, depth_blur_falloff=2.0,
                                depth_blur_vert_smooth=3, batch_size=12)
    res = [int((g.cpu().numpy() != wv).sum()) for g, wv in zip(got, want)]
    print("node", kind, "blur", blur, "mismatching values (stereo, dl, dr, mask):", res)
    if res[0]:
        gs = np.round(got[0].cpu().numpy() * 255).astype(int)
        ws = np.round(want[0] * 255).astype(int)
        bad = np.argwhere((gs != ws).any(-1))
        rows = sorted(set((int(f), int(r)) for f, r, c in bad))
        print(f"     {len(bad)} pixels in {len(rows)} rows; first rows {rows[:6]}")
        for f, r, c in bad[:8]:
            print(f"     frame {f} row {r} eye {c // w} col {c % w}: got {gs[f, r, c].tolist()} want {ws[f, r, c].tolist()}")
        f, r = rows[0]
        cols = sorted(int(c) for ff, rr, c in bad if ff == f and rr == r)
        print(f"     row ({f},{r}) bad cols: {cols[:40]}")
    bad_total += sum(res)
print("TOTAL mismatches:", bad_total)
sys.exit(1 if bad_total else 0)
