"""Where does polylines_sharp under the float64 disparity chain differ from the reference fixture?  (development aid)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from comfystereo_amd import engine, _native
from oracle import oracle
z = np.load(os.path.join(ROOT, "tests", "golden", "dialect_f64.npz"))
cases = json.loads(str(z["meta"]))["cases"]
for no_tile in (0, 1):
    _native.debug_set("no_tile", no_tile)
    for c in cases:
        for fill in ("polylines_soft", "polylines_sharp"):
            img, dep = z[f"{c['id']}/img"], z[f"{c['id']}/depth"]
            got = engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), c["divergence"], c["separation"],
                                                 c["exponent"], fill, c["convergence"], dialect="f64-disparity").cpu().numpy()
            want = z[f"{c['id']}/{fill}"]
            bad = np.argwhere(got != want)
            print("no_tile", no_tile, "case", c["id"], c["kind"], fill, "div", c["divergence"], "e", c["exponent"], "mismatches", len(bad),
                  [(int(a), int(b), int(ch), int(got[a, b, ch]), int(want[a, b, ch])) for a, b, ch in bad[:9]])

# the dialect fuzz's first mismatch (seed 819721: one row of 5 600 columns, fill none, divergence -14): which kernel, which dialect?
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import make_case
rng = np.random.default_rng(819721)
img, depth, div, sep, e, conv = make_case(rng)
if rng.random() < 0.5:
    reps = int(rng.integers(3, 9))
    img = np.concatenate([np.roll(img, int(rng.integers(0, img.shape[1])), axis=1) for _ in range(reps)], axis=1)
    depth = np.concatenate([np.roll(depth, int(rng.integers(0, depth.shape[1])), axis=1) for _ in range(reps)], axis=1)
    if rng.random() < 0.4:
        w = depth.shape[1]
        a = int(rng.integers(0, w // 2)); depth[:, a:a + w // 3] = 250.0 if rng.random() < 0.5 else 5.0
        div = float(rng.choice([-14.0, 14.0, 9.0]))
print("fuzz case", img.shape, div, sep, e, conv)
want = {}
for d in ("D32", "f64-disparity"):
    oracle.set_dialect(d)
    want[d] = {f: oracle.apply_stereo_divergence(img, depth, div, sep, e, f, conv) for f in ("none", "naive", "inverse")}
for no_tile in (0, 1):
    _native.debug_set("no_tile", no_tile)
    for f in ("none", "naive", "inverse"):
        for d in ("D32", "f64-disparity"):
            got = engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), div, sep, e, f, conv, dialect=d).cpu().numpy()
            print("no_tile", no_tile, f, "engine", d, "vs oracle D32:", int((got != want["D32"][f]).sum()), " vs oracle f64:", int((got != want["f64-disparity"][f]).sum()))
