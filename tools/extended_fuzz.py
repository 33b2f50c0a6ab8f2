"""Longer seeded fuzz than tests/test_gpu_fuzz.py (development aid, run on the GPU box): wide rows (several tiles, holes
longer than the parallel walks of naive_interpolating, halos up to the tiled path's limit) and node-level runs with
the depth blur, every fill.  Prints the first mismatch and exits 1, else a summary."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from oracle import node_oracle, oracle
from comfystereo_amd import engine
from test_gpu_fuzz import FILLS, make_case

if os.environ.get("CS_DBG"):   # a development switch for the whole run (30: the polylines row kernel in two column ranges)
    from comfystereo_amd import _native
    _native.debug_set("dbg", int(os.environ["CS_DBG"]))
if os.environ.get("CS_FUZZ_FILLS"):   # restrict the run to some techniques (a kernel under development): "polylines_soft,polylines_sharp"
    FILLS = [f for f in FILLS if f in os.environ["CS_FUZZ_FILLS"].split(",")]

# CS_FUZZ_DIALECT = f64-disparity | int64-sum | D64: the whole run under that arithmetic dialect (oracle and engine alike; gpu_warp has
# none and is left out) -- the dialect instantiations of the tile kernels (round 5)
DIALECT = os.environ.get("CS_FUZZ_DIALECT", "D32")
oracle.set_dialect(DIALECT)
engine.DIALECT = DIALECT
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
t0 = time.time()
n_asd = n_node = 0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
while time.time() - t0 < budget * 0.6:
    rng = np.random.default_rng(seed); seed += 1
    img, depth, div, sep, e, conv = make_case(rng)
    if rng.random() < 0.5:  # widen: tile the case horizontally with a random shift so that it spans several tiles
        reps = int(rng.integers(3, 9))
        img = np.concatenate([np.roll(img, int(rng.integers(0, img.shape[1])), axis=1) for _ in range(reps)], axis=1)
        depth = np.concatenate([np.roll(depth, int(rng.integers(0, depth.shape[1])), axis=1) for _ in range(reps)], axis=1)
        if rng.random() < 0.4:  # a long hole: a wide far plateau next to a near one, large divergence
            w = depth.shape[1]
            a = int(rng.integers(0, w // 2)); depth[:, a:a + w // 3] = 250.0 if rng.random() < 0.5 else 5.0
            div = float(rng.choice([-14.0, 14.0, 9.0]))
    for fill in FILLS:
        try:
            want = oracle.apply_stereo_divergence(img, depth, div, sep, e, fill, conv)
        except IndexError:
            continue
        try:
            got = engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), div, sep, e, fill,
                                                 conv, dialect=DIALECT).cpu().numpy()
        except RuntimeError as ex:   # (a dialect's row kernels keep 8 more bytes of LDS per column: a widened case can exceed them)
            if DIALECT != "D32" and "too wide" in str(ex):
                continue
            raise
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            print("MISMATCH asd", seed - 1, fill, img.shape, div, sep, e, conv, len(bad), bad[:3].tolist()); sys.exit(1)
        n_asd += 1
ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}
# (under a float64 dialect only the exponents the kernels evaluate exactly: for any other one the device library's pow() is within 1 ulp
# of libm's, and a result differs where that ulp crosses a rounding boundary -- DESIGN.md section 2; the node cases of session r05_s24 hit
# it once in 5 600 cases with exponent 1.4)
EXPONENTS = [1.0, 2.0, 1.4] if DIALECT == "D32" else [1.0, 2.0]
modes = ["left-right", "right-left", "top-bottom", "bottom-top", "red-cyan-anaglyph"]
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed); seed += 1
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 70)), int(rng.choice([64, 200, 516, 1028, 1540]))
    img = synth.image_f32(n, h, w, seed=seed)
    depth = synth.depth_batch(str(rng.choice(["blobs", "stepped", "radial", "noisy_ramp", "clipped", "clipped", "random8"])), n, h, w, channels=3)
    fill = str(rng.choice([f for f in FILLS if f in ui] + (["gpu_warp"] if DIALECT == "D32" else [])))  # (node level: the techniques a UI string reaches)
    args = (float(rng.choice([2.0, 5.0, 8.0, 12.0])), float(rng.choice([0.0, 0.5, -1.0])), str(rng.choice(modes)),
            float(rng.choice([0.0, 0.3, -0.5])), float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice(EXPONENTS)))
    blur = (float(rng.choice([20.0, 5.0, 33.0])), float(rng.choice([20.0, 3.0])), bool(rng.random() < 0.8))
    kw = dict(depth_blur_falloff=float(rng.choice([2.0, 1.0, 0.5, 3.0, 1.7])), depth_blur_vert_smooth=int(rng.integers(0, 8)),
              batch_size=int(rng.integers(1, 4)))
    want = node_oracle.generate(img, depth, *args, ui[fill], *blur, **kw)
    got = [t.cpu().numpy() for t in engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), *args, fill,
                                                    *blur, **kw)]
    for k, (g, w_) in enumerate(zip(got, want)):
        ok = np.abs(g - w_).max() <= 1e-4 if (fill == "gpu_warp" and k == 0) else np.array_equal(g, w_)
        if not ok:
            print("MISMATCH node", seed - 1, fill, (n, h, w), args, blur, kw, "output", k); sys.exit(1)
    n_node += 1
print(f"dialect {DIALECT}; ", end="")
print(f"extended fuzz OK: {n_asd} divergence cases, {n_node} node cases in {time.time() - t0:.0f} s")
