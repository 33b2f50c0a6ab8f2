"""Deterministic synthetic inputs (our own generators).

Used by tools/make_goldens.py (fixtures), tests/ and bench.py.  Pure numpy, no reference code.
"""
import numpy as np


def radial(h, w, cx=None, cy=None):
    """1 - r/r.max(), r = distance to the centre (float64 math, float32 result)."""
    cy = h / 2 if cy is None else cy
    cx = w / 2 if cx is None else cx
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = np.sqrt((y - cy) ** 2 + (x - cx) ** 2)
    return (1.0 - r / r.max()).astype(np.float32)


def stepped(h, w, levels=6, **kw):
    """Radial depth quantised to `levels` plateaus: sharp discontinuities -> disocclusions."""
    return (np.floor(radial(h, w, **kw).astype(np.float64) * levels) / levels).astype(np.float32)


def noisy_ramp(h, w, seed=0, amp=0.08):
    rng = np.random.default_rng(seed)
    ramp = np.linspace(0.0, 1.0, w, dtype=np.float64)[None, :].repeat(h, 0)
    return np.clip(ramp + amp * rng.standard_normal((h, w)), 0, 1).astype(np.float32)


def random8(h, w, seed=0):
    """Random 8-bit-origin depth (k/255): worst-case folding, exercises ties and long active lists."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (h, w)).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def blobs(h, w, seed=0, n=4):
    """Smooth background ramp with `n` flat elliptical foreground objects (like a real depth map)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    d = 0.15 + 0.35 * (y / max(h - 1, 1))
    for _ in range(n):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.1, 0.9) * w
        ry, rx = rng.uniform(0.08, 0.25) * h, rng.uniform(0.05, 0.2) * w
        lvl = rng.uniform(0.5, 1.0)
        d = np.where(((y - cy) / ry) ** 2 + ((x - cx) / rx) ** 2 < 1.0, lvl, d)
    return d.astype(np.float32)


def clipped(h, w, seed=0):
    """Saturated depth: blobs clipped to exactly 0 / 1 over large areas with smooth ramps between -- at convergence 0.5 the two
    plateaus have bit-equal |disparity|, so every overlap of a near and a far layer is an exact closeness tie (the realistic
    source of order-dependent rows: depth estimators saturate)."""
    d = (blobs(h, w, seed=seed) - np.float32(0.5)) * np.float32(4.0) + np.float32(0.5)
    return np.clip(d, 0.0, 1.0).astype(np.float32)


DEPTHS = {"clipped": clipped, "radial": radial, "stepped": stepped, "noisy_ramp": noisy_ramp, "random8": random8, "blobs": blobs}


def image_u8(h, w, seed=0, hazards=True):
    """Random uint8 RGB; with `hazards`, seeds the pixels that exercise the uint8-wrap quirk (Q5):
    (128,128,0) and (255,1,0) sum to 0 mod 256, plus genuinely black pixels."""
    rng = np.random.default_rng(1000 + seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if hazards and h * w >= 64:
        idx = rng.choice(h * w, size=max(3, h * w // 40), replace=False)
        flat = img.reshape(-1, 3)
        flat[idx[0::3]] = (128, 128, 0)
        flat[idx[1::3]] = (255, 1, 0)
        flat[idx[2::3]] = (0, 0, 0)
    return img


def image_f32(n, h, w, seed=0):
    """ComfyUI IMAGE batch [N,H,W,3] float32 in 0..1 with 8-bit-origin values (k/255)."""
    rng = np.random.default_rng(1000 + seed)
    return (rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8).astype(np.float32) / np.float32(255.0))


def depth_batch(kind, n, h, w, channels=3):
    """ComfyUI depth IMAGE batch [N,H,W,C]: per-frame moving centre for radial/stepped."""
    frames = []
    for i in range(n):
        if kind in ("radial", "stepped"):
            cx = w / 2 + (17 * i) % max(w // 4, 1)
            cy = h / 2 + (11 * i) % max(h // 4, 1)
            frames.append(DEPTHS[kind](h, w, cx=cx, cy=cy))
        else:
            frames.append(DEPTHS[kind](h, w, seed=i))
    d = np.stack(frames)[..., None]
    return np.ascontiguousarray(np.repeat(d, channels, axis=-1)).astype(np.float32)
