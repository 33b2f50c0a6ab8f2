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


def scene8(h, w, seed=0, soften=True):
    """Quantised smooth depth with object edges -- what real estimators deliver as an 8-bit map, the kind of depth the reference's own
    fixture maker draws (/root/reference/create_test_images.py:3-77, described, not copied): an 8-bit vertical gradient as the far-to-mid
    background, three flat ellipses (far, mid, near: levels 100 / 170 / 240) with hard silhouettes and a slightly brighter rim, values
    k / 255.  `seed` moves the ellipses a little (frames of a batch differ); `soften`: a mild [1 2 1] / 4 blur across the silhouettes,
    re-quantised to 8 bits (depth estimators do not deliver single-pixel steps)."""
    rng = np.random.default_rng(7000 + seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    d = np.floor(80.0 + (y / h) * 50.0)
    rim = max(2.0, 5.0 * min(h, w) / 600.0)
    # (centre x, centre y, radius x, radius y) as fractions of the frame; fill level, rim level
    for (cx, cy, rx, ry), lvl, rim_lvl in (((0.28, 0.375, 0.094, 0.125), 100.0, 120.0), ((0.5625, 0.5, 0.125, 0.167), 170.0, 190.0),
                                          ((0.375, 0.75, 0.125, 0.167), 240.0, 255.0)):
        cx, cy = (cx + rng.uniform(-0.04, 0.04)) * w, (cy + rng.uniform(-0.04, 0.04)) * h
        rx, ry = rx * w, ry * h
        q = ((x - cx) / rx) ** 2 + ((y - cy) / ry) ** 2
        q_in = ((x - cx) / max(rx - rim, 1.0)) ** 2 + ((y - cy) / max(ry - rim, 1.0)) ** 2
        d = np.where(q < 1.0, rim_lvl, d)
        d = np.where(q_in < 1.0, lvl, d)
    if soften:
        for axis in (0, 1):
            p = np.pad(d, [(1, 1) if a == axis else (0, 0) for a in (0, 1)], mode="edge")
            lo = p[:-2] if axis == 0 else p[:, :-2]
            hi = p[2:] if axis == 0 else p[:, 2:]
            d = (lo + 2.0 * d + hi) / 4.0
        d = np.floor(d + 0.5)
    return (np.clip(d, 0, 255).astype(np.float32) / np.float32(255.0)).astype(np.float32)


DEPTHS = {"clipped": clipped, "radial": radial, "stepped": stepped, "noisy_ramp": noisy_ramp, "random8": random8, "blobs": blobs,
          "scene8": scene8}


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
