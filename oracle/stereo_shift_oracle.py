"""CPU restatement of `stereo_shift_torch` (reference stereo_utils.py:15-88) -- TEST INFRASTRUCTURE ONLY.

numpy, float32 like CPU torch executes it for float32 inputs: the depth is normalised with the GLOBAL min / max of the whole
tensor (:36-45), `depth ** e` is exact for e = 1 (the reference's callers), 2 (x*x) and 0.5 (sqrt), the shift is
`int(depth_val * scale_factor_px)` with the product rounded to float32 and truncated toward zero (:63-64), and the sweep
order makes the largest source column win for a negative shift, the smallest for a positive one (:56-67).
Pinned by tests/golden/stereo_shift.npz (outputs of the imported reference, tools/make_goldens.py --only-stereo-shift).
Only tests/ may import this module."""
import numpy as np

from . import oracle

F32 = np.float32


def _norm_depth(depth):
    d = np.asarray(depth, dtype=F32)
    mn, mx = d.min(), d.max()
    if F32(mx - mn) > np.finfo(np.float32).eps:
        return (F32(1.0) * (d - mn)) / F32(mx - mn)
    return np.zeros_like(d)


def _create_stereo(inp, nd, scale_factor, exponent):
    b, c, h, w = inp.shape
    out = np.zeros_like(inp)
    scale_px = (scale_factor / 100.0) * w
    if exponent == 1.0:
        dv = nd
    elif exponent == 2.0:
        dv = nd * nd
    elif exponent == 0.5:
        dv = np.sqrt(nd)
    else:
        L = oracle.lib()
        dv = np.array([L.oracle_powf(float(v), float(F32(exponent))) for v in nd.ravel()], dtype=F32).reshape(nd.shape)
    off = np.trunc((dv * F32(scale_px)).astype(np.float64)).astype(np.int64)
    cols = np.arange(w)
    for bi in range(b):
        for r in range(h):
            cd = cols + off[bi, r]
            ok = (cd >= 0) & (cd < w)
            if scale_px < 0:   # ascending sweep: the later (larger) source column overwrites
                win = np.full(w, -1, dtype=np.int64)
                np.maximum.at(win, cd[ok], cols[ok])
            else:              # descending sweep: the smallest source column is written last
                win = np.full(w, 1 << 40, dtype=np.int64)
                np.minimum.at(win, cd[ok], cols[ok])
                win[win == (1 << 40)] = -1
            hit = win >= 0
            out[bi][:, r, hit] = inp[bi][:, r, :][:, win[hit]]
    return out


def stereo_shift(input_images, depthmaps, scale_factor=8.0, shift_both=False, stereo_offset_exponent=1.0):
    inp = np.ascontiguousarray(input_images, dtype=F32)
    nd = _norm_depth(depthmaps)
    if not shift_both:
        left, balance = inp, 0.0
    else:
        balance = 0.5
        left = _create_stereo(inp, nd, +1 * scale_factor * balance, stereo_offset_exponent)
    right = _create_stereo(inp, nd, -1 * scale_factor * (1 - balance), stereo_offset_exponent)
    return np.concatenate([left, right], axis=0)
