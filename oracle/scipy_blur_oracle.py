"""TEST INFRASTRUCTURE (oracle): the reference's scipy depth blur, `directional_motion_blur`
(/root/reference/stereoimage_generation.py:1346-1419) -- what `create_stereoimages` runs when it is handed numpy / PIL inputs with
the depth blur on (:1486-1494).  Only tests/ and tools/ import this; the product path is cs_scipyblur.hip.

The arithmetic lives in a third-party dependency that is not under /root/reference: scipy.ndimage (`sobel`, `convolve1d`;
requirements.txt pins no version, this container has scipy 1.15.3).  Restated here from its published algorithm
(scipy/ndimage/src/ni_filters.c, NI_Correlate1D) operation by operation:

  * every line is converted to float64, the correlation is accumulated in float64 and the result is rounded ONCE into the
    float32 output array;
  * odd kernels that are symmetric / antisymmetric (|w[c+i] -+ w[c-i]| <= DBL_EPSILON) take the folded loops
        o = x[0] * w[c];  for j = -size1 .. -1:  o += (x[j] +- x[-j]) * w[c + j]
    all others   o = x[size2] * w[c + size2];  for j = -size1 .. size2 - 1:  o += x[j] * w[c + j]      (size1 = n // 2, size2 = n - size1 - 1);
  * convolve1d = correlate1d with the kernel reversed and, for even kernels, the origin moved by -1: an even box of k taps covers
    the columns [l - k/2 + 1, l + k/2];
  * borders: 'reflect' (sobel's default: d c b a | a b c d | d c b a) and 'nearest' (the reference's convolve1d calls).
  * sobel(axis=1) = correlate1d([-1, 0, 1]) along x into a float32 array, then correlate1d([1, 2, 1]) along y.
Pinned by tests/golden/numpy_blur.npz (outputs of the reference itself, tools/make_goldens.py --only-numpy-blur): bit-exact for the
falloff exponents NumPy evaluates exactly (2.0: x * x, 1.0, 0.5: sqrt); any other exponent goes through `np.power` on a float32
ARRAY, which NumPy vectorises with its own SIMD routine on AVX-512 machines (not glibc's powf: one ulp apart now and then) --
there the restatement uses powf and the tests carry a tolerance, as for torch.pow on the tensor path.
"""
import ctypes

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def correlate1d(a32, weights, axis, mode, origin=0):
    """scipy.ndimage.correlate1d(float32 array, float64 weights) -> float32 (ni_filters.c NI_Correlate1D)."""
    a = np.moveaxis(np.asarray(a32, dtype=np.float32).astype(np.float64), axis, -1)
    n = a.shape[-1]
    w = np.asarray(weights, dtype=np.float64)
    fs = len(w)
    size1 = fs // 2
    size2 = fs - size1 - 1

    def at(idx):
        if mode == "nearest":
            j = np.clip(idx, 0, n - 1)
        elif mode == "reflect":
            j = np.mod(idx, 2 * n)
            j = np.where(j >= n, 2 * n - 1 - j, j)
        else:
            raise ValueError(mode)
        return a[..., j]

    l = np.arange(n)
    eps = np.finfo(np.float64).eps
    symmetric = 0
    if fs & 1:
        symmetric = 1
        for i in range(1, fs // 2 + 1):
            if abs(w[i + size1] - w[size1 - i]) > eps:
                symmetric = 0
                break
        if symmetric == 0:
            symmetric = -1
            for i in range(1, fs // 2 + 1):
                if abs(w[size1 + i] + w[size1 - i]) > eps:
                    symmetric = 0
                    break

    def x(j):
        return at(l + j - origin)

    if symmetric > 0:
        o = x(0) * w[size1]
        for j in range(-size1, 0):
            o = o + (x(j) + x(-j)) * w[j + size1]
    elif symmetric < 0:
        o = x(0) * w[size1]
        for j in range(-size1, 0):
            o = o + (x(j) - x(-j)) * w[j + size1]
    else:
        o = x(size2) * w[size2 + size1]
        for j in range(-size1, size2):
            o = o + x(j) * w[j + size1]
    return np.moveaxis(o.astype(np.float32), -1, axis)


def convolve1d(a32, weights, axis, mode):
    """scipy.ndimage.convolve1d: the kernel reversed, even kernels with origin -1 (scipy/ndimage/_filters.py)."""
    w = np.asarray(weights, dtype=np.float64)[::-1]
    return correlate1d(a32, w, axis, mode, -1 if (len(w) & 1) == 0 else 0)


def sobel_x(depth):
    """scipy.ndimage.sobel(depth, axis=1), default mode 'reflect' (reference :1381)."""
    return correlate1d(correlate1d(depth, [-1.0, 0.0, 1.0], 1, "reflect"), [1.0, 2.0, 1.0], 0, "reflect")


def power_f32(base, exponent):
    """float32 array ** Python float as NumPy evaluates it: exact shortcuts for 2 / 1 / 0.5, powf otherwise (see the header)."""
    base = np.asarray(base, dtype=np.float32)
    if exponent == 2.0:
        return base * base
    if exponent == 1.0:
        return base.copy()
    if exponent == 0.5:
        return np.sqrt(base)
    flat = np.array([_libm.powf(float(v), float(exponent)) for v in base.ravel()], dtype=np.float32)
    return flat.reshape(base.shape)


def directional_motion_blur(depth, blur_strength, edge_threshold, blur_mask_width=5, falloff_exponent=1.0, vert_smooth_px=0):
    """reference :1346-1419 on a float32 [H, W] depth map -> (left, right) float32."""
    depth = np.asarray(depth, dtype=np.float32)
    if blur_strength <= 0:                                                    # :1374
        return depth, depth
    bs = int(round(blur_strength))                                            # :1377
    radius = int(blur_mask_width)                                             # :1378
    h, w = depth.shape
    grad = sobel_x(depth)                                                     # :1381
    edge = np.abs(grad) / np.float32(10 * edge_threshold)                     # :1383 (float32 array / Python float)
    edge = np.clip(edge, 0, 1)
    left_mask = (grad > 0) & (edge > 0.5)                                     # :1385-1386
    right_mask = (grad < 0) & (edge > 0.5)
    cols = np.arange(w, dtype=np.float32)
    large = np.float32(radius + 1)

    def dist_weight(mask):                                                    # :1393-1404
        col_l = np.where(mask, np.broadcast_to(cols, (h, w)), np.float32(-1.0))
        last_l = np.maximum.accumulate(col_l, axis=1)
        dist_l = np.where(last_l >= 0, cols[None, :] - last_l, large)
        col_r = np.where(mask[:, ::-1], np.broadcast_to(cols, (h, w)), np.float32(-1.0))
        last_r = np.maximum.accumulate(col_r, axis=1)
        dist_r = np.where(last_r >= 0, cols[None, :] - last_r, large)[:, ::-1]
        dist = np.minimum(dist_l, dist_r)
        with np.errstate(divide="ignore", invalid="ignore"):
            base = np.clip(np.float32(1.0) - dist / np.float32(radius), np.float32(0.0), np.float32(1.0))
        return power_f32(base, falloff_exponent)

    lw, rw = dist_weight(left_mask), dist_weight(right_mask)
    if vert_smooth_px > 0:                                                    # :1410-1413
        vk = np.ones(2 * vert_smooth_px + 1) / (2 * vert_smooth_px + 1)
        lw = np.clip(convolve1d(lw, vk, 0, "nearest"), np.float32(0.0), np.float32(1.0))
        rw = np.clip(convolve1d(rw, vk, 0, "nearest"), np.float32(0.0), np.float32(1.0))
    bk = np.ones(bs) / bs                                                     # :1416-1418 (left and right kernels hold the same values)
    blurred = convolve1d(depth, bk, 1, "nearest")
    left = lw * blurred + (np.float32(1.0) - lw) * depth                      # :1420-1421
    right = rw * blurred + (np.float32(1.0) - rw) * depth
    return left, right
