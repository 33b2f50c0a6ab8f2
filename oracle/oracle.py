"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (as the
checker, never as the thing measured or shipped).  The product package never imports it.
Parity status: PINNED against tests/golden/ (fixtures captured from the imported reference by
tools/make_goldens.py); see tests/test_oracle_goldens.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CS_ORACLE_LIB: another build of the checker, e.g. the AddressSanitizer one (`make -C oracle asan`)
_LIB_PATH = os.environ.get("CS_ORACLE_LIB") or os.path.join(_HERE, "libstereo_oracle.so")

FILLS = {
    "none": 0,
    "naive": 1,
    "naive_interpolating": 2,
    "polylines_soft": 3,
    "polylines_sharp": 4,
    "inverse": 5,
    "hybrid_edge": 6,
    "none_post": 8,
    "inverse_post": 9,
    "hybrid_edge_plus": 10,
}

_lib = None


def build(force=False):
    """Compile oracle/libstereo_oracle.so with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("stereo_oracle.c", "oracle_math.h", "libm_exp_table.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, f32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_float)
        c_int, c_double = ctypes.c_int, ctypes.c_double
        L.oracle_set_threads.restype = None
        L.oracle_set_threads.argtypes = [c_int]
        L.oracle_max_threads.restype = c_int
        L.oracle_set_threads(1)  # single-threaded unless a caller (bench.py's cpu_baseline) asks for more
        L.oracle_powf.restype = ctypes.c_float
        L.oracle_powf.argtypes = [ctypes.c_float, ctypes.c_float]
        L.oracle_exp.restype = c_double
        L.oracle_exp.argtypes = [c_double]
        L.oracle_naive.restype = None
        L.oracle_naive.argtypes = [u8p, f32p, c_int, c_int, c_double, c_double, c_double, c_int, u8p]
        L.oracle_polylines.restype = c_int
        L.oracle_polylines.argtypes = [u8p, f32p, c_int, c_int, c_double, c_double, c_double, c_int, u8p]
        L.oracle_inverse.restype = None
        L.oracle_inverse.argtypes = [u8p, f32p, c_int, c_int, c_double, c_double, c_double, u8p]
        L.oracle_hybrid_edge.restype = None
        L.oracle_hybrid_edge.argtypes = [u8p, f32p, c_int, c_int, c_double, c_double, c_double, u8p, u8p]
        L.oracle_apply_stereo_divergence.restype = c_int
        L.oracle_apply_stereo_divergence.argtypes = [u8p, f32p, c_int, c_int, c_double, c_double, c_double, c_int,
                                                     c_double, u8p, f32p]
        if hasattr(L, "oracle_blur"):
            L.oracle_blur.restype = None
            L.oracle_blur.argtypes = [f32p, c_int, c_int, c_int, c_double, c_double, c_double, c_int, f32p, f32p]
            L.oracle_blur2.restype = None
            L.oracle_blur2.argtypes = [f32p, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int, f32p, f32p]
        if hasattr(L, "oracle_forward_warp_gpu"):
            L.oracle_forward_warp_gpu.restype = None
            L.oracle_forward_warp_gpu.argtypes = [f32p, f32p, c_int, c_int, c_int, c_double, c_double, c_double,
                                                  c_double, f32p, u8p]
            L.oracle_forward_warp_gpu2.restype = None
            L.oracle_forward_warp_gpu2.argtypes = [f32p, f32p, c_int, c_int, c_int, c_double, c_double, c_double,
                                                   c_double, c_double, c_int, f32p, u8p]
        if hasattr(L, "oracle_set_dialect"):
            L.oracle_set_dialect.restype = None
            L.oracle_set_dialect.argtypes = [c_int]
        if hasattr(L, "oracle_forward_warp_mesh"):
            L.oracle_forward_warp_mesh.restype = None
            L.oracle_forward_warp_mesh.argtypes = [f32p, f32p, c_int, c_int, c_int, c_double, c_double, c_double,
                                                   c_double, c_double, f32p, u8p]
        _lib = L
    return _lib


def set_dialect(name):
    """"D32" (default, pinned) | "f64-disparity" | "int64-sum" | "D64": see stereo_oracle.c (g_dialect)."""
    lib().oracle_set_dialect({"D32": 0, "f64-disparity": 1, "int64-sum": 2, "D64": 3}[name])


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def powf(x, y):
    return float(lib().oracle_powf(float(x), float(y)))


def set_threads(n):
    """Rows of the polylines technique and of the depth blur run on `n` OpenMP threads (the analogue of the reference's
    numba `prange`); 1 = the default.  Returns the number of host cores OpenMP sees."""
    L = lib()
    L.oracle_set_threads(int(n))
    return L.oracle_max_threads()


def apply_stereo_divergence(img_u8_hwc, depth_f32_hw, divergence, separation, exponent, fill, convergence=0.5,
                            return_nd=False):
    """reference stereoimage_generation.py:1576-1620 (`apply_stereo_divergence`)."""
    img = np.ascontiguousarray(img_u8_hwc, dtype=np.uint8)
    depth = np.ascontiguousarray(depth_f32_hw, dtype=np.float32)
    h, w = depth.shape
    assert img.shape == (h, w, 3)
    out = np.empty_like(img)
    nd = np.empty_like(depth)
    rc = lib().oracle_apply_stereo_divergence(_u8(img), _f32(depth), h, w, float(divergence), float(separation),
                                              float(exponent), FILLS[fill], float(convergence), _u8(out), _f32(nd))
    if rc == -1:
        raise IndexError("csg overflow (the reference raises IndexError here)")
    return (out, nd) if return_nd else out


def blur(depth_f32, strength, edge_threshold, falloff=1.0, vert_smooth=0, mask_width=None):
    """reference stereoimage_generation.py:1171-1251 (`directional_motion_blur_gpu`), depth [B,H,W] or [H,W].
    mask_width: `blur_mask_width` (the reference's callers pass the blur strength, the default here)."""
    d = np.ascontiguousarray(depth_f32, dtype=np.float32)
    shp = d.shape
    d3 = d.reshape((-1,) + shp[-2:])
    L = np.empty_like(d3)
    R = np.empty_like(d3)
    lib().oracle_blur2(_f32(d3), d3.shape[0], d3.shape[1], d3.shape[2], float(strength), float(edge_threshold),
                       float(strength if mask_width is None else mask_width), float(falloff), int(vert_smooth), _f32(L), _f32(R))
    return L.reshape(shp), R.reshape(shp)


def forward_warp_gpu(image_f32_bchw, depth_f32_bhw, divergence_px, separation_px, exponent, convergence=0.5,
                     gradient_threshold=1.5, max_stretch=8):
    """reference stereoimage_generation.py:277-450 (`forward_warp_gpu`) -> (warped [B,C,H,W] f32, gap mask bool)."""
    img = np.ascontiguousarray(image_f32_bchw, dtype=np.float32)
    dep = np.ascontiguousarray(depth_f32_bhw, dtype=np.float32)
    b, c, h, w = img.shape
    assert c == 3 and dep.shape == (b, h, w)
    out = np.empty_like(img)
    mask = np.empty((b, h, w), dtype=np.uint8)
    lib().oracle_forward_warp_gpu2(_f32(img), _f32(dep), b, h, w, float(divergence_px), float(separation_px),
                                   float(exponent), float(convergence), float(gradient_threshold), int(max_stretch), _f32(out), _u8(mask))
    return out, mask.astype(bool)


def forward_warp_mesh(image_f32_bchw, depth_f32_bhw, divergence_px, separation_px, exponent, convergence=0.5,
                      gradient_threshold=1.5):
    """The mesh-quality warp (reference stereoimage_generation.py:453-689, `forward_warp_mesh`) as specified in
    stereo_oracle.c -- PARITY UNPINNED (no moderngl / OpenGL here) -> (warped [B,C,H,W] f32, gap mask bool)."""
    img = np.ascontiguousarray(image_f32_bchw, dtype=np.float32)
    dep = np.ascontiguousarray(depth_f32_bhw, dtype=np.float32)
    b, c, h, w = img.shape
    assert c == 3 and dep.shape == (b, h, w)
    out = np.empty_like(img)
    mask = np.empty((b, h, w), dtype=np.uint8)
    lib().oracle_forward_warp_mesh(_f32(img), _f32(dep), b, h, w, float(divergence_px), float(separation_px),
                                   float(exponent), float(convergence), float(gradient_threshold), _f32(out), _u8(mask))
    return out, mask.astype(bool)
