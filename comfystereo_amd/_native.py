"""ctypes binding of libcomfystereo_hip.so (the C ABI declared in include/comfystereo_amd.h).

There is no CPU fallback: if the HIP library is missing or does not load, importing this module's
`lib()` raises.  Build it with `python __graft_entry__.py` or `make -C comfystereo_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CS_LIB_PATH: development override (A/B timing of two builds of the same library in one GPU session)
LIB_PATH = os.environ.get("CS_LIB_PATH") or os.path.join(_HERE, "libcomfystereo_hip.so")

CS_OK, CS_EINVAL, CS_EWORKSPACE, CS_ELIMIT, CS_EHIP = 0, -1, -2, -3, -4

FILL = {
    "none": 0, "naive": 1, "naive_interpolating": 2, "polylines_soft": 3, "polylines_sharp": 4, "inverse": 5,
    "hybrid_edge": 6, "gpu_warp": 7,
    # branches of the reference dispatcher no UI string reaches (stereoimage_generation.py:1605-1610)
    "none_post": 8, "inverse_post": 9, "hybrid_edge_plus": 10,
}
MODE = {
    "left-right": 0, "right-left": 1, "top-bottom": 2, "bottom-top": 3, "red-cyan-anaglyph": 4, "left-only": 5,
    "only-right": 6, "cyan-red-reverseanaglyph": 7,
}

# the ABI version the ctypes signatures below were written for (include/comfystereo_amd.h CS_ABI_VERSION)
ABI_VERSION = 4

EXPORTS = [
    "cs_version", "cs_last_error", "cs_max_width", "cs_max_width_mode", "cs_max_width_params", "cs_output_shape", "cs_workspace_bytes", "cs_generate",
    "cs_asd_workspace_bytes", "cs_asd_workspace_bytes_for", "cs_apply_stereo_divergence", "cs_apply_stereo_divergence2", "cs_blur_workspace_bytes", "cs_directional_blur", "cs_blur_scipy_workspace_bytes", "cs_directional_blur_scipy",
    "cs_warp_workspace_bytes", "cs_forward_warp", "cs_forward_warp2", "cs_warp_mesh_workspace_bytes", "cs_forward_warp_mesh", "cs_expand_u8", "cs_pack_u8", "cs_host_expand_u8", "cs_host_copy", "cs_take_f32", "cs_host_replicate_f32", "cs_stereo_shift_workspace_bytes", "cs_stereo_shift", "cs_profile", "cs_profile_read", "cs_profile_tiles", "cs_debug_set",
    "cs_test_powf", "cs_test_exp", "cs_test_edge_threshold",
]

# enum cs_debug_key (development switches; tests and profiling tools only)
DEBUG = {"dbg": 0, "no_tile": 1, "pt_variant": 2, "blur_two_pass": 3, "blur_edges_scalar": 4, "blur_full_copy": 5, "chunks": 6, "no_replay_kernel": 7,
         "blur_no_pre_edges": 8, "hybrid_unfused": 9, "gpuwarp_full_maps": 10,
         "hybrid_full_maps": 11}


class Params(ctypes.Structure):
    """struct cs_params (include/comfystereo_amd.h)."""
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("depth_h", ctypes.c_int32), ("depth_w", ctypes.c_int32), ("depth_c", ctypes.c_int32),
        ("fill", ctypes.c_int32), ("mode", ctypes.c_int32), ("batch_size", ctypes.c_int32),
        ("depth_map_blur", ctypes.c_int32), ("depth_blur_vert_smooth", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("divergence", ctypes.c_double), ("separation", ctypes.c_double), ("stereo_balance", ctypes.c_double),
        ("convergence_point", ctypes.c_double), ("stereo_offset_exponent", ctypes.c_double),
        ("depth_blur_strength", ctypes.c_double), ("depth_blur_edge_threshold", ctypes.c_double),
        ("depth_blur_falloff", ctypes.c_double),
    ]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"comfystereo_amd native call failed ({code}): {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run `python __graft_entry__.py` or "
            "`make -C comfystereo_amd/csrc`).  comfystereo_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, c_int, c_double, c_size = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    pp = ctypes.POINTER(Params)
    ip = ctypes.POINTER(ctypes.c_int)
    L.cs_version.restype = c_int
    L.cs_version.argtypes = []
    if L.cs_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.cs_version()}, these bindings were written for {ABI_VERSION}: "
                          "rebuild it (`make -C comfystereo_amd/csrc`)")
    L.cs_last_error.restype = ctypes.c_char_p
    L.cs_last_error.argtypes = []
    L.cs_max_width_mode.restype = c_int
    L.cs_max_width_mode.argtypes = [c_int, c_int]
    L.cs_max_width_params.restype = c_int
    L.cs_max_width_params.argtypes = [ctypes.POINTER(Params)]
    L.cs_max_width.restype = c_int
    L.cs_max_width.argtypes = [c_int]
    L.cs_output_shape.restype = c_int
    L.cs_output_shape.argtypes = [pp, ip, ip, ip, ip]
    L.cs_workspace_bytes.restype = c_size
    L.cs_workspace_bytes.argtypes = [pp]
    L.cs_generate.restype = c_int
    L.cs_generate.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, c_size, vp]
    L.cs_asd_workspace_bytes.restype = c_size
    L.cs_asd_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.cs_asd_workspace_bytes_for.restype = c_size
    L.cs_asd_workspace_bytes_for.argtypes = [c_int, c_int, c_int, c_int]
    L.cs_apply_stereo_divergence2.restype = c_int
    L.cs_apply_stereo_divergence2.argtypes = [vp, vp, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_double, c_int,
                                              vp, vp, c_size, vp]
    L.cs_apply_stereo_divergence.restype = c_int
    L.cs_apply_stereo_divergence.argtypes = [vp, vp, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_double,
                                             vp, vp, c_size, vp]
    L.cs_blur_workspace_bytes.restype = c_size
    L.cs_blur_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.cs_directional_blur.restype = c_int
    L.cs_directional_blur.argtypes = [vp, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int, vp, vp, vp,
                                      c_size, vp]
    L.cs_blur_scipy_workspace_bytes.restype = c_size
    L.cs_blur_scipy_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.cs_directional_blur_scipy.restype = c_int
    L.cs_directional_blur_scipy.argtypes = [vp, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int, vp, vp, vp, c_size, vp]
    L.cs_warp_workspace_bytes.restype = c_size
    L.cs_warp_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.cs_warp_mesh_workspace_bytes.restype = c_size
    L.cs_warp_mesh_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.cs_forward_warp_mesh.restype = c_int
    L.cs_forward_warp_mesh.argtypes = [vp, vp, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_double, vp, vp, vp, c_size, vp]
    L.cs_forward_warp.restype = c_int
    L.cs_forward_warp.argtypes = [vp, vp, c_int, c_int, c_int, c_double, c_double, c_double, c_double, vp, vp, vp, c_size, vp]
    L.cs_forward_warp2.restype = c_int
    L.cs_forward_warp2.argtypes = [vp, vp, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_int, vp, vp, vp, c_size, vp]
    L.cs_pack_u8.restype = c_int
    L.cs_pack_u8.argtypes = [vp, vp, c_size, c_int, c_int, vp]
    L.cs_host_expand_u8.restype = c_int
    L.cs_host_expand_u8.argtypes = [vp, vp, c_size, c_int, c_int, c_int]
    L.cs_host_copy.restype = c_int
    L.cs_host_copy.argtypes = [vp, vp, c_size, c_int]
    L.cs_take_f32.restype = c_int
    L.cs_take_f32.argtypes = [vp, vp, c_size, c_int, vp]
    L.cs_host_replicate_f32.restype = c_int
    L.cs_host_replicate_f32.argtypes = [vp, vp, c_size, c_int, c_int]
    L.cs_stereo_shift_workspace_bytes.restype = c_size
    L.cs_stereo_shift_workspace_bytes.argtypes = []
    L.cs_stereo_shift.restype = c_int
    L.cs_stereo_shift.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_double, c_int, c_double, vp, vp, c_size, vp]
    L.cs_expand_u8.restype = c_int
    L.cs_expand_u8.argtypes = [vp, vp, c_size, vp]
    L.cs_profile.restype = c_int
    L.cs_profile.argtypes = [c_int]
    L.cs_profile_read.restype = c_int
    L.cs_profile_read.argtypes = [ctypes.POINTER(c_double), ip]
    L.cs_profile_tiles.restype = c_int
    L.cs_profile_tiles.argtypes = [ctypes.POINTER(c_double)]
    L.cs_debug_set.restype = c_int
    L.cs_debug_set.argtypes = [c_int, c_int]
    L.cs_test_powf.restype = c_int
    L.cs_test_powf.argtypes = [vp, ctypes.c_float, vp, c_size, vp]
    L.cs_test_exp.restype = c_int
    L.cs_test_exp.argtypes = [vp, vp, c_size, vp]
    L.cs_test_edge_threshold.restype = ctypes.c_float
    L.cs_test_edge_threshold.argtypes = [ctypes.c_float]
    _lib = L
    return L


def debug_set(key, value):
    """cs_debug_set: development switch `key` (see DEBUG) := value.  Process-wide; reset it to 0 afterwards."""
    check(lib().cs_debug_set(DEBUG[key], int(value)))


def check(rc):
    if rc != CS_OK:
        raise NativeError(rc, lib().cs_last_error().decode("utf-8", "replace"))
