"""Device-side engine: thin torch-tensor wrappers over the C ABI (libcomfystereo_hip.so).

Everything here takes and returns tensors that live on the MI355X (`cuda` device in PyTorch-ROCm);
PyTorch only provides the memory and the stream -- the arithmetic is in the HIP kernels.
"""
import ctypes

import torch

from . import _native
from ._native import FILL, MODE, Params


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _dev(t):
    if not t.is_cuda:
        raise ValueError("comfystereo_amd engine calls need device-resident tensors (there is no CPU path)")
    return t


# The reference picks its gpu_warp implementation at import time: forward_warp_mesh when `moderngl` is importable,
# forward_warp_gpu otherwise (stereoimage_generation.py:18-24, :1068-1071).  Here both are HIP kernels and the choice is
# this switch (cs_params.flags bit 2): False = forward_warp_gpu semantics (the parity-pinned default), True = mesh quality.
MESH_WARP = False
# Arithmetic dialect of the CPU techniques (cs_params.flags bits 3 / 4): "D32" = the reference as it runs WITHOUT numba
# (float32 disparities, wrapping uint8 pixel sums) -- what the goldens pin; "D64" = the typing numba gives the same
# source lines (float64 disparities, int64 sums), available for none / naive /
# naive_interpolating / inverse.
DIALECT = "D32"


def make_params(n, h, w, depth_h, depth_w, depth_c, fill, mode, divergence, separation, stereo_balance,
                convergence_point, stereo_offset_exponent, depth_map_blur, depth_blur_strength,
                depth_blur_edge_threshold, depth_blur_falloff, depth_blur_vert_smooth, batch_size):
    if mode not in MODE:
        raise ValueError(f"Unknown mode: {mode}")
    p = Params()
    p.n, p.h, p.w, p.depth_h, p.depth_w, p.depth_c = n, h, w, depth_h, depth_w, depth_c
    p.fill, p.mode, p.batch_size = FILL[fill], MODE[mode], int(batch_size)
    p.depth_map_blur, p.depth_blur_vert_smooth = int(bool(depth_map_blur)), int(depth_blur_vert_smooth)
    p.divergence, p.separation, p.stereo_balance = float(divergence), float(separation), float(stereo_balance)
    p.convergence_point, p.stereo_offset_exponent = float(convergence_point), float(stereo_offset_exponent)
    p.depth_blur_strength, p.depth_blur_edge_threshold = float(depth_blur_strength), float(depth_blur_edge_threshold)
    p.depth_blur_falloff = float(depth_blur_falloff)
    if MESH_WARP and fill == 'gpu_warp':
        p.flags |= 4
    if DIALECT != "D32":
        p.flags |= DIALECTS[DIALECT] << 3   # (cs_generate refuses techniques without a D64 instantiation)
    return p


def output_shape(p):
    L = _native.lib()
    oh, ow, mh, mw = (ctypes.c_int() for _ in range(4))
    _native.check(L.cs_output_shape(ctypes.byref(p), ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(mh), ctypes.byref(mw)))
    return oh.value, ow.value, mh.value, mw.value


class Plan:
    """Pre-allocated outputs + workspace for repeated calls of one configuration (what bench.py times)."""

    def __init__(self, p, device, stereo_u8=False, tie_pool_bytes=0):
        """stereo_u8: the stereoscope is produced as its uint8 codes k (value k/255, CPU techniques only) --
        the compact form frame shards are all-gathered in; expand with `expand_u8`.
        tie_pool_bytes: workspace beyond cs_workspace_bytes; the polylines techniques add it to the pool their order-dependent
        rows export their stretches through (4 KB per image row by default: enough for saturated depth maps; a batch whose
        every row is ONE stretch -- depth noise, blur off -- needs 6 B per pixel and eye to keep all rows off the in-row replay)."""
        L = _native.lib()
        self.p = p
        if stereo_u8:
            p.flags |= 2
        oh, ow, mh, mw = output_shape(p)
        f32 = dict(dtype=torch.float32, device=device)
        self.stereo = torch.empty((p.n, oh, ow, 3), dtype=torch.uint8 if stereo_u8 else torch.float32, device=device)
        self.depth_l = torch.empty((p.n, p.h, p.w, 3), **f32)
        self.depth_r = torch.empty((p.n, p.h, p.w, 3), **f32)
        self.mask = torch.empty((p.n, mh, mw), **f32)
        self.ws_bytes = L.cs_workspace_bytes(ctypes.byref(p)) + int(tie_pool_bytes)
        self.ws = torch.empty((max(self.ws_bytes, 256),), dtype=torch.uint8, device=device)

    def run(self, image, depth):
        L = _native.lib()
        _native.check(L.cs_generate(ctypes.byref(self.p), _ptr(image), _ptr(depth), _ptr(self.stereo), _ptr(self.depth_l),
                                    _ptr(self.depth_r), _ptr(self.mask), _ptr(self.ws), self.ws_bytes, _stream()))
        return self.stereo, self.depth_l, self.depth_r, self.mask

    def stats(self):
        """Per-frame diagnostics words (see cs_common.h ST_*), e.g. polylines rows replayed sequentially."""
        return self.ws[: self.p.n * 64].view(torch.int32).view(self.p.n, 16).cpu()


def expand_u8(codes, out=None):
    """uint8 codes -> float32 k/255 on the device (true division, like the reference's np2tensor)."""
    L = _native.lib()
    codes = _dev(codes).contiguous()
    assert codes.dtype == torch.uint8
    if out is None:
        out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    _native.check(L.cs_expand_u8(_ptr(codes), _ptr(out), codes.numel(), _stream()))
    return out


def generate(image, depth_map, divergence, separation, modes, stereo_balance, convergence_point,
             stereo_offset_exponent, fill, depth_blur_edge_threshold, depth_blur_strength, depth_map_blur,
             depth_blur_falloff=1.0, depth_blur_vert_smooth=0, batch_size=4):
    """Fused batch path on device tensors: image [N,H,W,3], depth_map [N,H',W',C] float32 -> 4 device tensors."""
    image = _dev(image).contiguous().float()
    depth_map = _dev(depth_map).contiguous().float()
    n, h, w, c = image.shape
    if c != 3:
        raise ValueError("image must be [N,H,W,3]")
    p = make_params(n, h, w, depth_map.shape[1], depth_map.shape[2], depth_map.shape[3], fill, modes, divergence,
                    separation, stereo_balance, convergence_point, stereo_offset_exponent, depth_map_blur,
                    depth_blur_strength, depth_blur_edge_threshold, depth_blur_falloff, depth_blur_vert_smooth, batch_size)
    return Plan(p, image.device).run(image, depth_map)


DIALECTS = {"D32": 0, "D64": 3, "f64-disparity": 1, "int64-sum": 2}


def apply_stereo_divergence(image_u8, depth, divergence, separation, stereo_offset_exponent, fill, convergence_point=0.5,
                            dialect="D32"):
    """reference stereoimage_generation.py:1576-1620 for [N,H,W,3] uint8 + [N,H,W] float32 device tensors.
    dialect: "D32" = the reference without numba (the pinned contract); "D64" = numba's typing (float64 disparities,
    int64 pixel sums; for polylines_soft / polylines_sharp float64 point coordinates and the float64 sweep -- a literal
    one-lane replay per row, ~100x slower than D32); none / naive / naive_interpolating / inverse / polylines_* / hybrid_edge only."""
    L = _native.lib()
    image_u8 = _dev(image_u8).contiguous()
    depth = _dev(depth).contiguous().float()
    assert image_u8.dtype == torch.uint8
    squeeze = image_u8.dim() == 3
    if squeeze:
        image_u8, depth = image_u8[None], depth[None]
    n, h, w, _ = image_u8.shape
    out = torch.empty_like(image_u8)
    nb = L.cs_asd_workspace_bytes_for(n, h, w, FILL[fill])
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=image_u8.device)
    _native.check(L.cs_apply_stereo_divergence2(_ptr(image_u8), _ptr(depth), n, h, w, float(divergence), float(separation),
                                                float(stereo_offset_exponent), FILL[fill], float(convergence_point),
                                                DIALECTS[dialect], _ptr(out), _ptr(ws), nb, _stream()))
    return out[0] if squeeze else out


def directional_blur(depth, blur_strength, edge_threshold, falloff_exponent=1.0, vert_smooth_px=0, blur_mask_width=None):
    """reference stereoimage_generation.py:1171-1251 for a [N,H,W] (or [H,W]) float32 device tensor, 0..255 scale.
    blur_mask_width: reach of the blur weights from an edge (default: the blur strength, like the reference's callers)."""
    L = _native.lib()
    depth = _dev(depth).contiguous().float()
    shp = depth.shape
    d3 = depth.reshape((-1,) + tuple(shp[-2:]))
    n, h, w = d3.shape
    out_l, out_r = torch.empty_like(d3), torch.empty_like(d3)
    nb = L.cs_blur_workspace_bytes(n, h, w)
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=depth.device)
    mw = float(blur_strength if blur_mask_width is None else blur_mask_width)
    _native.check(L.cs_directional_blur(_ptr(d3), n, h, w, float(blur_strength), float(edge_threshold), mw,
                                        float(falloff_exponent), int(vert_smooth_px), _ptr(out_l), _ptr(out_r), _ptr(ws),
                                        nb, _stream()))
    return out_l.reshape(shp), out_r.reshape(shp)


def directional_blur_scipy(depth, blur_strength, edge_threshold, blur_mask_width=5, falloff_exponent=1.0, vert_smooth_px=0):
    """reference stereoimage_generation.py:1346-1419 (`directional_motion_blur`, the scipy blur of the numpy / PIL input path)
    for a [N,H,W] (or [H,W]) float32 device tensor, used as given (no 0..255 rescaling) -> (left, right)."""
    L = _native.lib()
    depth = _dev(depth).contiguous().float()
    if blur_strength <= 0:   # (:1374)
        return depth, depth
    shp = depth.shape
    d3 = depth.reshape((-1,) + tuple(shp[-2:]))
    n, h, w = d3.shape
    out_l, out_r = torch.empty_like(d3), torch.empty_like(d3)
    nb = L.cs_blur_scipy_workspace_bytes(n, h, w)
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=depth.device)
    _native.check(L.cs_directional_blur_scipy(_ptr(d3), n, h, w, float(blur_strength), float(edge_threshold), float(blur_mask_width),
                                              float(falloff_exponent), int(vert_smooth_px), _ptr(out_l), _ptr(out_r), _ptr(ws), nb, _stream()))
    return out_l.reshape(shp), out_r.reshape(shp)


def forward_warp(image, depth, divergence_px, separation_px, stereo_offset_exponent, convergence_point=0.5,
                 gradient_threshold=1.5, max_stretch=8):
    """reference stereoimage_generation.py:277-450: image [B,3,H,W], depth [B,H,W] -> (warped, gap mask bool)."""
    L = _native.lib()
    image = _dev(image).contiguous().float()
    depth = _dev(depth).contiguous().float()
    b, c, h, w = image.shape
    assert c == 3
    warped = torch.empty_like(image)
    mask = torch.empty((b, h, w), dtype=torch.uint8, device=image.device)
    nb = L.cs_warp_workspace_bytes(b, h, w)
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=image.device)
    _native.check(L.cs_forward_warp2(_ptr(image), _ptr(depth), b, h, w, float(divergence_px), float(separation_px),
                                     float(stereo_offset_exponent), float(convergence_point), float(gradient_threshold),
                                     int(max_stretch), _ptr(warped), _ptr(mask), _ptr(ws), nb, _stream()))
    return warped, mask.bool()


def forward_warp_mesh(image, depth, divergence_px, separation_px, stereo_offset_exponent, convergence_point=0.5,
                      gradient_threshold=1.5):
    """reference stereoimage_generation.py:453-689 (the mesh-quality warp): image [B,3,H,W], depth [B,H,W] ->
    (warped, gap mask bool)."""
    L = _native.lib()
    image = _dev(image).contiguous().float()
    depth = _dev(depth).contiguous().float()
    b, c, h, w = image.shape
    assert c == 3
    warped = torch.empty_like(image)
    mask = torch.empty((b, h, w), dtype=torch.uint8, device=image.device)
    nb = L.cs_warp_mesh_workspace_bytes(b, h, w)
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=image.device)
    _native.check(L.cs_forward_warp_mesh(_ptr(image), _ptr(depth), b, h, w, float(divergence_px), float(separation_px),
                                         float(stereo_offset_exponent), float(convergence_point), float(gradient_threshold),
                                         _ptr(warped), _ptr(mask), _ptr(ws), nb, _stream()))
    return warped, mask.bool()


def stereo_shift(input_images, depthmaps, scale_factor=8.0, shift_both=False, stereo_offset_exponent=1.0):
    """reference stereo_utils.py:15-88 for device tensors: input [B,C,H,W], depth [B,H,W] float32 -> [2B,C,H,W]."""
    L = _native.lib()
    x = _dev(input_images).contiguous().float()
    d = _dev(depthmaps).contiguous().float()
    b, c, h, w = x.shape
    assert tuple(d.shape) == (b, h, w)
    out = torch.empty((2 * b, c, h, w), dtype=torch.float32, device=x.device)
    nb = L.cs_stereo_shift_workspace_bytes()
    ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=x.device)
    _native.check(L.cs_stereo_shift(_ptr(x), _ptr(d), b, c, h, w, float(scale_factor), int(bool(shift_both)),
                                    float(stereo_offset_exponent), _ptr(out), _ptr(ws), nb, _stream()))
    return out


def test_powf(x, y):
    L = _native.lib()
    x = _dev(x).contiguous().float()
    out = torch.empty_like(x)
    _native.check(L.cs_test_powf(_ptr(x), float(y), _ptr(out), x.numel(), _stream()))
    return out


def test_exp(x):
    L = _native.lib()
    x = _dev(x).contiguous().double()
    out = torch.empty_like(x)
    _native.check(L.cs_test_exp(_ptr(x), _ptr(out), x.numel(), _stream()))
    return out
