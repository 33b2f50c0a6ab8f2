"""Node-level CPU oracle (numpy glue around oracle/stereo_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Restates the reference's drivers in the D32 dialect:
  create_stereoimages      reference stereoimage_generation.py:1422-1574
  create_stereoimages_gpu  reference stereoimage_generation.py:1005-1128 (with forward_warp_gpu, :277-450)
  generate                 reference GenerateStereo.py:79-353 (+ generate_mask :355-361, convertResult :365-378)
Pinned by tests/golden/node_generate.npz (outputs of the imported reference's StereoImageNode.generate).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from . import oracle

F32 = np.float32

FILL_KEYS = {
    'GPU Warp (Fast)': 'gpu_warp', 'No fill': 'none', 'No fill - Reverse projection': 'inverse',
    'Imperfect fill - Hybrid Edge': 'hybrid_edge', 'Fill - Naive': 'naive',
    'Fill - Naive interpolating': 'naive_interpolating', 'Fill - Polylines Soft': 'polylines_soft',
    'Fill - Polylines Sharp': 'polylines_sharp',
    # mapped by the reference (GenerateStereo.py:97-99) although its combo list leaves them out (:56-57)
    'Fill - Post-fill': 'none_post', 'Fill - Reverse projection with Post-fill': 'inverse_post',
    'Fill - Hybrid Edge with fill': 'hybrid_edge_plus',
}


def f32_to_u8_wrap(a):
    """numpy float32 -> uint8 astype on x86-64 (cvttss2si then low byte), quirk Q7."""
    a = np.asarray(a, dtype=F32)
    bad = ~((a > F32(-2147483904.0)) & (a < F32(2147483648.0)))
    i = np.where(bad, 0, a).astype(np.int64)
    i = np.where(bad, -2147483648, i)
    return (i & 0xFF).astype(np.uint8)


def gray(depth_nhwc):
    """GenerateStereo.py:134-139 / 206-209: float32 (c0*R + c1*G) + c2*B, separate roundings."""
    d = np.asarray(depth_nhwc, dtype=F32)
    c = d.shape[-1]
    if c == 3:
        return (F32(0.2989) * d[..., 0] + F32(0.5870) * d[..., 1]) + F32(0.1140) * d[..., 2]
    if c == 1:
        return d[..., 0]
    return d[..., 0]


def resize_bilinear(depth_bhw, size):
    """F.interpolate(mode='bilinear', align_corners=False) -- delegated to torch CPU (what the reference calls)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(depth_bhw, dtype=F32)).unsqueeze(1)
    return torch.nn.functional.interpolate(t, size=tuple(size), mode='bilinear', align_corners=False).squeeze(1).numpy()


def assemble(left, right, mode, chan_axis, w_axis, h_axis):
    if mode == 'left-right':
        return np.concatenate([left, right], axis=w_axis)
    if mode == 'right-left':
        return np.concatenate([right, left], axis=w_axis)
    if mode == 'top-bottom':
        return np.concatenate([left, right], axis=h_axis)
    if mode == 'bottom-top':
        return np.concatenate([right, left], axis=h_axis)
    if mode in ('red-cyan-anaglyph', 'cyan-red-reverseanaglyph'):
        a, b = (left, right) if mode == 'red-cyan-anaglyph' else (right, left)
        out = np.array(b, copy=True)
        idx = [slice(None)] * out.ndim
        idx[chan_axis] = 0
        out[tuple(idx)] = a[tuple(idx)]
        return out
    if mode == 'left-only':
        return left
    if mode == 'only-right':
        return right
    raise ValueError('Unknown mode')


def create_stereoimages(img_chw, depth_hw, divergence, separation=0.0, modes=None, stereo_balance=0.0,
                        stereo_offset_exponent=1.0, fill_technique='polylines_sharp', depth_blur_strength=0.0,
                        depth_blur_edge_threshold=6.0, direction_aware_depth_blur=False, convergence_point=0.5,
                        depth_blur_falloff=1.0, depth_blur_vert_smooth=0):
    """-> (list of uint8 [H',W',3], mod_left uint8 [H,W], mod_right uint8 [H,W])."""
    modes = ['left-right'] if modes is None else (modes if isinstance(modes, list) else [modes])
    depth = np.asarray(depth_hw, dtype=F32)
    if depth.max() <= 1.0:
        depth = depth * F32(255.0)
    if direction_aware_depth_blur:
        left_d, right_d = oracle.blur(depth, depth_blur_strength, depth_blur_edge_threshold, depth_blur_falloff,
                                      depth_blur_vert_smooth)
    else:
        left_d = right_d = depth
    img = np.asarray(img_chw, dtype=F32).transpose(1, 2, 0)
    img8 = np.clip(img * F32(255), 0, 255).astype(np.uint8)
    mod_left, mod_right = f32_to_u8_wrap(left_d * F32(255)), f32_to_u8_wrap(right_d * F32(255))
    left_div = divergence * (1 + stereo_balance)
    right_div = divergence * (1 - stereo_balance)
    left_eye = img8 if left_div < 0.001 else oracle.apply_stereo_divergence(
        img8, left_d, +1 * left_div, -1 * separation, stereo_offset_exponent, fill_technique, convergence_point)
    right_eye = img8 if right_div < 0.001 else oracle.apply_stereo_divergence(
        img8, right_d, -1 * right_div, separation, stereo_offset_exponent, fill_technique, convergence_point)
    return [assemble(left_eye, right_eye, m, 2, 1, 0) for m in modes], mod_left, mod_right


def create_stereoimages_gpu(img_bchw, depth_bhw, divergence, separation=0.0, modes=None, stereo_balance=0.0,
                            stereo_offset_exponent=1.0, convergence_point=0.5, depth_blur_strength=0.0,
                            depth_blur_edge_threshold=6.0, direction_aware_depth_blur=False, depth_blur_falloff=1.0,
                            depth_blur_vert_smooth=0, mesh=False):
    """-> (list of float32 [B,3,H',W'], left_depth [B,H,W], right_depth [B,H,W], mask bool [B,H,W]).
    mesh: warp_fn = forward_warp_mesh (the reference's choice when moderngl is importable, :1068-1071)."""
    modes = ['left-right'] if modes is None else (modes if isinstance(modes, list) else [modes])
    img = np.ascontiguousarray(img_bchw, dtype=F32)
    depth = np.ascontiguousarray(depth_bhw, dtype=F32)
    B, _, H, W = img.shape
    if depth.max() <= 1.0:
        depth = depth * F32(255.0)
    if direction_aware_depth_blur and depth_blur_strength > 0:
        left_d, right_d = oracle.blur(depth, depth_blur_strength, depth_blur_edge_threshold, depth_blur_falloff,
                                      depth_blur_vert_smooth)
    else:
        left_d = right_d = depth
    left_div = divergence * (1 + stereo_balance)
    right_div = divergence * (1 - stereo_balance)
    left_px, right_px, sep_px = (left_div / 100.0) * W, (right_div / 100.0) * W, (separation / 100.0) * W
    lmask = rmask = np.zeros((B, H, W), dtype=bool)
    left_eye = right_eye = img
    warp_fn = oracle.forward_warp_mesh if mesh else oracle.forward_warp_gpu
    if not left_div < 0.001:
        left_eye, lmask = warp_fn(img, left_d, +left_px, -sep_px, stereo_offset_exponent, convergence_point)
    if not right_div < 0.001:
        right_eye, rmask = warp_fn(img, right_d, -right_px, sep_px, stereo_offset_exponent, convergence_point)
    results = [assemble(left_eye, right_eye, m, 1, 3, 2) for m in modes]
    lo = left_d / F32(255.0) if left_d.max() > 1.0 else left_d
    ro = right_d / F32(255.0) if right_d.max() > 1.0 else right_d
    return results, lo, ro, lmask | rmask


def generate(image, depth_map, divergence, separation, modes, stereo_balance, convergence_point, stereo_offset_exponent,
             fill_technique, depth_blur_edge_threshold, depth_blur_strength, depth_map_blur, depth_blur_falloff=1.0,
             depth_blur_vert_smooth=0, batch_size=4, mesh=False):
    """-> (stereoscope [N,H',W',3], depth_left [N,H,W,3], depth_right [N,H,W,3], mask [N,H',W' | H,W]) float32."""
    image = np.asarray(image, dtype=F32)
    depth_map = np.asarray(depth_map, dtype=F32)
    fill = FILL_KEYS.get(fill_technique, 'gpu_warp')
    n, H, W = image.shape[0], image.shape[1], image.shape[2]
    stereo, dls, drs, masks = [], [], [], []
    if fill == 'gpu_warp':
        bs = min(batch_size, n)
        for b0 in range(0, n, bs):
            img = image[b0:b0 + bs].transpose(0, 3, 1, 2)
            dm = gray(depth_map[b0:b0 + bs])
            if dm.shape[1:] != (H, W):
                dm = resize_bilinear(dm, (H, W))
            res, lo, ro, m = create_stereoimages_gpu(img, dm, divergence, separation, [modes], stereo_balance,
                                                     stereo_offset_exponent, convergence_point, depth_blur_strength,
                                                     depth_blur_edge_threshold, depth_map_blur,
                                                     depth_blur_falloff=depth_blur_falloff,
                                                     depth_blur_vert_smooth=depth_blur_vert_smooth, mesh=mesh)
            stereo.append(res[0].transpose(0, 2, 3, 1))
            dls.append(np.repeat(np.clip(lo, 0, 1)[..., None], 3, -1))
            drs.append(np.repeat(np.clip(ro, 0, 1)[..., None], 3, -1))
            masks.append(m.astype(F32))
    else:
        for i in range(n):
            dm = gray(depth_map[i:i + 1])
            if dm.shape[1:] != (H, W):
                dm = resize_bilinear(dm, (H, W))
            res, ml, mr = create_stereoimages(image[i].transpose(2, 0, 1), dm[0], divergence, separation, [modes],
                                              stereo_balance, stereo_offset_exponent, fill, depth_blur_strength,
                                              depth_blur_edge_threshold, depth_map_blur,
                                              convergence_point=convergence_point, depth_blur_falloff=depth_blur_falloff,
                                              depth_blur_vert_smooth=depth_blur_vert_smooth)
            r8 = res[0]
            stereo.append((r8.astype(F32) / F32(255.0))[None])
            dls.append(np.repeat((ml.astype(F32) / F32(255.0))[..., None], 3, -1)[None])
            drs.append(np.repeat((mr.astype(F32) / F32(255.0))[..., None], 3, -1)[None])
            masks.append(((r8.astype(np.int64).sum(-1) == 0).astype(np.uint8) * 255).astype(F32)[None] / F32(255.0))
    return (np.concatenate(stereo), np.concatenate(dls), np.concatenate(drs), np.concatenate(masks))
