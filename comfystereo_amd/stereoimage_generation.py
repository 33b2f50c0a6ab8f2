"""Drop-in for the two module functions GenerateStereo.py calls in the reference
(`from . import stereoimage_generation as sig`, reference GenerateStereo.py:25,151,222):

    create_stereoimages      reference stereoimage_generation.py:1422-1574
    create_stereoimages_gpu  reference stereoimage_generation.py:1005-1128

Same names, argument order, defaults, return types and error behaviour; the arithmetic runs in the
HIP kernels behind the C ABI (there is no CPU fallback: without a GPU these functions raise).
Extras kept for callers of the reference's building blocks: apply_stereo_divergence (:1576-1620),
directional_motion_blur_gpu (:1171-1251), forward_warp_gpu (:277-450), forward_warp_mesh (:453-689).
"""
import numpy as np
import torch
from PIL import Image

from . import engine

_CPU_FILLS = ('none', 'naive', 'naive_interpolating', 'polylines_soft', 'polylines_sharp', 'inverse', 'hybrid_edge',
              'none_post', 'inverse_post', 'hybrid_edge_plus')  # the last three: no UI string reaches them (:1605-1610)
_MODES = ('left-right', 'right-left', 'top-bottom', 'bottom-top', 'red-cyan-anaglyph', 'left-only', 'only-right',
          'cyan-red-reverseanaglyph')


# The reference's import-time switch (:18-24): True makes create_stereoimages_gpu warp through forward_warp_mesh, the
# mesh-quality rasteriser, instead of forward_warp_gpu (:1068-1071).  Both are HIP kernels here; False (the default) is the
# parity-pinned path.  engine.MESH_WARP is the same switch for the node.
MODERNGL_AVAILABLE = False


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("comfystereo_amd needs an MI355X (PyTorch-ROCm `cuda` device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_u8(x):
    """float32 outputs of the CPU-technique path are k/255 exactly -> recover k."""
    return torch.round(x * 255.0).to(torch.uint8)


def create_stereoimages(original_image, depthmap, divergence, separation=0.0, modes=None,
                        stereo_balance=0.0, stereo_offset_exponent=1.0, fill_technique='polylines_sharp',
                        depth_blur_strength=0.0, depth_blur_edge_threshold=6.0,
                        direction_aware_depth_blur=False, return_modified_depth=True, convergence_point=0.5,
                        depth_blur_falloff=1.0, depth_blur_vert_smooth=0):
    """One frame, CPU-technique semantics: returns PIL images like the reference.

    original_image: torch [3,H,W] (or [H,W,3]) float 0..1;  depthmap: torch [H,W] (or [1,H,W]).
    -> (list[PIL RGB], PIL L left, PIL L right) | (list, PIL L) | list       (reference :1564-1574)
    """
    if modes is None:
        modes = ['left-right']
    if not isinstance(modes, list):
        modes = [modes]
    if len(modes) == 0:
        return []
    if not (isinstance(depthmap, torch.Tensor) and isinstance(original_image, torch.Tensor)):
        return _create_stereoimages_numpy(original_image, depthmap, divergence, separation, modes, stereo_balance,
                                          stereo_offset_exponent, fill_technique, depth_blur_strength,
                                          direction_aware_depth_blur, return_modified_depth, convergence_point,
                                          depth_blur_edge_threshold, depth_blur_falloff, depth_blur_vert_smooth)
    for m in modes:
        if m not in _MODES:
            raise Exception('Unknown mode')
    dev = _device()
    depth = depthmap.to(dev, torch.float32)
    if depth.dim() == 3:
        depth = depth.squeeze()
    img = original_image.to(dev, torch.float32)
    if img.dim() == 3 and img.shape[0] == 3:
        img = img.permute(1, 2, 0)
    h, w = depth.shape
    assert tuple(img.shape[:2]) == (h, w), 'Depthmap and the image must have the same size'
    img = img.contiguous()[None]
    dep = depth.contiguous()[None, :, :, None]
    # an unknown technique falls through the reference's dispatch and returns the source image for both eyes (:1620)
    fill = fill_technique if fill_technique in _CPU_FILLS else 'none'
    div = divergence if fill_technique in _CPU_FILLS else 0.0
    results, left8, right8 = [], None, None
    for m in modes:
        stereo, dl, dr, _ = engine.generate(img, dep, div, separation, m, stereo_balance, convergence_point,
                                            stereo_offset_exponent, fill, depth_blur_edge_threshold, depth_blur_strength,
                                            direction_aware_depth_blur, depth_blur_falloff, depth_blur_vert_smooth, 1)
        results.append(_to_u8(stereo[0]).cpu().numpy())
        left8, right8 = _to_u8(dl[0, :, :, 0]).cpu().numpy(), _to_u8(dr[0, :, :, 0]).cpu().numpy()
    stereo_images = [Image.fromarray(r) for r in results]
    if return_modified_depth:
        if direction_aware_depth_blur:
            return stereo_images, Image.fromarray(left8), Image.fromarray(right8)
        return stereo_images, Image.fromarray(left8)
    return stereo_images


def _create_stereoimages_numpy(original_image, depthmap, divergence, separation, modes, stereo_balance,
                               stereo_offset_exponent, fill_technique, depth_blur_strength, direction_aware_depth_blur,
                               return_modified_depth, convergence_point, depth_blur_edge_threshold=6.0, depth_blur_falloff=1.0,
                               depth_blur_vert_smooth=0):
    """numpy / PIL inputs (reference :1486-1499, 1519-1574): the uint8 image is used as it is, the depth map as float32 without
    the x255 step of the tensor path, each eye is apply_stereo_divergence (HIP, cs_apply_stereo_divergence), the modified
    depth is clip(depth, 0, 255).astype(uint8).  With the depth blur ON the reference takes its scipy blur
    (`directional_motion_blur`: scipy.ndimage sobel + convolve1d in float64, :1346-1419, called with blur_mask_width =
    blur_strength, :1490-1493): cs_directional_blur_scipy (round 4), each eye warped with its own blurred map."""
    for m in modes:
        if m not in _MODES:
            raise Exception('Unknown mode')
    image = np.asarray(original_image)
    depth = np.asarray(depthmap).astype(np.float32)
    if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
        raise ValueError("numpy / PIL images must be uint8 [H, W, 3] (the reference indexes them as such)")
    assert depth.shape == image.shape[:2], 'Depthmap and the image must have the same size'   # (reference :1586)
    dev = _device()
    img_t = torch.from_numpy(np.array(image, dtype=np.uint8, order='C', copy=True)).to(dev)   # (PIL arrays are read-only)
    dep_t = torch.from_numpy(np.ascontiguousarray(depth)).to(dev)
    dep_l = dep_r = dep_t
    if direction_aware_depth_blur:   # (strength <= 0: the depth map itself for both eyes, :1374)
        # a strength that rounds to a box of 0 taps: scipy's convolve1d raises in the reference as well (:1402-1405).  Tested here,
        # not by re-labelling whatever the native call raises (a HIP failure or an out-of-memory error must stay what it is)
        if depth_blur_strength > 0 and int(round(depth_blur_strength)) < 1:
            raise RuntimeError("no filter weights given")
        dep_l, dep_r = engine.directional_blur_scipy(dep_t, depth_blur_strength, depth_blur_edge_threshold, depth_blur_strength,
                                                     depth_blur_falloff, depth_blur_vert_smooth)
    left_div, right_div = divergence * (1 + stereo_balance), divergence * (1 - stereo_balance)

    def eye(dep, div_signed, sep_signed, enabled):
        if not enabled or fill_technique not in _CPU_FILLS:   # (< 0.001: the source image, :1536; unknown technique: :1620)
            return img_t
        return engine.apply_stereo_divergence(img_t, dep, div_signed, sep_signed, stereo_offset_exponent, fill_technique,
                                              convergence_point)

    left = eye(dep_l, +1 * left_div, -1 * separation, not (left_div < 0.001))
    right = eye(dep_r, -1 * right_div, separation, not (right_div < 0.001))

    def anaglyph(a, b):   # overlap_red_cyan (:1996-2010): R from a, G and B from b
        return torch.cat([a[..., :1], b[..., 1:]], dim=-1)

    results = []
    for m in modes:
        r = {'left-right': lambda: torch.cat([left, right], 1), 'right-left': lambda: torch.cat([right, left], 1),
             'top-bottom': lambda: torch.cat([left, right], 0), 'bottom-top': lambda: torch.cat([right, left], 0),
             'red-cyan-anaglyph': lambda: anaglyph(left, right), 'left-only': lambda: left, 'only-right': lambda: right,
             'cyan-red-reverseanaglyph': lambda: anaglyph(right, left)}[m]()
        results.append(Image.fromarray(r.contiguous().cpu().numpy()))
    if not return_modified_depth:
        return results
    # np.clip(depth, 0, 255).astype(np.uint8) (:1567-1572): truncation of the clamped float
    if direction_aware_depth_blur:
        return (results, Image.fromarray(torch.clamp(dep_l, 0, 255).to(torch.uint8).cpu().numpy()),
                Image.fromarray(torch.clamp(dep_r, 0, 255).to(torch.uint8).cpu().numpy()))
    return results, Image.fromarray(torch.clamp(dep_t, 0, 255).to(torch.uint8).cpu().numpy())


def create_stereoimages_gpu(image_tensor, depth_tensor, divergence, separation=0.0, modes=None,
                            stereo_balance=0.0, stereo_offset_exponent=1.0, convergence_point=0.5,
                            depth_blur_strength=0.0, depth_blur_edge_threshold=6.0,
                            direction_aware_depth_blur=False, depth_blur_falloff=1.0,
                            depth_blur_vert_smooth=0):
    """Batched 'gpu_warp' driver: image [B,C,H,W] 0..1, depth [B,H,W] ->
    (list[Tensor[B,C,H',W']], left_depth [B,H,W], right_depth [B,H,W], BoolTensor[B,H,W]) on the device."""
    if modes is None:
        modes = ['left-right']
    if not isinstance(modes, list):
        modes = [modes]
    if len(modes) == 0:
        return [], None, None, None
    for m in modes:
        if m not in _MODES:
            raise ValueError(f'Unknown mode: {m}')
    dev = _device()
    img = image_tensor.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()
    dep = depth_tensor.to(dev, torch.float32).contiguous()[..., None]
    b, h, w, _ = img.shape
    results, lo, ro, mask = [], None, None, None
    for m in modes:
        p = engine.make_params(b, h, w, h, w, 1, 'gpu_warp', m, divergence, separation, stereo_balance, convergence_point,
                               stereo_offset_exponent, direction_aware_depth_blur, depth_blur_strength,
                               depth_blur_edge_threshold, depth_blur_falloff, depth_blur_vert_smooth, b)
        p.flags |= 1  # module-level depth outputs are not clamped (the node clamps them afterwards)
        if MODERNGL_AVAILABLE:
            p.flags |= 4
        stereo, dl, dr, mk = engine.Plan(p, dev).run(img, dep)
        results.append(stereo.permute(0, 3, 1, 2))
        lo, ro, mask = dl[..., 0], dr[..., 0], mk > 0.5
    return results, lo, ro, mask


def apply_stereo_divergence(original_image, depth, divergence, separation, stereo_offset_exponent, fill_technique,
                            convergence_point=0.5):
    """uint8 [H,W,3] image + float32 [H,W] depth (numpy or torch) -> uint8 [H,W,3] numpy (reference :1576-1620)."""
    dev = _device()
    img = torch.as_tensor(np.asarray(original_image) if not isinstance(original_image, torch.Tensor) else original_image)
    dep = torch.as_tensor(np.asarray(depth) if not isinstance(depth, torch.Tensor) else depth)
    assert tuple(img.shape[:2]) == tuple(dep.shape), 'Depthmap and the image must have the same size'
    if fill_technique not in _CPU_FILLS:
        return img.cpu().numpy()  # reference fallback (:1620)
    out = engine.apply_stereo_divergence(img.to(dev, torch.uint8), dep.to(dev, torch.float32), divergence, separation,
                                         stereo_offset_exponent, fill_technique, convergence_point)
    return out.cpu().numpy()


def directional_motion_blur_gpu(depth_tensor, blur_strength, edge_threshold, blur_mask_width=5,
                                falloff_exponent=1.0, vert_smooth_px=0):
    """[H,W] or [B,H,W] depth on the 0..255 scale -> (left, right) device tensors (reference :1171-1251)."""
    if blur_strength <= 0:
        return depth_tensor, depth_tensor
    return engine.directional_blur(depth_tensor.to(_device(), torch.float32), blur_strength, edge_threshold,
                                   falloff_exponent, vert_smooth_px, blur_mask_width=blur_mask_width)


def forward_warp_gpu(image_tensor, depth_tensor, divergence_px, separation_px, stereo_offset_exponent,
                     convergence_point=0.5, gradient_threshold=1.5, max_stretch=8):
    """reference :277-450 -> (warped [B,C,H,W], gap_mask bool [B,H,W]).  gradient_threshold (connectivity, :339-340) and
    max_stretch (scatter rounds, :365) are kernel parameters; the reference's own call sites use the defaults."""
    dev = _device()
    return engine.forward_warp(image_tensor.to(dev, torch.float32), depth_tensor.to(dev, torch.float32), divergence_px,
                               separation_px, stereo_offset_exponent, convergence_point, gradient_threshold, max_stretch)


def forward_warp_mesh(image_tensor, depth_tensor, divergence_px, separation_px, stereo_offset_exponent,
                      convergence_point=0.5, gradient_threshold=1.5, max_stretch=8):
    """reference :453-689 -> (warped [B,C,H,W], gap_mask bool [B,H,W]); `max_stretch` is unused there as well."""
    dev = _device()
    return engine.forward_warp_mesh(image_tensor.to(dev, torch.float32), depth_tensor.to(dev, torch.float32), divergence_px,
                                    separation_px, stereo_offset_exponent, convergence_point, gradient_threshold)
