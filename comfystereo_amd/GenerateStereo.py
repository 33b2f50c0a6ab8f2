"""`StereoImageNode` -- ComfyUI node with the reference's surface (reference GenerateStereo.py:46-80,
460-466): same INPUT_TYPES keys / defaults / ranges, RETURN_TYPES, RETURN_NAMES and FUNCTION.

`generate` hands the whole batch to the fused device path (one C-ABI call per chunk) instead of the
reference's per-frame Python loop.  CPU tensors (what ComfyUI passes) go through the pinned-memory staging
pipeline of host_pipeline.py and come back as CPU float32 tensors like the reference's; device tensors stay on
the device.
"""
import torch

from . import engine, host_pipeline

_IN_COMFYUI = True
try:  # ComfyUI's progress bar when running inside ComfyUI (reference GenerateStereo.py:27,110)
    from comfy.utils import ProgressBar
except Exception:  # noqa: BLE001 - standalone use
    _IN_COMFYUI = False

    class ProgressBar:
        def __init__(self, total):
            self.total, self.current = total, 0

        def update(self, k):
            self.current += k

FILL_TECHNIQUES = {  # the combo entries of the widget (reference GenerateStereo.py:52-61)
    'GPU Warp (Fast)': 'gpu_warp',
    'No fill': 'none',
    'No fill - Reverse projection': 'inverse',
    'Imperfect fill - Hybrid Edge': 'hybrid_edge',
    'Fill - Naive': 'naive',
    'Fill - Naive interpolating': 'naive_interpolating',
    'Fill - Polylines Soft': 'polylines_soft',
    'Fill - Polylines Sharp': 'polylines_sharp',
}
# generate() also maps the three strings the reference keeps out of its combo list (commented out at :56-57) but still
# translates at :97-99 -- an API workflow may pass them
FILL_TECHNIQUE_MAPPING = dict(FILL_TECHNIQUES, **{
    'Fill - Post-fill': 'none_post',
    'Fill - Reverse projection with Post-fill': 'inverse_post',
    'Fill - Hybrid Edge with fill': 'hybrid_edge_plus',
})

# frames handed to one cs_generate call: bounded by a byte budget, not by the widget's batch_size
# (288 GB of HBM3E: a chunk of 64 4K frames needs ~55 GB including outputs)
CHUNK_BYTES = 96 << 30


# OPT-IN warm-up.  Importing this module allocates nothing and starts no thread (like the reference's).  A deployment that wants
# the first `generate` of the process to find its page-locked staging buffers and device buffers cached names a shape
# (frames, height, width) -- here, or as COMFYSTEREO_PREWARM=FxHxW in ComfyUI's environment -- and a daemon thread allocates and
# releases them into PyTorch's caching allocators while ComfyUI is still loading its models (host_pipeline.prewarm: result
# tensors are pinned only while everything stays under host_pipeline.PINNED_POOL_BYTES, 8 GB by default; for 32 x 4K frames
# that is 4.5 GB of pinned staging and 12 GB of HBM).  Without it the first call pins its staging buffers itself, the second
# slot's on a helper thread (profiles/r05_host.txt: first-call numbers with and without).
PREWARM = None


def _prewarm_shape():
    import os
    spec = os.environ.get("COMFYSTEREO_PREWARM", "")
    if PREWARM:
        return tuple(PREWARM)
    try:
        f, h, w = (int(v) for v in spec.lower().split("x"))
        return (f, h, w) if min(f, h, w) > 0 else None
    except ValueError:
        return None


if _IN_COMFYUI and _prewarm_shape():
    host_pipeline.prewarm_async(*_prewarm_shape())


class StereoImageNode:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "image": ("IMAGE",),
                "depth_map": ("IMAGE",),
                "modes": (["left-right", "right-left", "top-bottom", "bottom-top", "red-cyan-anaglyph"],),
                "fill_technique": (list(FILL_TECHNIQUES), {"default": "GPU Warp (Fast)",
                                                           "tooltip": "How disoccluded areas are filled."}),
            },
            "optional": {
                "divergence": ("FLOAT", {"default": 4.5, "min": 0.05, "max": 15, "step": 0.01,
                                         "tooltip": "Strength of the 3D effect, percent of the image width."}),
                "separation": ("FLOAT", {"default": 0, "min": -5, "max": 5, "step": 0.01,
                                         "tooltip": "Extra horizontal shift between the eyes, percent of the width."}),
                "stereo_balance": ("FLOAT", {"default": 0, "min": -0.95, "max": 0.95, "step": 0.05,
                                             "tooltip": "How the divergence is split between the two eyes."}),
                "convergence_point": ("FLOAT", {"default": 0.5, "min": 0.0, "max": 1.0, "step": 0.05,
                                                "tooltip": "Normalised depth that lands on the screen plane."}),
                "stereo_offset_exponent": ("FLOAT", {"default": 2, "min": 0.1, "max": 2, "step": 0.1,
                                                     "tooltip": "Exponent of the depth-to-offset curve."}),
                "depth_map_blur": ("BOOLEAN", {"default": True, "tooltip": "Direction-aware blur of the depth map."}),
                "depth_blur_edge_threshold": ("FLOAT", {"default": 20, "min": 0.1, "max": 60, "step": 0.1,
                                                        "tooltip": "Gradient threshold of the edge detector."}),
                "depth_blur_strength": ("FLOAT", {"default": 20, "min": 0.1, "max": 200, "step": 0.1,
                                                  "tooltip": "Width of the blur kernel in pixels."}),
                "depth_blur_falloff": ("FLOAT", {"default": 2.0, "min": 0.1, "max": 4.0, "step": 0.1,
                                                 "tooltip": "Falloff exponent of the blur weight away from edges."}),
                "depth_blur_vert_smooth": ("INT", {"default": 6, "min": 0, "max": 15, "step": 1,
                                                   "tooltip": "Vertical smoothing radius of the blur weights."}),
                "batch_size": ("INT", {"default": 12, "min": 1, "max": 64, "step": 1,
                                       "tooltip": "Frames per sub-batch (the reference's GPU sub-batch size)."}),
            }
        }

    RETURN_TYPES = ("IMAGE", "IMAGE", "IMAGE", "MASK")
    RETURN_NAMES = ("stereoscope", "blurred_depthmap_left", "blurred_depthmap_right", "no_fill_imperfect_mask")
    FUNCTION = "generate"

    def generate(self, image, depth_map, divergence, separation, modes,
                 stereo_balance, convergence_point, stereo_offset_exponent, fill_technique, depth_blur_edge_threshold,
                 depth_blur_strength, depth_map_blur, depth_blur_falloff=1.0, depth_blur_vert_smooth=0, batch_size=4):
        fill = FILL_TECHNIQUE_MAPPING.get(fill_technique, 'gpu_warp')  # unknown strings fall back like the reference (:102)
        if not torch.cuda.is_available():
            raise RuntimeError("comfystereo_amd needs an MI355X (PyTorch-ROCm `cuda` device); there is no CPU fallback")
        on_device = image.is_cuda
        total = len(image)
        pbar = ProgressBar(total)
        if not on_device:  # ComfyUI's case: CPU tensors in and out -> staged through pinned memory, chunks overlapped
            return host_pipeline.generate_host(image, depth_map, divergence, separation, modes, stereo_balance,
                                               convergence_point, stereo_offset_exponent, fill, depth_blur_edge_threshold,
                                               depth_blur_strength, depth_map_blur, depth_blur_falloff,
                                               depth_blur_vert_smooth, batch_size, progress=pbar.update)
        dev = image.device
        h, w = image.shape[1], image.shape[2]
        per_frame = 4 * h * w * (3 + depth_map.shape[3] + 6 + 6 + 2 + 5)
        chunk = max(1, min(total, CHUNK_BYTES // max(per_frame, 1)))
        if fill == 'gpu_warp':  # keep the reference's sub-batch boundaries (its 0..255 tests are per sub-batch)
            sub = min(batch_size, total)
            chunk = max(sub, (chunk // sub) * sub)
        outs = [[], [], [], []]
        for b0 in range(0, total, chunk):
            b1 = min(b0 + chunk, total)
            res = engine.generate(image[b0:b1].to(dev, torch.float32), depth_map[b0:b1].to(dev, torch.float32), divergence,
                                  separation, modes, stereo_balance, convergence_point, stereo_offset_exponent, fill,
                                  depth_blur_edge_threshold, depth_blur_strength, depth_map_blur, depth_blur_falloff,
                                  depth_blur_vert_smooth, batch_size)
            for k in range(4):
                outs[k].append(res[k])
            pbar.update(b1 - b0)
        return tuple(o[0] if len(o) == 1 else torch.cat(o, dim=0) for o in outs)


NODE_CLASS_MAPPINGS = {
    "StereoImageNode": StereoImageNode,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "StereoImageNode": "Stereo Image Node",
}
