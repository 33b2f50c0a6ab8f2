"""Drop-in for the one function of the reference's stereo_utils.py that lies on the depth-to-stereo path:
`stereo_shift_torch` (reference stereo_utils.py:15-88; called on diffusion latents by stereodiffusion_nodes.py:650, :664).
Same name, argument order, defaults and return shape; the shift runs in a HIP kernel behind the C ABI (cs_stereo_shift) --
no CPU fallback.  The attention-editing half of that file (BNAttention, ...) is diffusion plumbing and out of scope."""
import torch

from . import engine


def stereo_shift_torch(input_images: torch.Tensor, depthmaps: torch.Tensor, scale_factor: float = 8.0,
                       shift_both: bool = False, stereo_offset_exponent: float = 1.0) -> torch.Tensor:
    """input_images [B,C,H,W], depthmaps [B,H,W] -> [2B,C,H,W]: left views (the input unless shift_both) then right views."""
    if not torch.cuda.is_available():
        raise RuntimeError("comfystereo_amd needs an MI355X (PyTorch-ROCm `cuda` device); there is no CPU fallback")
    if depthmaps.dtype != torch.float32:
        # the reference normalises, applies pow and multiplies in the depth tensor's OWN dtype (stereo_utils.py:44-58); the
        # destination columns of a half / bfloat16 depth map round differently from a float32 one, and only float32 is pinned
        raise TypeError(f"stereo_shift_torch: depthmaps must be float32 (got {depthmaps.dtype}); parity with the reference is "
                        "pinned for float32 depth only")
    dev = input_images.device if input_images.is_cuda else torch.device("cuda", torch.cuda.current_device())
    out = engine.stereo_shift(input_images.to(dev, torch.float32), depthmaps.to(dev, torch.float32), scale_factor, shift_both,
                              stereo_offset_exponent)
    out = out.to(input_images.dtype)
    return out if input_images.is_cuda else out.to(input_images.device)
