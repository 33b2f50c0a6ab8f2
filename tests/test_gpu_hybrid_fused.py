"""hybrid_edge with the node outputs written by the splat kernel itself (cs_rowwarp.hip k_hybrid_splat_tile<true>) instead of a
streaming pass of its own (k_hybrid_out4, cs_debug_set(CS_DEBUG_HYBRID_UNFUSED, 1)): identical bits, and the oracle's
(reference stereoimage_generation.py:1622-1661 splat, :1745-1774 gap fill, GenerateStereo.py:355-361 mask)."""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle

pytestmark = pytest.mark.gpu
UI = "Imperfect fill - Hybrid Edge"


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(engine, img, depth, mode, div, balance=0.0, blur=False):
    out = engine.generate(cuda(img), cuda(depth), div, 0.0, mode, balance, 0.5, 2.0, "hybrid_edge", 20.0, 20.0, blur,
                          depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    return [o.cpu().numpy() for o in out]


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("mode", ["left-right", "right-left", "top-bottom", "bottom-top"])
@pytest.mark.parametrize("h,w,div", [(40, 1284, 6.0), (37, 250, 9.0), (24, 1601, 3.0)])
def test_fused_equals_streaming_pass_and_oracle(engine, dev_switch, mode, h, w, div):
    n = 2
    img = synth.image_f32(n, h, w, seed=11)
    img[0, :4, :9] = 0.0   # black source pixels: mask 1 although touched
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    fused = run(engine, img, depth, mode, div)
    dev_switch("hybrid_unfused", 1)
    plain = run(engine, img, depth, mode, div)
    for k, (a, b) in enumerate(zip(fused, plain)):
        assert np.array_equal(bits(a), bits(b)), (mode, h, w, k)
    want = node_oracle.generate(img, depth, div, 0.0, mode, 0.0, 0.5, 2.0, UI, 20.0, 20.0, False,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(fused[k], want[k]), (mode, h, w, k)


@pytest.mark.parametrize("balance", [1.0, -1.0])
def test_one_eye_is_the_source_image(engine, dev_switch, balance):
    """stereo_balance +-1: one eye's divergence is 0 (< 0.001: the source image, quirk Q10) -- the splat kernel writes that
    eye's outputs too."""
    n, h, w = 2, 33, 772
    img = synth.image_f32(n, h, w, seed=12)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    fused = run(engine, img, depth, "left-right", 5.0, balance=balance, blur=True)
    dev_switch("hybrid_unfused", 1)
    plain = run(engine, img, depth, "left-right", 5.0, balance=balance, blur=True)
    for k, (a, b) in enumerate(zip(fused, plain)):
        assert np.array_equal(bits(a), bits(b)), (balance, k)
    want = node_oracle.generate(img, depth, 5.0, 0.0, "left-right", balance, 0.5, 2.0, UI, 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(fused[k], want[k]), (balance, k)


def test_uint8_stereoscope(engine, dev_switch):
    """engine.Plan(stereo_u8=True): the compact form of the stereoscope the host pipeline and the sharded job use."""
    n, h, w = 2, 40, 644
    img = synth.image_f32(n, h, w, seed=13)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    p = engine.make_params(n, h, w, h, w, 3, "hybrid_edge", "left-right", 6.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 4)
    plan = engine.Plan(p, torch.device("cuda"), stereo_u8=True)
    fused = [o.clone().cpu().numpy() for o in plan.run(cuda(img), cuda(depth))]
    dev_switch("hybrid_unfused", 1)
    plain = [o.clone().cpu().numpy() for o in plan.run(cuda(img), cuda(depth))]
    for k, (a, b) in enumerate(zip(fused, plain)):
        assert np.array_equal(bits(a), bits(b)), k
    assert fused[0].dtype == np.uint8
