"""One pass over an RGB depth map for the gray conversion AND the blur's edge bit rows (cs_blur.hip k_gray_edges).

The x255 decision (reference stereoimage_generation.py:1475 per frame, :1045 per gpu_warp sub-batch) needs the frame's
maximum, so the kernel builds the bit rows for both hypotheses and the blur picks a plane afterwards; the edge test
`clamp(|g| / den, 0, 1) > 0.5` (:1213-1222) is a comparison with a threshold computed on the host.  cs_debug_set(
CS_DEBUG_BLUR_NO_PRE_EDGES, 1) runs k_gray + k_blur_edges4 instead: both forms must give identical bits and equal the oracle.
"""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle

pytestmark = pytest.mark.gpu

UI = {"none": "No fill", "polylines_soft": "Fill - Polylines Soft", "hybrid_edge": "Imperfect fill - Hybrid Edge",
      "gpu_warp": "GPU Warp (Fast)"}


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(engine, img, depth, fill, mode="left-right", thr=20.0, strength=20.0, vert=6, div=6.0):
    out = engine.generate(cuda(img), cuda(depth), div, 0.0, mode, 0.1, 0.5, 2.0, fill, thr, strength, True,
                          depth_blur_falloff=2.0, depth_blur_vert_smooth=vert)
    return [o.cpu().numpy() for o in out]


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


# widths: a multiple of 256 (whole waves), 4 mod 256 (one live lane in the last wave), 252 mod 256 (lane 63 of the last wave
# is the frame's last group), below one wave; heights: not a multiple of the 32-row strips nor of the 4-row blocks
@pytest.mark.parametrize("h,w", [(70, 1284), (33, 512), (97, 252), (130, 1028), (32, 64), (5, 8)])
@pytest.mark.parametrize("rng", ["unit", "bytes", "mixed"])
def test_one_pass_equals_two_pass_and_oracle(engine, dev_switch, h, w, rng):
    n = 3
    img = synth.image_f32(n, h, w, seed=3)
    depth = synth.depth_batch("blobs", n, h, w, channels=3).astype(np.float32)
    if rng == "bytes":          # depth maps in 0..255: the x1 hypothesis for every frame
        depth = depth * 255.0
    elif rng == "mixed":        # one frame of each kind in the batch
        depth[1] = depth[1] * 255.0
    one = run(engine, img, depth, "polylines_soft")
    dev_switch("blur_no_pre_edges", 1)
    two = run(engine, img, depth, "polylines_soft")
    for k, (a, b) in enumerate(zip(one, two)):
        assert np.array_equal(bits(a), bits(b)), (h, w, rng, k)
    want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.1, 0.5, 2.0, UI["polylines_soft"], 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(one[k], want[k]), (h, w, rng, k)


@pytest.mark.parametrize("thr", [0.5, 5.0, 20.0, 37.3, 100.0])
def test_edge_thresholds(engine, dev_switch, thr):
    """The host-side threshold replaces the division for every edge_threshold: noisy depth puts many Sobel responses next to
    the decision value."""
    n, h, w = 2, 66, 772
    img = synth.image_f32(n, h, w, seed=4)
    rs = np.random.RandomState(17)
    depth = synth.depth_batch("blobs", n, h, w, channels=3).astype(np.float32)
    # the decision value is |Sobel| = 5 * edge_threshold in 0..255 units: noise of that order on top of the blobs
    amp = np.float32(5.0 * thr / 255.0 / 3.0)
    depth = np.clip(depth + rs.uniform(-1.0, 1.0, size=depth.shape).astype(np.float32) * amp, 0.0, 1.0).astype(np.float32)
    one = run(engine, img, depth, "none", thr=thr)
    dev_switch("blur_no_pre_edges", 1)
    two = run(engine, img, depth, "none", thr=thr)
    for k, (a, b) in enumerate(zip(one, two)):
        assert np.array_equal(bits(a), bits(b)), (thr, k)
    want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.1, 0.5, 2.0, UI["none"], thr, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(one[k], want[k]), (thr, k)


@pytest.mark.parametrize("fill", ["hybrid_edge", "gpu_warp", "none"])
def test_complete_map_consumers(engine, dev_switch, fill):
    """hybrid_edge and gpu_warp read complete blurred maps (k_blur_copy_tiles); gpu_warp takes the x255 decision over the
    sub-batch, so frames in 0..1 next to a frame in 0..255 all take the x1 plane."""
    n, h, w = 4, 72, 640
    img = synth.image_f32(n, h, w, seed=6)
    depth = synth.depth_batch("blobs", n, h, w, channels=3).astype(np.float32)
    depth[2] = depth[2] * 255.0
    one = run(engine, img, depth, fill)
    dev_switch("blur_no_pre_edges", 1)
    two = run(engine, img, depth, fill)
    for k, (a, b) in enumerate(zip(one, two)):
        assert np.array_equal(bits(a), bits(b)), (fill, k)
