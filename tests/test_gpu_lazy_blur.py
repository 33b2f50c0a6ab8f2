"""Lazy depth-blur tiles (cs_blur.hip k_blur_classify + the tile map read by the point-owner polylines kernel).

With the depth blur on and polylines_soft as the technique, tiles of the blurred depth maps without an edge in reach are
not written: the warp kernel reads gray * scale for them (reference stereoimage_generation.py:1171-1251 blends
w*blur + (1-w)*depth with w == 0 there).  cs_debug_set(CS_DEBUG_BLUR_FULL_COPY, 1) writes every tile instead: both forms
must give identical bits, and equal the oracle.
"""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(engine, img, depth, mode, div=6.0, sep=0.0, fill="polylines_soft", **kw):
    out = engine.generate(cuda(img), cuda(depth), div, sep, mode, 0.1, 0.5, 2.0, fill, 20.0, 20.0, True,
                          depth_blur_falloff=2.0, depth_blur_vert_smooth=6, **kw)
    return [o.cpu().numpy() for o in out]


UI = {"none": "No fill", "inverse": "No fill - Reverse projection", "naive": "Fill - Naive",
      "naive_interpolating": "Fill - Naive interpolating", "polylines_sharp": "Fill - Polylines Sharp",
      "polylines_soft": "Fill - Polylines Soft"}


@pytest.mark.parametrize("fill", ["none", "inverse", "naive", "naive_interpolating", "polylines_sharp"])
@pytest.mark.parametrize("mode", ["left-right", "red-cyan-anaglyph"])
def test_lazy_other_tile_kernels(engine, dev_switch, fill, mode):
    """cs_fwdtile.hip (none / inverse / naive / naive_interpolating, both eyes per workgroup) and cs_polytile.hip (sharp) read
    the tile map as well; the rows `naive` hands to the row kernel are completed first."""
    n, h, w = 2, 100, 1284
    img = synth.image_f32(n, h, w, seed=5)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    lazy = run(engine, img, depth, mode, fill=fill)
    dev_switch("blur_full_copy", 1)
    full = run(engine, img, depth, mode, fill=fill)
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 6.0, 0.0, mode, 0.1, 0.5, 2.0, UI[fill], 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), (fill, mode, k)


@pytest.mark.parametrize("fill,div", [("none", 90.0), ("polylines_soft", 130.0), ("polylines_soft", 180.0)])
def test_lazy_when_the_call_falls_back_to_wider_kernels(engine, dev_switch, fill, div):
    """Divergences whose halo exceeds the tile kernels' (none: row kernel for every row -> every row completed first;
    polylines_soft: the first-generation tile kernel, then the row kernel): same bits as the full copy and the oracle."""
    n, h, w = 1, 40, 1600
    img = synth.image_f32(n, h, w, seed=8)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    lazy = run(engine, img, depth, "left-right", div=div, fill=fill)
    dev_switch("blur_full_copy", 1)
    full = run(engine, img, depth, "left-right", div=div, fill=fill)
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, div, 0.0, "left-right", 0.1, 0.5, 2.0, UI[fill], 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), (fill, div, k)


@pytest.mark.parametrize("mode", ["left-right", "top-bottom", "red-cyan-anaglyph", "right-left"])
@pytest.mark.parametrize("kind", ["stepped", "blobs"])
def test_lazy_equals_full_copy_and_oracle(engine, dev_switch, mode, kind):
    n, h, w = 2, 200, 1284      # 21 tile columns (the last one partial), 7 tile rows (the last one partial)
    img = synth.image_f32(n, h, w, seed=3)
    depth = synth.depth_batch(kind, n, h, w, channels=3)
    lazy = run(engine, img, depth, mode)
    dev_switch("blur_full_copy", 1)
    full = run(engine, img, depth, mode)
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 6.0, 0.0, mode, 0.1, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), (mode, kind, k)


def test_lazy_with_unit_range_depth_scaled_by_255(engine, dev_switch):
    """depth in 0..1: the frame is multiplied by 255 (reference :1504); lazy tiles apply that factor in the warp kernel."""
    n, h, w = 2, 96, 2048 + 64
    img = synth.image_f32(n, h, w, seed=9)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    depth[1] *= 255.0   # one frame of each kind in the batch
    lazy = run(engine, img, depth, "left-right")
    dev_switch("blur_full_copy", 1)
    full = run(engine, img, depth, "left-right")
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.1, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), k


def test_lazy_rows_the_tile_kernel_flags_are_completed(engine, dev_switch):
    """Depth clipped to 0 / 1 around convergence 0.5 ties everywhere: every row goes to the general row kernel, which reads
    complete depth rows -- k_lazy_rows fills in the unwritten tiles first."""
    n, h, w = 1, 70, 1600
    img = synth.image_f32(n, h, w, seed=4)
    d = (synth.blobs(h, w, seed=2) > 0.5).astype(np.float32)
    depth = np.repeat(d[None, :, :, None], 3, axis=3)
    lazy = run(engine, img, depth, "left-right", div=3.0)
    dev_switch("blur_full_copy", 1)
    full = run(engine, img, depth, "left-right", div=3.0)
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 3.0, 0.0, "left-right", 0.1, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), k


def test_lazy_falloff_zero_marks_every_tile(engine, dev_switch):
    """falloff 0: every weight is 1, no tile is a copy -- the map is full and nothing is read from the gray depth."""
    n, h, w = 1, 64, 640
    img = synth.image_f32(n, h, w, seed=6)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    a = engine.generate(cuda(img), cuda(depth), 5.0, 0.0, "left-right", 0.0, 0.5, 2.0, "polylines_soft", 12.0, 12.0, True,
                        depth_blur_falloff=0.0, depth_blur_vert_smooth=3)
    want = node_oracle.generate(img, depth, 5.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 12.0, 12.0, True,
                                depth_blur_falloff=0.0, depth_blur_vert_smooth=3)
    for k in range(4):
        assert np.array_equal(a[k].cpu().numpy(), want[k]), k


@pytest.mark.parametrize("fill,mode", [("polylines_soft", "left-right"), ("none", "red-cyan-anaglyph"), ("polylines_sharp", "top-bottom")])
def test_lazy_with_the_compact_uint8_stereoscope(engine, fill, mode):
    """cs_params.flags bit 1 (the form multi-GPU shards are gathered in) with the blur on: the uint8 codes expand to the
    float32 output of the same call."""
    n, h, w = 2, 72, 1284
    img = cuda(synth.image_f32(n, h, w, seed=7))
    depth = cuda(synth.depth_batch("blobs", n, h, w, channels=3))
    ref = engine.generate(img, depth, 6.0, 0.0, mode, 0.0, 0.5, 2.0, fill, 20.0, 20.0, True, depth_blur_falloff=2.0,
                          depth_blur_vert_smooth=6)
    p = engine.make_params(n, h, w, h, w, 3, fill, mode, 6.0, 0.0, 0.0, 0.5, 2.0, True, 20.0, 20.0, 2.0, 6, 4)
    plan = engine.Plan(p, img.device, stereo_u8=True)
    out = plan.run(img, depth)
    assert out[0].dtype == torch.uint8
    assert torch.equal(engine.expand_u8(out[0]), ref[0])
    for k in (1, 2, 3):
        assert torch.equal(out[k], ref[k])


@pytest.mark.parametrize("shape", [(2, 100, 1282), (1, 61, 1284), (2, 20, 60)])
def test_lazy_odd_shapes_and_resized_depth(engine, dev_switch, shape):
    """Widths that are not multiples of 4 (the blur then writes complete maps), heights that end inside a tile row, frames
    smaller than one tile, and a depth map at another resolution (the tile map then refers to the RESIZED gray depth)."""
    n, h, w = shape
    img = synth.image_f32(n, h, w, seed=2)
    for dshape in ((h, w), (h // 2 + 3, w // 2 + 5)):
        depth = synth.depth_batch("blobs", n, dshape[0], dshape[1], channels=3)
        lazy = run(engine, img, depth, "left-right")
        dev_switch("blur_full_copy", 1)
        full = run(engine, img, depth, "left-right")
        dev_switch("blur_full_copy", 0)
        for a, b in zip(lazy, full):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (shape, dshape)
        if dshape == (h, w):
            want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.1, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                        depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
            for k in range(4):
                assert np.array_equal(lazy[k], want[k]), (shape, k)


@pytest.mark.parametrize("mode,balance", [("left-right", 0.0), ("top-bottom", 0.3), ("red-cyan-anaglyph", 0.0), ("left-right", 1.0)])
def test_gpu_warp_reads_the_tile_map(engine, dev_switch, mode, balance):
    """gpu_warp (rows of at most 2048 columns) reads the lazy tiles as well (cs_gpuwarp.hip pass 1; stereo_balance 1: one eye is
    the source image and its depth map is written by the tail loop): identical bits with complete maps
    (cs_debug_set(CS_DEBUG_GPUWARP_FULL_MAPS, 1)), mask equal to the oracle's, colours within the gpu_warp tolerance."""
    n, h, w = 3, 70, (1284 if balance == 0.0 else 3080)   # (3080 columns: the second word of tile bits, 1024-thread workgroups)
    img = synth.image_f32(n, h, w, seed=21)
    depth = synth.depth_batch("blobs", n, h, w, channels=3).astype(np.float32)
    depth[1] *= 255.0   # the x255 decision is taken over the sub-batch (reference :1045): these three frames stay unscaled
    args = (5.0, 0.5, mode, balance, 0.5, 2.0)
    lazy = [o.cpu().numpy() for o in engine.generate(cuda(img), cuda(depth), *args, "gpu_warp", 20.0, 20.0, True,
                                                     depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=3)]
    dev_switch("gpuwarp_full_maps", 1)
    full = [o.cpu().numpy() for o in engine.generate(cuda(img), cuda(depth), *args, "gpu_warp", 20.0, 20.0, True,
                                                     depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=3)]
    for k, (a, b) in enumerate(zip(lazy, full)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mode, balance, k)
    want = node_oracle.generate(img, depth, *args, "GPU Warp (Fast)", 20.0, 20.0, True, depth_blur_falloff=2.0,
                                depth_blur_vert_smooth=6, batch_size=3)
    assert np.array_equal(lazy[3], want[3]) and np.array_equal(lazy[1], want[1]) and np.array_equal(lazy[2], want[2])
    assert np.abs(lazy[0] - want[0]).max() <= 1e-4


@pytest.mark.parametrize("mode", ["left-right", "bottom-top"])
def test_hybrid_edge_reads_the_tile_map(engine, dev_switch, mode):
    """hybrid_edge through the fused splat tile kernel reads the lazy tiles as well: identical bits with complete maps
    (cs_debug_set(CS_DEBUG_HYBRID_FULL_MAPS, 1)) and the oracle's."""
    n, h, w = 2, 70, 1284
    img = synth.image_f32(n, h, w, seed=22)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    lazy = run(engine, img, depth, mode, fill="hybrid_edge")
    dev_switch("hybrid_full_maps", 1)
    full = run(engine, img, depth, mode, fill="hybrid_edge")
    for a, b in zip(lazy, full):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 6.0, 0.0, mode, 0.1, 0.5, 2.0, "Imperfect fill - Hybrid Edge", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(lazy[k], want[k]), (mode, k)
