"""Frame chunks of cs_generate (cs_abi.hip plan_chunks / generate_chunk; a development option -- measured zero-sum, the
default is one chunk): a batch is cut into chunks of frames, the pre-pass of chunk c + 1 (gray depth, min / max, depth blur)
runs on an auxiliary stream under the warp of chunk c.

Every quantity of the path is per frame (reference GenerateStereo.py:181-269 loops over frames; gpu_warp decides its two
0..255 scalings per sub-batch of `batch_size` frames, :119-128 and stereoimage_generation.py:1045, :313-316), so the chunked
schedule must give the bits of the one-chunk schedule -- and of the oracle.
"""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle

pytestmark = pytest.mark.gpu

UI = {"none": "No fill", "inverse": "No fill - Reverse projection", "naive": "Fill - Naive",
      "naive_interpolating": "Fill - Naive interpolating", "polylines_sharp": "Fill - Polylines Sharp",
      "polylines_soft": "Fill - Polylines Soft", "hybrid_edge": "Imperfect fill - Hybrid Edge", "gpu_warp": "GPU Warp (Fast)"}


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(engine, img, depth, fill, mode, blur, batch_size=12, **kw):
    out = engine.generate(cuda(img), cuda(depth), 6.0, 0.3, mode, 0.1, 0.5, 2.0, fill, 20.0, 20.0, blur,
                          depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=batch_size, **kw)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("fill,mode", [("polylines_soft", "left-right"), ("polylines_soft", "red-cyan-anaglyph"),
                                       ("polylines_sharp", "top-bottom"), ("none", "red-cyan-anaglyph"),
                                       ("naive_interpolating", "right-left"), ("inverse", "left-right"),
                                       ("hybrid_edge", "left-right")])
@pytest.mark.parametrize("blur", [False, True])
def test_chunked_equals_one_chunk_and_the_oracle(engine, dev_switch, fill, mode, blur):
    n, h, w = 7, 72, 700   # 3 chunks of 3 + 3 + 1 frames
    img = synth.image_f32(n, h, w, seed=11)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    dev_switch("chunks", 1)
    one = run(engine, img, depth, fill, mode, blur)
    dev_switch("chunks", 3)
    three = run(engine, img, depth, fill, mode, blur)
    for a, b in zip(one, three):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = node_oracle.generate(img, depth, 6.0, 0.3, mode, 0.1, 0.5, 2.0, UI[fill], 20.0, 20.0, blur,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6)
    for k in range(4):
        assert np.array_equal(three[k], want[k]), (fill, mode, blur, k)


def test_chunked_gpu_warp_keeps_the_sub_batch_decisions(engine, dev_switch):
    """gpu_warp: chunk boundaries fall on the reference's sub-batches, whose maximum decides the x255 scaling (depth maximum
    straddling 1.0 in one sub-batch only)."""
    n, h, w, bs = 10, 64, 512, 2
    img = synth.image_f32(n, h, w, seed=12)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    depth[3] *= 40.0   # this frame's sub-batch is "already 0..255"
    dev_switch("chunks", 1)
    one = run(engine, img, depth, "gpu_warp", "left-right", True, batch_size=bs)
    dev_switch("chunks", 3)   # 3 chunks of 4 + 4 + 2 frames (multiples of batch_size)
    three = run(engine, img, depth, "gpu_warp", "left-right", True, batch_size=bs)
    for a, b in zip(one, three):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_chunked_uint8_stereoscope_and_stats(engine, dev_switch):
    """cs_params.flags bit 1 (uint8 stereoscope codes): the chunk offsets are in bytes of that form; the statistics words of
    all frames stay one array at the start of the workspace."""
    n, h, w = 5, 64, 640
    img = synth.image_f32(n, h, w, seed=13)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    dev_switch("chunks", 2)
    def params():
        return engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 6.0, 0.0, 0.0, 0.5, 2.0, True, 20.0, 20.0, 2.0, 6, 12)

    pf = engine.Plan(params(), torch.device("cuda:0"))
    f32 = pf.run(cuda(img), cuda(depth))[0].cpu().numpy()
    pu = engine.Plan(params(), torch.device("cuda:0"), stereo_u8=True)
    u8 = pu.run(cuda(img), cuda(depth))[0].cpu().numpy()
    assert u8.dtype == np.uint8
    assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), f32)
    st = pf.stats()
    assert st.shape == (n, 16) and int(st[:, 9].sum()) == 0


def test_chunks_on_a_1080p_batch(engine, dev_switch):
    """Four chunks of a 1080p batch (auxiliary stream at the default priority) equal the one-chunk schedule."""
    n, h, w = 32, 1080, 1920
    img = synth.image_f32(1, h, w, seed=14).repeat(n, axis=0)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    dev_switch("chunks", 104)
    auto = run(engine, img, depth, "polylines_soft", "left-right", True)
    dev_switch("chunks", 1)
    one = run(engine, img, depth, "polylines_soft", "left-right", True)
    for a, b in zip(auto, one):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_profile_tiles_counts_on_the_device_over_all_chunks(engine, dev_switch):
    """cs_profile_tiles (the share of 64 x 32 blur tiles the profiled calls wrote, which bench.py turns into the kernel's own byte
    count): counted on the device into a buffer the LIBRARY owns, accumulated over the chunks of a call -- ADVICE r4: the first
    version read the last chunk's map through a pointer into the caller's workspace, which dangles once that is freed."""
    import ctypes
    from comfystereo_amd import _native
    L = _native.lib()
    n, h, w = 6, 160, 1100
    img = cuda(synth.image_f32(n, h, w, seed=5))
    depth = cuda(synth.depth_batch("stepped", n, h, w, channels=3))
    p = lambda: engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 6.0, 0.0, 0.0, 0.5, 2.0, True, 20.0, 20.0, 2.0, 6, 12)
    frac = ctypes.c_double(-2.0)

    def measure(chunks):
        dev_switch("chunks", chunks)
        plan = engine.Plan(p(), img.device)
        L.cs_profile(1)
        plan.run(img, depth)
        L.cs_profile(0)
        del plan   # the workspace goes back to the allocator before the count is read
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        _native.check(L.cs_profile_tiles(ctypes.byref(frac)))
        return frac.value

    one, three = measure(1), measure(3)
    assert 0.0 < one < 1.0 and one == three
    # nothing profiled since cs_profile(1): -1
    L.cs_profile(1); L.cs_profile(0)
    _native.check(L.cs_profile_tiles(ctypes.byref(frac)))
    assert frac.value == -1.0
    # a technique whose warp kernel does not read lazy tiles (complete maps): -1 as well
    L.cs_profile(1)
    q = engine.make_params(2, h, w, h, w, 3, "polylines_soft", "left-only", 6.0, 0.0, 0.0, 0.5, 2.0, True, 20.0, 20.0, 2.0, 6, 12)
    engine.Plan(q, img.device).run(img[:2], depth[:2])
    L.cs_profile(0)
    _native.check(L.cs_profile_tiles(ctypes.byref(frac)))
    assert frac.value == -1.0
