"""Full-size (4K) checks of the HIP path: one whole frame against the oracle, plus size-independent properties
(batch independence, run-to-run determinism, layout consistency between modes, Q10 pass-through) and ragged /
extreme shapes against the oracle.  -m gpu."""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle, oracle

pytestmark = pytest.mark.gpu
H4, W4 = 2160, 3840


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gen(engine, img, depth, fill, mode, blur=True, div=8.0, **kw):
    kw = dict(dict(depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12), **kw)
    return [t.cpu().numpy() for t in engine.generate(cuda(img), cuda(depth), div, 0.0, mode, 0.0, 0.5, 2.0, fill, 20.0, 20.0,
                                                     blur, **kw)]


def test_4k_frame_bit_exact_vs_oracle(engine):
    """The bench workload itself (one frame of it): 4K, stepped depth, divergence 8, polylines_soft, blur on."""
    img = synth.image_f32(1, H4, W4, seed=1)
    depth = synth.depth_batch("stepped", 1, H4, W4, channels=3)
    got = gen(engine, img, depth, "polylines_soft", "left-right")
    want = node_oracle.generate(img, depth, 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), name


@pytest.mark.parametrize("fill", ["polylines_soft", "none", "hybrid_edge", "gpu_warp"])
def test_4k_batch_independence_and_determinism(engine, fill):
    """Frame i of a batch == that frame processed alone; two runs are bit-identical (no atomic-order effects)."""
    n = 3
    img = synth.image_f32(n, H4, W4, seed=2)
    depth = synth.depth_batch("blobs", n, H4, W4, channels=1)
    a = gen(engine, img, depth, fill, "left-right", batch_size=1)
    b = gen(engine, img, depth, fill, "left-right", batch_size=1)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    solo = gen(engine, img[1:2], depth[1:2], fill, "left-right", batch_size=1)
    for x, y in zip(a, solo):
        assert np.array_equal(x[1:2], y)


def test_4k_layout_consistency(engine):
    """left-right / right-left / top-bottom / anaglyph / single-eye outputs are re-arrangements of the same two eyes."""
    img = synth.image_f32(1, H4, W4, seed=3)
    depth = synth.depth_batch("stepped", 1, H4, W4, channels=3)
    lr = gen(engine, img, depth, "polylines_soft", "left-right")
    L, R = lr[0][:, :, :W4], lr[0][:, :, W4:]
    rl = gen(engine, img, depth, "polylines_soft", "right-left")
    assert np.array_equal(rl[0][:, :, :W4], R) and np.array_equal(rl[0][:, :, W4:], L)
    tb = gen(engine, img, depth, "polylines_soft", "top-bottom")
    assert np.array_equal(tb[0][:, :H4], L) and np.array_equal(tb[0][:, H4:], R)
    assert np.array_equal(tb[3][:, :H4], lr[3][:, :, :W4]) and np.array_equal(tb[3][:, H4:], lr[3][:, :, W4:])
    an = gen(engine, img, depth, "polylines_soft", "red-cyan-anaglyph")
    assert np.array_equal(an[0][..., 0], L[..., 0]) and np.array_equal(an[0][..., 1:], R[..., 1:])
    assert np.array_equal(an[3], (an[0].sum(-1) == 0).astype(np.float32))
    for k in (1, 2):
        assert np.array_equal(lr[k], rl[k]) and np.array_equal(lr[k], tb[k])


def test_4k_tiny_divergence_passes_the_image_through(engine):
    """Q10: an eye whose divergence is < 0.001 is the source image (here: both eyes)."""
    img = synth.image_f32(1, H4, W4, seed=4)
    depth = synth.depth_batch("radial", 1, H4, W4, channels=3)
    out = gen(engine, img, depth, "polylines_soft", "left-right", blur=False, div=0.0009)
    k = np.clip(img * np.float32(255), 0, 255).astype(np.uint8).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(out[0][:, :, :W4], k) and np.array_equal(out[0][:, :, W4:], k)


@pytest.mark.parametrize("shape", [(1, 7), (3, 2), (5, 33), (17, 130), (9, 515), (4, 1030), (2, 3841)])
@pytest.mark.parametrize("fill", ["none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp", "inverse",
                                  "hybrid_edge", "none_post", "inverse_post", "hybrid_edge_plus"])
def test_ragged_shapes_vs_oracle(engine, shape, fill):
    """Widths that are not multiples of 4 / 64 / the tile, single rows, 4K+1: every technique, both signs."""
    h, w = shape
    if w > 3000 and fill in ("polylines_soft", "polylines_sharp", "hybrid_edge", "hybrid_edge_plus"):
        h = 1
    img = synth.image_u8(h, w, seed=h * 1000 + w, hazards=True)
    depth = synth.noisy_ramp(h, w, seed=w, amp=0.02) * np.float32(255)
    for div, sep in ((5.0, 0.0), (-5.0, 0.7)):
        got = engine.apply_stereo_divergence(cuda(img), cuda(depth), div, sep, 2.0, fill, 0.5).cpu().numpy()
        want = oracle.apply_stereo_divergence(img, depth, div, sep, 2.0, fill, 0.5)
        assert np.array_equal(got, want), (shape, fill, div, int((got != want).sum()))


@pytest.mark.parametrize("fill", ["polylines_soft", "polylines_sharp"])
@pytest.mark.parametrize("div", [15.0, 45.0])
def test_large_halo(engine, fill, div):
    """Convergence 0, balance 0.9 at 1080p.  Divergence 15 %: halo ~ 294 px, the tiled path stages 2.2x the tile;
    45 %: halo > 700 px, beyond the tiled path -> general row kernel.  Same answer as the oracle either way."""
    h, w = 6, 1920
    img = synth.image_f32(1, h, w, seed=9)
    depth = synth.depth_batch("blobs", 1, h, w, channels=3)
    ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
    args = (div, 1.0, "left-right", 0.9, 0.0, 1.0)
    got = engine.generate(cuda(img), cuda(depth), *args, fill, 20.0, 20.0, False)
    want = node_oracle.generate(img, depth, *args, ui, 20.0, 20.0, False)
    for g, w_ in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w_)


def test_flat_and_one_channel_depth(engine):
    h, w = 32, 200
    img = synth.image_f32(2, h, w, seed=5)
    flat = np.full((2, h, w, 1), 0.37, np.float32)
    for fill, ui in (("polylines_soft", "Fill - Polylines Soft"), ("gpu_warp", "GPU Warp (Fast)"), ("inverse", "No fill - Reverse projection")):
        args = (6.0, 0.0, "left-right", 0.0, 0.5, 2.0)
        got = engine.generate(cuda(img), cuda(flat), *args, fill, 20.0, 20.0, False)
        want = node_oracle.generate(img, flat, *args, ui, 20.0, 20.0, False)
        for k, (g, w_) in enumerate(zip(got, want)):
            if fill == "gpu_warp" and k == 0:
                assert np.abs(g.cpu().numpy() - w_).max() <= 2e-6
            else:
                assert np.array_equal(g.cpu().numpy(), w_), (fill, k)


@pytest.mark.parametrize("hole", [100, 159, 160, 161, 162, 163, 164, 200])
def test_naive_interpolating_hole_lengths_around_the_walk_limit(engine, hole):
    """naive_interpolating computes the hole ramps in parallel through walks bounded at 160 pixels and replays longer
    intervals sequentially: holes just below / at / above the bound, ending in filled pixels with channel sum 0 (which
    the ramp overwrites), both sweep directions."""
    h, w = 6, 2000
    rng = np.random.default_rng(hole)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[rng.random((h, w)) < 0.3] = [128, 128, 0]
    depth = np.zeros((h, w), np.float32)
    depth[:, w // 2:] = 255.0
    depth[1::2, w // 3: w // 2] = 255.0  # a second geometry on the odd rows
    for sign in (1.0, -1.0):
        div = sign * hole * 100.0 / w
        got = engine.apply_stereo_divergence(cuda(img), cuda(depth), div, 0.0, 1.0, "naive_interpolating", 0.0).cpu().numpy()
        want = oracle.apply_stereo_divergence(img, depth, div, 0.0, 1.0, "naive_interpolating", 0.0)
        assert np.array_equal(got, want), (hole, sign, int((got != want).sum()))


@pytest.mark.parametrize("hole", [12, 40, 63, 64, 65, 90, 128, 200])
def test_naive_interpolating_tile_kernel_retrigger_chains(engine, hole):
    """The NODE path of naive_interpolating (k_fwdtile, round 5: flags as bit rows, quirk intervals replayed by a whole wave): holes
    shorter than / equal to / longer than a wave and longer than the tile's window margin (those rows go back to the row kernel),
    starting at tile boundaries on some rows, an image in which 30 % of the pixels sum to 0 mod 256 -- so that most intervals
    re-trigger, several times in a row -- and genuinely black FILLED pixels inside the intervals (a ramp written over them becomes a
    right border of the next trigger).  Both eyes (opposite sweep directions), every output, against the oracle."""
    h, w = 10, 2100
    rng = np.random.default_rng(1000 + hole)
    img8 = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    img8[rng.random((1, h, w)) < 0.3] = [128, 128, 0]
    img8[rng.random((1, h, w)) < 0.1] = [255, 1, 0]
    img8[rng.random((1, h, w)) < 0.1] = 0
    img = img8.astype(np.float32) / np.float32(255.0)
    depth = np.zeros((1, h, w, 3), np.float32)
    for r in range(h):                      # a step per row, at a different column: some land on tile boundaries (w / 4, w / 3 ...)
        c = [w // 2, w // 4, w // 3, 640, 641, 1279, 1280, 700, 1400, 1050][r]
        depth[0, r, c:] = 1.0
        if r % 3 == 0:
            depth[0, r, c + 300: c + 500] = 0.0   # and back: holes of the other eye
    div = hole * 100.0 / w                  # exponent 1, convergence 0: the far side moves by `hole` columns
    got = engine.generate(cuda(img), cuda(depth), div, 0.0, "left-right", 0.0, 0.0, 1.0, "naive_interpolating", 20.0, 20.0, False)
    want = node_oracle.generate(img, depth, div, 0.0, "left-right", 0.0, 0.0, 1.0, "Fill - Naive interpolating", 20.0, 20.0, False)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        g = g.cpu().numpy()
        assert np.array_equal(g, w_), (hole, name, int((g != w_).sum()))


@pytest.mark.parametrize("div", [2.0, 9.0, 15.0])
def test_gpu_warp_wide_gaps_and_empty_rows(engine, div):
    """k_gpuwarp's gap fill on bit rows (round 5): gaps wider than one and than two 32-bit words, gaps that start at column 0 (no
    filled column to the left), rows whose every pair is disconnected (nothing filled at all), next to ordinary rows.  Mask exact,
    colours within the forward-warp tolerances, against the oracle."""
    h, w = 12, 1500
    img = synth.image_f32(1, h, w, seed=77)
    depth = np.zeros((1, h, w, 3), np.float32)
    depth[0, :, w // 2:] = 1.0
    depth[0, 1] = np.tile(np.array([0.0, 1.0], np.float32), w // 2)[:, None]   # every pair disconnected
    depth[0, 2, :40] = 1.0
    depth[0, 3, 100:700] = np.linspace(0.0, 1.0, 600, dtype=np.float32)[:, None]
    depth[0, 4:, :] = synth.depth_batch("stepped", 1, h - 4, w, channels=3)[0]
    got = engine.generate(cuda(img), cuda(depth), div, 0.0, "left-right", 0.0, 0.5, 1.0, "gpu_warp", 20.0, 20.0, False)
    want = node_oracle.generate(img, depth, div, 0.0, "left-right", 0.0, 0.5, 1.0, "GPU Warp (Fast)", 20.0, 20.0, False)
    got = [g.cpu().numpy() for g in got]
    assert np.array_equal(got[3], want[3])
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert np.abs(got[0] - want[0]).max() <= 1e-4


def test_digests_at_baseline_sizes_on_the_gpu():
    """cfg 1 (512 x 512 naive_interpolating), cfg 2 (1080p polylines_soft), cfg 3 at a quarter (1080p hybrid_edge), depth blur
    on: the HIP path against SHA-256 digests of the reference's own outputs (tests/golden/digests.json)."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd import engine
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    for cid, c in dig.items():
        img = synth.image_u8(c["h"], c["w"], seed=c["image_seed"], hazards=False)[None].astype(np.float32) / np.float32(255.0)
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), c["divergence"], 0.0, c["mode"], 0.0,
                              0.5, 2.0, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], 20.0, 20.0, c["blur"], depth_blur_falloff=2.0,
                              depth_blur_vert_smooth=6, batch_size=12)
        got = [t.cpu().numpy() for t in got]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"], cid
        assert sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid


def test_digest_of_the_metric_frame_at_4k_on_the_gpu(engine):
    """The metric's own frame through the HIP path against the REFERENCE node's outputs (tests/golden/digest_metric_4k.json): the uint8
    codes, the mask and the float32 arrays themselves, bit for bit -- `roofline` and `value` are quoted on a path whose result is the
    reference's."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    c = json.load(open(os.path.join(GOLDEN, "digest_metric_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
    depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
    got = gen(engine, img, depth, "polylines_soft", c["mode"], blur=c["blur"], div=c["divergence"])
    k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
    assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"]
    assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"]
    assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"]


def test_digests_of_the_other_bench_configurations_at_4k_on_the_gpu(engine):
    """One 4K frame of cfg 3 (hybrid_edge + blur), cfg 5 (`none`, red-cyan anaglyph + mask), naive_interpolating and polylines_sharp through
    the HIP path against the REFERENCE node's own outputs (tests/golden/digests_4k.json): uint8 codes, mask, float32 arrays -- every
    bench configuration but gpu_warp (tolerance-checked elsewhere) is quoted on a path whose full-size result is the reference's."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = gen(engine, img, depth, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], c["mode"], blur=c["blur"], div=c["divergence"])
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_order_dependent_depth_at_full_width_on_the_gpu(engine):
    """The tie path against the REFERENCE itself (tests/golden/digests_ties.json): a 4K frame of saturated depth and 48 rows of 8-bit noise
    through polylines_soft and polylines_sharp, blur off -- tile kernel, second tier, lean row kernel, lane and wave replay -- uint8 codes,
    mask and float32 arrays equal to what the reference's sequential sweep produced; and the rows did reach the row kernel."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests_ties.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        p = engine.make_params(1, c["h"], c["w"], c["h"], c["w"], 3, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], c["mode"], c["divergence"], 0.0, 0.0, 0.5, 2.0,
                               c["blur"], 20.0, 20.0, 2.0, 6, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
        st = plan.stats()
        assert int(st[:, 9].sum()) == 0 and int(st[:, 11].sum()) > 0, cid   # no kernel error flag; rows were handed to the row kernel
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_scene8_depth_at_4k_on_the_gpu(engine):
    """Estimator-like depth at full size against the REFERENCE itself (tests/golden/digests_scene8_4k.json): one 4K frame of scene8 depth at
    the metric's divergence through polylines_soft (blur on) and polylines_sharp (blur on / off) -- first tier, second tier, lean row pass in
    column ranges -- uint8 codes, mask and float32 arrays."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests_scene8_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    img = synth.image_f32(1, 2160, 3840, seed=1)
    depth = synth.depth_batch("scene8", 1, 2160, 3840, channels=3)
    for cid, c in dig.items():
        got = gen(engine, img, depth, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], c["mode"], blur=c["blur"], div=c["divergence"])
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_8k_frames_on_the_gpu(engine):
    """One 8K frame (7680 x 4320) against the REFERENCE itself (tests/golden/digests_8k.json): polylines_soft side by side (ten tiles per
    row) and naive_interpolating as a red-cyan anaglyph -- refused until round 6 -- divergence 8, blur on: uint8 codes, mask, float32 arrays."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests_8k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    img = synth.image_f32(1, 4320, 7680, seed=1)
    depth = synth.depth_batch("stepped", 1, 4320, 7680, channels=3)
    for cid, c in dig.items():
        got = gen(engine, img, depth, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], c["mode"], blur=c["blur"], div=c["divergence"])
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid
        del got, k


def test_digests_at_the_widths_round_6_opened_on_the_gpu():
    """The HIP path against the REFERENCE node's own outputs at the widths round 6 opened (tests/golden/digests_wide.json,
    tools/make_goldens.py --only-wide): wide anaglyphs of the forward and post fills, their new side-by-side limits, polylines_sharp
    at 8 192 columns -- by SHA-256 of the uint8 codes and the mask, like the BASELINE-size digests above."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from comfystereo_amd import engine
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
    dig = json.load(open(os.path.join(GOLDEN, "digests_wide.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        img[:, :, c["black"][0]:c["black"][1]] = 0.0
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = engine.generate(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), c["divergence"], 0.0, c["mode"], 0.0,
                              0.5, 2.0, FILL_TECHNIQUE_MAPPING[c["fill_ui"]], 20.0, 20.0, c["blur"], depth_blur_falloff=2.0,
                              depth_blur_vert_smooth=6, batch_size=12)
        got = [t.cpu().numpy() for t in got]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"], cid
        assert sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid


def test_forward_warp_1080p_rows_on_the_gpu():
    """forward_warp_gpu at 1080p against rows captured from the reference (gap mask exact; colours to the last ulps outside
    the gaps, <= 1e-4 for the few gap pixels whose torch.sqrt weight is not correctly rounded)."""
    from conftest import Golden, assert_warp_colours
    from comfystereo_amd import engine
    g = Golden("forward_warp_1080p.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        h, w = case["h"], case["w"]
        img = synth.image_f32(1, h, w, seed=case["image_seed"]).transpose(0, 3, 1, 2).copy()
        depth = synth.stepped(h, w)[None] * np.float32(255.0)
        warped, mask = engine.forward_warp(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), case["divergence_px"],
                                           case["separation_px"], case["exponent"], case["convergence"])
        warped, mask = warped.cpu().numpy(), mask.cpu().numpy()
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        assert np.array_equal(mask, want_mask), cid
        rows = case["rows"]
        assert_warp_colours(warped[:, :, rows, :], g[f"{cid}/rows"], want_mask[:, rows, :], cid)


UI = {"none": "No fill", "naive": "Fill - Naive", "naive_interpolating": "Fill - Naive interpolating",
      "polylines_soft": "Fill - Polylines Soft", "polylines_sharp": "Fill - Polylines Sharp",
      "inverse": "No fill - Reverse projection", "hybrid_edge": "Imperfect fill - Hybrid Edge", "gpu_warp": "GPU Warp (Fast)"}


@pytest.mark.parametrize("fill", sorted(UI))
def test_8k_wide_rows_side_by_side(engine, fill):
    """7680-pixel rows (an 8K frame is 7680 wide) through the node path, SBS: the techniques whose row state fits the LDS at
    that width (cs_max_width_mode; the anaglyph modes need 2 more bytes per pixel and stay below 8K for some of them)."""
    from comfystereo_amd import _native
    L = _native.lib()
    h, w = 6, 7680
    assert L.cs_max_width_mode(engine.FILL[fill], engine.MODE["left-right"]) >= w
    img = synth.image_f32(1, h, w, seed=8)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    got = gen(engine, img, depth, fill, "left-right", blur=False, div=3.0)
    want = node_oracle.generate(img, depth, 3.0, 0.0, "left-right", 0.0, 0.5, 2.0, UI[fill], 20.0, 20.0, False, batch_size=12)
    if fill == "gpu_warp":
        assert np.array_equal(got[3], want[3])
        assert np.abs(got[0] - want[0]).max() <= 1e-4
    else:
        for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
            assert np.array_equal(g, w_), name


@pytest.mark.parametrize("fill,w", [("none_post", 7680), ("inverse_post", 7680), ("none_post", 11578), ("inverse_post", 9004),
                                    ("naive", 11578), ("naive_interpolating", 9536)])
def test_wide_rows_at_the_new_limits(engine, fill, w):
    """Round 6: none_post / inverse_post (reference :1804-1833; no UI string reaches them) at 8K and at their new limits -- the nearest-valid-
    column arrays of the np.interp post-fill overlay the normalised depth and the forward map's winner / key words in LDS (7 368 / 6 234
    columns until then) -- and naive / naive_interpolating at theirs (nearest-filled columns / new colours over the dead normalised
    depth: 9 004 -> 11 578, 8 104 -> 9 536).  Depth with long holes (stepped, divergence 6) and a noise row, SBS, against the oracle."""
    from comfystereo_amd import _native
    L = _native.lib()
    assert L.cs_max_width_mode(engine.FILL[fill], engine.MODE["left-right"]) >= w
    ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
    h = 6
    img = synth.image_f32(1, h, w, seed=8)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    depth[0, 2] = synth.depth_batch("random8", 1, h, w, channels=3)[0, 2]
    got = gen(engine, img, depth, fill, "right-left", blur=False, div=6.0)
    want = node_oracle.generate(img, depth, 6.0, 0.0, "right-left", 0.0, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), (fill, w, name)


@pytest.mark.parametrize("fill,kind", [("polylines_sharp", "stepped"), ("polylines_sharp", "clipped"), ("polylines_soft", "clipped")])
def test_8k_wide_rows_anaglyph_polylines(engine, fill, kind):
    """7680-pixel rows as an ANAGLYPH through the polylines techniques (round 5): both kernels write the eyes side by side as uint8
    codes, the composition follows -- the row kernel behind the tile kernel no longer needs its anaglyph stash (polylines_sharp was
    limited to 6 395 columns in these modes).  `clipped` depth: rows with exact closeness ties reach the row kernel and the
    replay kernel, which the anaglyph modes did not use before."""
    from comfystereo_amd import _native
    L = _native.lib()
    h, w = 5, 7680
    for mode in ("red-cyan-anaglyph", "cyan-red-reverseanaglyph"):
        assert L.cs_max_width_mode(engine.FILL[fill], engine.MODE[mode]) >= w
    img = synth.image_f32(1, h, w, seed=8)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    mode = "red-cyan-anaglyph"
    p = engine.make_params(1, h, w, h, w, 3, fill, mode, 3.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
    st = plan.stats()
    if kind == "clipped":
        assert int(st[:, 11].sum()) > 0, "no row was flagged: the test does not reach the row kernel"
    assert int(st[:, 9].sum()) == 0   # kernel error flags
    want = node_oracle.generate(img, depth, 3.0, 0.0, mode, 0.0, 0.5, 2.0, UI[fill], 20.0, 20.0, False, batch_size=12)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), (fill, kind, name)


@pytest.mark.parametrize("fill,w,blur", [("naive_interpolating", 7680, False), ("naive_interpolating", 9536, True), ("naive", 10240, False),
                                         ("inverse", 9004, False), ("none", 10240, True), ("none_post", 10240, False), ("inverse_post", 8192, True)])
def test_wide_anaglyph_rows_through_the_side_by_side_form(engine, fill, w, blur):
    """Round 6: an anaglyph wider than the row kernel takes WITH its anaglyph stash (two more bytes of LDS per column) runs the row kernel
    in its side-by-side form into uint8 scratch and composes afterwards (cs_abi.hip run_rows, ana_wide), like the polylines techniques
    since round 5: the anaglyph modes' width limit is the side-by-side modes' limit.  (Every case but the first is beyond the stash form;
    the first is the 8K anaglyph of naive_interpolating, refused until round 6 -- 7 365 columns -- and inside the stash form since the
    technique's scratch arrays overlay the dead normalised depth.)  Both anaglyph modes, blur off and on (lazy blur tiles completed
    per row), against the oracle."""
    from comfystereo_amd import _native
    L = _native.lib()
    ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
    h = 8
    img = synth.image_f32(1, h, w, seed=8)
    img[:, :, 500:560] = 0.0   # genuinely black pixels (mask, quirk Q6)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    for mode in ("red-cyan-anaglyph", "cyan-red-reverseanaglyph"):
        assert L.cs_max_width_mode(engine.FILL[fill], engine.MODE[mode]) == L.cs_max_width_mode(engine.FILL[fill], engine.MODE["left-right"]) >= w
        got = gen(engine, img, depth, fill, mode, blur=blur, div=4.0)
        want = node_oracle.generate(img, depth, 4.0, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur, depth_blur_falloff=2.0,
                                    depth_blur_vert_smooth=6, batch_size=12)
        for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
            assert np.array_equal(g, w_), (fill, mode, name)


def test_too_wide_frames_are_refused_not_truncated(engine):
    """Beyond cs_max_width_mode the call fails (CS_ELIMIT), it does not truncate.  What is left with a limit below 8 192 after round 6:
    gpu_warp (7 763), an anaglyph of polylines_sharp whose halo the tile kernels do not take (the row kernel's own anaglyph form:
    6 395), and -- outside the UI's reach -- the three hidden techniques, the mesh warp and D64 polylines.  (polylines_sharp itself:
    7 990 -> 8 206 columns in round 6, test_8192_wide_sharp_rows.)"""
    from comfystereo_amd import _native
    L = _native.lib()
    for w, mode in ((8320, "left-right"), (8320, "red-cyan-anaglyph")):
        assert 8192 <= L.cs_max_width_mode(engine.FILL["polylines_sharp"], engine.MODE[mode]) < w
        img = synth.image_f32(1, 4, w, seed=8)
        depth = synth.depth_batch("stepped", 1, 4, w, channels=3)
        with pytest.raises(RuntimeError, match="too wide"):
            gen(engine, img, depth, "polylines_sharp", mode, blur=False, div=3.0)
    # an anaglyph at 7680 whose halo is beyond the tile kernels (convergence 1.0, divergence 15 %: 1 152 columns)
    img = synth.image_f32(1, 4, 7680, seed=8)
    depth = synth.depth_batch("stepped", 1, 4, 7680, channels=3)
    with pytest.raises(RuntimeError, match="too wide"):
        engine.generate(cuda(img), cuda(depth), 15.0, 0.0, "red-cyan-anaglyph", 0.0, 1.0, 2.0, "polylines_sharp", 20.0, 20.0, False)
    assert L.cs_max_width_mode(engine.FILL["hybrid_edge_plus"], engine.MODE["left-right"]) < 7680
    for fill in ("none_post", "inverse_post"):   # (round 6: their post-fill arrays overlay dead LDS -- the limits of none / inverse)
        assert L.cs_max_width_mode(engine.FILL[fill], engine.MODE["left-right"]) >= 8192
    # one column beyond the limit of a forward fill's anaglyph (the side-by-side limit since round 6)
    wmax = L.cs_max_width_mode(engine.FILL["naive_interpolating"], engine.MODE["red-cyan-anaglyph"])
    img = synth.image_f32(1, 4, wmax + 1, seed=8)
    depth = synth.depth_batch("stepped", 1, 4, wmax + 1, channels=3)
    with pytest.raises(RuntimeError, match="too wide"):
        gen(engine, img, depth, "naive_interpolating", "red-cyan-anaglyph", blur=False, div=3.0)


@pytest.mark.parametrize("kind,mode", [("stepped", "left-right"), ("clipped", "left-right"), ("clipped", "red-cyan-anaglyph"), ("scene8", "top-bottom")])
def test_8192_wide_sharp_rows(engine, kind, mode):
    """Round 6: polylines_sharp rows of 8 192 columns (limit 8 206; 7 990 until then).  The row kernel behind the tile kernel keeps a
    per-pixel list capacity of only 2 w + 2 + 2 048 entries at that width (cs_rowwarp.hip poly_cap) and evaluates the rows the tile
    kernel flags in column ranges; `clipped` depth: exact ties, rows that reach the row kernel and the replay."""
    from comfystereo_amd import _native
    L = _native.lib()
    h, w = 6, 8192
    assert L.cs_max_width_mode(engine.FILL["polylines_sharp"], engine.MODE[mode]) >= w
    img = synth.image_f32(1, h, w, seed=8)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    p = engine.make_params(1, h, w, h, w, 3, "polylines_sharp", mode, 5.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
    st = plan.stats()
    if kind == "clipped":
        assert int(st[:, 11].sum()) > 0, "no row was flagged: the test does not reach the row kernel"
    assert int(st[:, 9].sum()) == 0   # kernel error flags
    want = node_oracle.generate(img, depth, 5.0, 0.0, mode, 0.0, 0.5, 2.0, UI["polylines_sharp"], 20.0, 20.0, False, batch_size=12)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), (kind, mode, name)


@pytest.mark.parametrize("kind", ["clipped", "random8"])
def test_8k_sharp_rows_with_ties(engine, kind):
    """7680-wide polylines_sharp rows that the tile kernel FLAGS (exact closeness ties): the row kernel behind it runs with a
    reduced capacity for its per-pixel segment lists at this width (cs_rowwarp.hip poly_cap) -- rows whose lists overflow are
    exported as one whole-row stretch to the replay kernel (sliding windows over 15 362 sorted points) or swept sequentially.
    Same pixels as the oracle; the statistics show that rows did take those paths."""
    h, w = 5, 7680
    img = synth.image_f32(1, h, w, seed=21)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    want = node_oracle.generate(img, depth, 4.0, 0.0, "left-right", 0.0, 0.5, 2.0, UI["polylines_sharp"], 20.0, 20.0, False, batch_size=12)
    p = engine.make_params(1, h, w, h, w, 3, "polylines_sharp", "left-right", 4.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
    st = plan.stats()
    assert int(st[:, 11].sum()) > 0, "no row was flagged: the test does not reach the row kernel"
    assert int(st[:, 9].sum()) == 0   # kernel error flags
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), (kind, name)


@pytest.mark.parametrize("kind", ["clipped", "random8", "stepped"])
def test_row_kernel_column_ranges(engine, dev_switch, kind):
    """The polylines row kernel evaluates a row whose per-pixel segment lists do not fit its LDS in 2 .. 4 column ranges (wide
    polylines_sharp rows need that; cs_debug_set(dbg, 30) forces two ranges on any row): same pixels as one pass and as the oracle,
    with the tile kernels off so that every row takes the row kernel, both polylines variants, lean + replay path and full kernel."""
    h, w = 6, 1540
    img = synth.image_f32(1, h, w, seed=31)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    for fill in ("polylines_soft", "polylines_sharp"):
        want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.0, 0.5, 2.0, UI[fill], 20.0, 20.0, False, batch_size=12)
        for switches in ((("dbg", 30),), (("dbg", 30), ("no_tile", 1)), (("dbg", 30), ("no_tile", 1), ("no_replay_kernel", 1))):
            for k, v in switches:
                dev_switch(k, v)
            got = gen(engine, img, depth, fill, "left-right", blur=False, div=6.0)
            for k, v in switches:
                dev_switch(k, 0)
            for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
                assert np.array_equal(g, w_), (kind, fill, switches, name)
    # a general exponent: the second eye's disparities need the powf tables again, whose LDS block the later ranges borrow
    want = node_oracle.generate(img, depth, 6.0, 0.0, "left-right", 0.0, 0.5, 1.4, UI["polylines_sharp"], 20.0, 20.0, False, batch_size=12)
    dev_switch("dbg", 30); dev_switch("no_tile", 1)
    got = [t.cpu().numpy() for t in engine.generate(cuda(img), cuda(depth), 6.0, 0.0, "left-right", 0.0, 0.5, 1.4, "polylines_sharp",
                                                    20.0, 20.0, False, depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)]
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g, w_), (kind, "exponent 1.4", name)


def test_order_dependent_rows_take_the_wave_replay(engine, dev_switch):
    """Noise depth on 8-bit levels: dozens of overlapping layers per pixel -- the per-pixel segment lists of the row kernel
    overflow and (where two layers tie) the rows are order-dependent.  With the column ranges switched off (round 4's ranges
    evaluate such rows in parallel after all: cs_debug_set(dbg, 31)) every row goes through the sequential replay of the
    reference's active list as ONE whole-row stretch -- done by a whole wave since round 2, by the replay kernel since round 4.
    Bit-exact against the oracle in both forms, at a width that spans several tiles, both polylines variants, SBS and anaglyph."""
    h, w = 24, 1540
    img = synth.image_f32(1, h, w, seed=12)
    depth = synth.depth_batch("random8", 1, h, w, channels=3)
    for fill, ui in (("polylines_soft", "Fill - Polylines Soft"), ("polylines_sharp", "Fill - Polylines Sharp")):
        # (anaglyph: the tile kernels write the eyes into scratch, the flagged rows are written in final form by the row kernel
        # and skipped by the composition)
        for mode in ("left-right", "red-cyan-anaglyph"):
            want = node_oracle.generate(img, depth, 6.0, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
            p = engine.make_params(1, h, w, h, w, 3, fill, mode, 6.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
            for ranges_off in (1, 0):
                dev_switch("dbg", 31 if ranges_off else 0)
                plan = engine.Plan(p, torch.device("cuda"))
                got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
                assert int(plan.stats()[:, 11].sum()) > 0   # rows the tile kernel handed to the row kernel
                if ranges_off:
                    assert int(plan.stats()[:, 10].sum()) > 0   # rows replayed sequentially
                for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
                    assert np.array_equal(g, w_), (fill, mode, ranges_off, name)
            dev_switch("dbg", 0)


@pytest.mark.parametrize("fill,ui", [("polylines_soft", "Fill - Polylines Soft"), ("polylines_sharp", "Fill - Polylines Sharp")])
def test_saturated_depth_ties_are_replayed_in_stretches(engine, fill, ui):
    """Depth clipped to exactly 0 / 1 (what saturating depth estimators deliver) at convergence 0.5: every overlap of a near
    and a far layer is an exact closeness tie.  The rows are order-dependent only inside those overlaps; the row kernel replays
    the stretches between two pixels where the reference's active list holds a single segment (poly_replay_stretch), not the
    rows.  Bit-exact against the oracle, several frames so that rows with zero, one and many stretches occur."""
    n, h, w = 3, 96, 1540
    img = synth.image_f32(n, h, w, seed=21)
    depth = synth.depth_batch("clipped", n, h, w, channels=3)
    for mode, div in (("left-right", 6.0), ("red-cyan-anaglyph", 9.0), ("top-bottom", 2.5)):
        want = node_oracle.generate(img, depth, div, 0.0, mode, 0.2, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
        p = engine.make_params(n, h, w, h, w, 3, fill, mode, div, 0.0, 0.2, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
        st = plan.stats()
        assert int(st[:, 10].sum()) > 0 and int(st[:, 9].sum()) == 0   # rows with replayed pixels; no kernel error flags
        for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
            assert np.array_equal(g, w_), (fill, mode, name)


def test_anaglyph_through_the_tile_kernels_mixes_flagged_and_composed_rows(engine):
    """Anaglyph polylines: a frame whose upper half ties everywhere (rows redone by the row kernel, final form) and whose
    lower half is smooth (eyes composed from the tile kernels' scratch)."""
    h, w = 32, 1028
    img = synth.image_f32(1, h, w, seed=13)
    depth = synth.depth_batch("blobs", 1, h, w, channels=3)
    depth[:, : h // 2] = synth.depth_batch("random8", 1, h // 2, w, channels=3)
    for mode in ("red-cyan-anaglyph", "cyan-red-reverseanaglyph"):
        want = node_oracle.generate(img, depth, 5.0, 0.5, mode, 0.1, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, False, batch_size=12)
        p = engine.make_params(1, h, w, h, w, 3, "polylines_soft", mode, 5.0, 0.5, 0.1, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
        redone = int(plan.stats()[:, 11].sum())
        assert 0 < redone < 2 * h
        for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
            assert np.array_equal(g, w_), (mode, name)


def test_bench_batch_of_64_frames_through_the_lazy_blur_tiles(engine, dev_switch):
    """The bench call itself: 64 x 4K with the blur on.  It is the one call whose buffer distances (2.1 GB between the gray
    depth and a blurred map) come near the 32-bit offsets the tile kernels select with; every frame must equal the same
    frame processed in a batch of four with complete blurred maps."""
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * (1 << 30):
        pytest.skip("needs ~60 GB of device memory")
    n = 64
    dev = torch.device("cuda")
    img = torch.from_numpy(synth.image_f32(1, H4, W4, seed=1)).to(dev).expand(n, -1, -1, -1).contiguous()
    depth = torch.cat([torch.from_numpy(synth.depth_batch("stepped", 8, H4, W4, channels=3)).to(dev) for _ in range(n // 8)])
    depth[8:] += 0.001 * torch.arange(n - 8, device=dev).view(-1, 1, 1, 1)   # (frames differ: a mixed-up frame index would show)
    mk = lambda k: engine.make_params(k, H4, W4, H4, W4, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, True, 20.0,
                                      20.0, 2.0, 6, 12)
    out = engine.Plan(mk(n), dev).run(img, depth)
    dev_switch("blur_full_copy", 1)
    small = engine.Plan(mk(4), dev)
    for f0 in range(0, n, 4):
        ref = small.run(img[f0:f0 + 4], depth[f0:f0 + 4])
        for k in range(4):
            assert torch.equal(out[k][f0:f0 + 4], ref[k]), (f0, k)


# ---- every BASELINE.json configuration at its full frame size against the oracle (round 3) -------------------------------
NAMES = ("stereoscope", "depth_left", "depth_right", "mask")


def test_cfg3_4k_hybrid_edge_blur_on_vs_oracle(engine):
    """BASELINE cfg 3: one 4K frame, divergence 8, hybrid_edge + edge-aware depth blur (reference
    stereoimage_generation.py:1622-1661, :1745-1774): stereoscope, both depth maps and the imperfect-fill mask, bit for bit."""
    img = synth.image_f32(1, H4, W4, seed=21)
    depth = synth.depth_batch("stepped", 1, H4, W4, channels=3)
    got = gen(engine, img, depth, "hybrid_edge", "left-right")
    want = node_oracle.generate(img, depth, 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Imperfect fill - Hybrid Edge", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    for g, w_, name in zip(got, want, NAMES):
        assert g.shape == w_.shape and np.array_equal(g, w_), name
    assert want[3].sum() > 0   # the mask is not trivially empty: pixels with no touched neighbour stay black


def test_cfg5_4k_no_fill_anaglyph_mask_vs_oracle(engine):
    """BASELINE cfg 5: one 4K frame, `No fill`, red-cyan anaglyph (:1850-1868, :1996-2010) with the no-fill mask
    (GenerateStereo.py:355-361) compared with array_equal -- "bit-exact masks vs reference"."""
    img = synth.image_f32(1, H4, W4, seed=22)
    depth = synth.depth_batch("stepped", 1, H4, W4, channels=3)
    got = gen(engine, img, depth, "none", "red-cyan-anaglyph")
    want = node_oracle.generate(img, depth, 8.0, 0.0, "red-cyan-anaglyph", 0.0, 0.5, 2.0, "No fill", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    for g, w_, name in zip(got, want, NAMES):
        assert g.shape == w_.shape and np.array_equal(g, w_), name
    assert 0 < want[3].mean() < 0.5   # disocclusions of both eyes' holes that coincide in the composite + genuine black pixels


def test_cfg4_1080p_gpu_warp_sub_batch_vs_oracle(engine):
    """BASELINE cfg 4: one reference sub-batch (batch_size 12) of 1080p frames, gpu_warp, radial depth with a moving centre
    (stereoimage_generation.py:1005-1128, :277-450): masks and depth maps exact, colours within the bounds of
    conftest.assert_warp_colours (the oracle mirrors torch's arithmetic: 2e-6)."""
    n, h, w = 12, 1080, 1920
    img = synth.image_f32(1, h, w, seed=23).repeat(n, axis=0)
    img[1::2] = img[1::2, ::-1]   # (two different images, cheaply)
    depth = synth.depth_batch("radial", n, h, w, channels=3)
    got = gen(engine, img, depth, "gpu_warp", "left-right", div=4.5)
    want = node_oracle.generate(img, depth, 4.5, 0.0, "left-right", 0.0, 0.5, 2.0, "GPU Warp (Fast)", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    for k in (1, 2, 3):
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), NAMES[k]
    assert np.abs(got[0] - want[0]).max() <= 2e-6


def test_cfg2_1080p_polylines_soft_frame_vs_oracle(engine):
    """BASELINE cfg 2: 1080p, divergence 3.5, polylines_soft, left-right (:1912-1992)."""
    n, h, w = 2, 1080, 1920
    img = synth.image_f32(n, h, w, seed=24)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    got = gen(engine, img, depth, "polylines_soft", "left-right", div=3.5)
    want = node_oracle.generate(img, depth, 3.5, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                                depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    for g, w_, name in zip(got, want, NAMES):
        assert np.array_equal(g, w_), name


def test_4k_saturated_depth_stretch_replay_kernel_vs_oracle(engine, dev_switch):
    """Round 3: the stretches of order-dependent rows are replayed by a kernel of their own (k_poly_replay: one wave per stretch
    over the row's dumped sorted points) -- same bits as the replay inside the row kernel (cs_debug_set no_replay_kernel) and as
    the oracle (reference :1961-1980).  At 4K and divergence 8 an eye shifts up to 77 points out of the frame: a stretch that
    starts at column 0 begins with the reference's bulk add + remove of all of them (more than the 64-entry list), the closed
    form in poly_replay_stretch."""
    h, w = 240, 3840   # (a band of a 4K frame: the oracle replays every row)
    img = synth.image_f32(2, h, w, seed=31)
    depth = np.stack([synth.clipped(2160, w, seed=s)[900:900 + h] for s in (0, 3)])[..., None].repeat(3, -1)
    args = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0)
    want = node_oracle.generate(img, depth, *args, "Fill - Polylines Soft", 20.0, 20.0, False, batch_size=12)
    p = engine.make_params(2, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
    assert int(plan.stats()[:, 10].sum()) > 0 and int(plan.stats()[:, 9].sum()) == 0
    dev_switch("no_replay_kernel", 1)
    inrow = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(cuda(img), cuda(depth))]
    for g, r, w_, name in zip(got, inrow, want, NAMES):
        assert np.array_equal(g, r), ("replay kernel vs row kernel", name)
        assert np.array_equal(g, w_), ("vs oracle", name)


@pytest.mark.parametrize("fill,ui", [("polylines_soft", "Fill - Polylines Soft"), ("polylines_sharp", "Fill - Polylines Sharp")])
def test_replay_pool_nearly_full_never_hands_out_overlapping_windows(engine, dev_switch, fill, ui):
    """a row that finds the dump pool full gives its units back.  With a plain subtraction two failed rows
    refunding around a successful one could hand a later row a window inside the successful row's live region (its stretches
    were then replayed from another row's sorted points: silently wrong, nondeterministic pixels).  The refund is now one
    compare-and-swap that only succeeds while the failed reservation is the topmost one.  A deliberately tiny pool
    (cs_debug_set pt_variant 48: ~100 bytes per row) on saturated depth -- hundreds of flagged rows reserving at once, most of
    them failing -- must give the oracle's bits, several runs in a row (the interleaving differs from run to run)."""
    n, h, w = 4, 160, 2048
    img = synth.image_f32(n, h, w, seed=77)
    depth = np.stack([synth.clipped(1200, w, seed=s)[500:500 + h] for s in range(n)])[..., None].repeat(3, -1)
    want = node_oracle.generate(img, depth, 7.0, 0.0, "left-right", 0.0, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
    p = engine.make_params(n, h, w, h, w, 3, fill, "left-right", 7.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
    dev_switch("pt_variant", 48)
    plan = engine.Plan(p, torch.device("cuda"))
    dimg, ddepth = cuda(img), cuda(depth)
    for rep in range(6):
        got = [t.cpu().numpy() for t in plan.run(dimg, ddepth)]
        assert int(plan.stats()[:, 10].sum()) > 0 and int(plan.stats()[:, 9].sum()) == 0
        for g, w_, name in zip(got, want, NAMES):
            assert np.array_equal(g, w_), (rep, name)


@pytest.mark.parametrize("fill,ui", [("polylines_soft", "Fill - Polylines Soft"), ("polylines_sharp", "Fill - Polylines Sharp")])
def test_lean_pass_on_the_flagged_tiles_column_ranges(engine, dev_switch, fill, ui):
    """Round 5: k_polypoint records WHICH tiles of a row-eye raised the hazard (tile hints) and the lean first pass of the row kernel
    stages, sorts, lists and evaluates only those tiles' columns plus a margin (technique_polylines, PolyRange) -- the rest of the row
    keeps the tile kernel's pixels.  4K-wide rows, smooth depth with saturated patches (exact ties) at the left end, in the middle and
    at the right end of the row, so that ranges touch either sentinel and rows have one, two or three flagged tiles; against the
    oracle, and the same bits as the whole-row form of round 4 (cs_debug_set pt_variant 44)."""
    n, h, w = 2, 40, 3840
    img = synth.image_f32(n, h, w, seed=41)
    ramp = np.linspace(0.2, 0.8, w, dtype=np.float32)[None, None, :] + 0.05 * np.sin(np.arange(h, dtype=np.float32))[None, :, None]
    depth = np.repeat(ramp, n, axis=0)
    clip = np.stack([synth.clipped(2160, w, seed=s)[700:700 + h] for s in (1, 2)])
    depth[0, :, :420] = clip[0, :, :420]                 # frame 0: left end (every row) ...
    depth[0, 10:, 1700:2300] = clip[0, 10:, 1700:2300]   # ... and the middle (rows 10 on: two or three flagged tiles)
    depth[1, :, 3400:] = clip[1, :, 3400:]               # frame 1: right end
    depth[1, ::3, 900:1100] = clip[1, ::3, 900:1100]
    depth = np.ascontiguousarray(depth[..., None].repeat(3, -1))
    for mode, div in (("left-right", 8.0), ("top-bottom", 5.0)):
        want = node_oracle.generate(img, depth, div, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, False, batch_size=12)
        p = engine.make_params(n, h, w, h, w, 3, fill, mode, div, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
        plan = engine.Plan(p, torch.device("cuda"))
        got = [t.cpu().numpy() for t in plan.run(cuda(img), cuda(depth))]
        st = plan.stats()
        assert int(st[:, 11].sum()) > 0 and int(st[:, 9].sum()) == 0   # rows went back to the row kernel; no kernel error flags
        dev_switch("pt_variant", 44)
        whole = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(cuda(img), cuda(depth))]
        # (... and the stretches replayed by a lane each, k_poly_replay_lanes, against a wave each alone: pt_variant 45)
        dev_switch("pt_variant", 45)
        waves = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(cuda(img), cuda(depth))]
        dev_switch("pt_variant", 0)
        for g, r, v, w_, name in zip(got, whole, waves, want, NAMES):
            assert np.array_equal(g, r), (mode, "column ranges vs whole rows", name)
            assert np.array_equal(g, v), (mode, "lane replay vs wave replay", name)
            assert np.array_equal(g, w_), (mode, "vs oracle", name)


def test_naive_interpolating_second_tier_on_saturated_depth(engine, dev_switch):
    """Round 5: rows whose holes outgrow k_fwdtile's window (halo + 8 on either side: the holes between a near and a far plateau of depth
    saturated to 0 / 1 are up to twice the halo wide) no longer go to the whole-row kernel but through the tile kernel once more with a
    window of 2 halo + 16 (persistent workgroups over the flagged-row list); what that flags -- intervals lengthened by black pixels --
    still does.  4K-wide rows of saturated depth with black patches, SBS and anaglyph, blur off and on: the oracle's bits, and the same
    bits without the second tier (cs_debug_set pt_variant 47)."""
    n, h, w = 2, 48, 3840
    img = synth.image_f32(n, h, w, seed=51)
    img[:, 5:9, 1000:1400] = 0.0                      # black pixels lengthen intervals: beyond any window
    depth = np.stack([synth.clipped(2160, w, seed=s)[600:600 + h] for s in (4, 5)])[..., None].repeat(3, -1)
    depth = np.ascontiguousarray(depth)
    for mode, blur in (("left-right", False), ("red-cyan-anaglyph", False), ("left-right", True)):
        want = node_oracle.generate(img, depth, 8.0, 0.0, mode, 0.0, 0.5, 2.0, "Fill - Naive interpolating", 20.0, 20.0, blur,
                                    depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        p = engine.make_params(n, h, w, h, w, 3, "naive_interpolating", mode, 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0, 20.0, 2.0, 6, 12)
        got = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(cuda(img), cuda(depth))]
        dev_switch("pt_variant", 47)
        plan1 = engine.Plan(p, torch.device("cuda"))
        one = [t.cpu().numpy() for t in plan1.run(cuda(img), cuda(depth))]
        rows_one_tier = int(plan1.stats()[:, 11].sum())
        dev_switch("pt_variant", 0)
        plan2 = engine.Plan(p, torch.device("cuda"))
        plan2.run(cuda(img), cuda(depth))
        rows_two_tiers = int(plan2.stats()[:, 11].sum())
        if not blur:
            assert rows_two_tiers < rows_one_tier, (mode, rows_two_tiers, rows_one_tier)   # the second tier takes rows off the row kernel
        for g, r, w_, name in zip(got, one, want, NAMES):
            assert np.array_equal(g, r), (mode, blur, "two tiers vs one", name)
            assert np.array_equal(g, w_), (mode, blur, "vs oracle", name)
