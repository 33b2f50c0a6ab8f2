"""Parity of the HIP kernels (through the C ABI) against the CPU oracle and the golden fixtures.  -m gpu."""
import numpy as np
import pytest
import torch

import synth
from conftest import assert_warp_colours, node_case_expected, node_case_inputs
from oracle import node_oracle, oracle

pytestmark = pytest.mark.gpu

CPU_FILLS = ["none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp", "inverse", "hybrid_edge"]
HIDDEN_FILLS = ["none_post", "inverse_post", "hybrid_edge_plus"]  # dispatcher branches no UI string reaches (:1605-1610)
GPU_WARP_COLOUR_TOL = 1e-4


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_device_powf_bit_exact(engine):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(2_000_000, dtype=np.float32), np.float32([0, 1, 0.5, 1e-30, 1e-40, 1e-44, 2.5, 100.0])])
    L = oracle.lib()
    for e in [i / 10 for i in range(1, 21)] + [2.7, 0.05]:
        got = engine.test_powf(cuda(x), e).cpu().numpy()
        want = np.array([L.oracle_powf(float(v), float(np.float32(e))) for v in x[:: 97]], dtype=np.float32)
        assert np.array_equal(got[:: 97].view(np.uint32), want.view(np.uint32)), e


def test_device_powf_exponent_shortcuts(engine):
    """The tile kernel's shortcuts for exponents 2.0 (x * x unless flagged risky) and 1.0 (identity): the test kernel
    returns a NaN marker where a shortcut disagrees with the full routine -- a whole binade, a tiny one, the stored
    vectors where powf(x, 2) != x * x (tests/golden/powf_square.npz) and a strided sweep of all finite x >= 0."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "powf_square.npz"))
    L = oracle.lib()
    parts = [np.arange(0x3f000000, 0x3f800000, dtype=np.uint32), np.arange(0x35800000, 0x36000000, 3, dtype=np.uint32),
             np.arange(0, 0x7f800000, 509, dtype=np.uint32), g["x"].view(np.uint32)]
    x = np.concatenate(parts).view(np.float32)
    for e in (2.0, 1.0):
        got = engine.test_powf(cuda(x), e).cpu().numpy()
        assert not np.isnan(got[~np.isnan(x)]).any(), e
        idx = np.concatenate([np.arange(0, x.size, 9973), np.arange(x.size - g["x"].size, x.size)])
        want = np.array([L.oracle_powf(float(v), e) for v in x[idx]], dtype=np.float32)
        assert np.array_equal(got[idx].view(np.uint32), want.view(np.uint32)), e
    assert np.array_equal(engine.test_powf(cuda(g["x"]), 2.0).cpu().numpy().view(np.uint32), g["powf_x_2"].view(np.uint32))


def test_device_exp_bit_exact(engine):
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.random(200000) * 330.0, -(rng.random(200000, dtype=np.float32).astype(np.float64) ** 2) * 2,
                        [0.0, -0.0, -0.5, -1.0, -1e-300]])
    got = engine.test_exp(cuda(x)).cpu().numpy()
    L = oracle.lib()
    want = np.array([L.oracle_exp(float(v)) for v in x[:: 13]])
    assert np.array_equal(got[:: 13].view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("fill", CPU_FILLS)
def test_apply_stereo_divergence_goldens(engine, golden_asd, fill):
    """Bit-exact against the outputs captured from the imported reference."""
    g = golden_asd
    for case in g.meta["cases"]:
        cid = case["id"]
        got = engine.apply_stereo_divergence(cuda(g[f"{cid}/img"]), cuda(g[f"{cid}/depth"]), case["divergence"],
                                             case["separation"], case["exponent"], fill, case["convergence"])
        assert np.array_equal(got.cpu().numpy(), g[f"{cid}/out/{fill}"]), (cid, fill)


@pytest.mark.parametrize("fill", HIDDEN_FILLS)
def test_hidden_techniques_goldens(engine, golden_asd, golden_hidden, fill):
    """none_post / inverse_post / hybrid_edge_plus: bit-exact against the outputs captured from the reference."""
    for case in golden_hidden.meta["cases"]:
        cid = case["id"]
        got = engine.apply_stereo_divergence(cuda(golden_asd[f"{cid}/img"]), cuda(golden_asd[f"{cid}/depth"]), case["divergence"],
                                             case["separation"], case["exponent"], fill, case["convergence"])
        assert np.array_equal(got.cpu().numpy(), golden_hidden[f"{cid}/out/{fill}"]), (cid, fill)


@pytest.mark.parametrize("fill", CPU_FILLS + HIDDEN_FILLS)
@pytest.mark.parametrize("kind", ["blobs", "stepped", "noisy_ramp", "radial", "random8"])
def test_apply_stereo_divergence_vs_oracle(engine, fill, kind):
    """Seeded inputs at sizes the oracle finishes in seconds, batch of frames, both signs, exponents, separation."""
    h, w, n = 40, 640, 3
    params = [(6.0, 0.0, 2.0, 0.5), (-6.0, 0.4, 1.3, 0.4), (3.0, -0.8, 0.5, 0.0), (-9.0, 0.0, 1.0, 1.0)]
    for pi, (div, sep, e, conv) in enumerate(params):
        if kind == "random8" and (fill.startswith("poly") or fill == "hybrid_edge_plus"):
            div = 1.2 if div > 0 else -1.2
        img = np.stack([synth.image_u8(h, w, seed=10 * pi + j) for j in range(n)])
        depth = np.stack([synth.DEPTHS[kind](h, w, **({"cx": w / 2 + 31 * j} if kind in ("radial", "stepped") else {"seed": j}))
                          for j in range(n)]) * np.float32(255.0)
        got = engine.apply_stereo_divergence(cuda(img), cuda(depth), div, sep, e, fill, conv).cpu().numpy()
        for j in range(n):
            want = oracle.apply_stereo_divergence(img[j], depth[j], div, sep, e, fill, conv)
            bad = np.argwhere(got[j] != want)
            assert bad.size == 0, (fill, kind, pi, j, len(bad), bad[:4].tolist())


def test_node_goldens(engine, golden_node):
    """StereoImageNode.generate outputs captured from the reference (every UI technique x mode + variants)."""
    g = golden_node
    ran = 0
    for case in g.meta["cases"]:
        img, depth = node_case_inputs(g, case)
        kw = dict(case["kw"])
        got = engine.generate(cuda(img), cuda(depth), case["divergence"], case["separation"], case["mode"], case["balance"],
                              case["convergence"], case["exponent"], case["fill"], case["edge_threshold"], case["strength"],
                              case["blur"], **kw)
        got = [t.cpu().numpy() for t in got]
        want = node_case_expected(g, case)
        cid = case["id"]
        assert list(got[0].shape) == case["shapes"]["stereo"] and list(got[3].shape) == case["shapes"]["mask"], cid
        if case["fill"] == "gpu_warp":
            assert np.abs(got[0] - want[0]).max() <= GPU_WARP_COLOUR_TOL, (cid, np.abs(got[0] - want[0]).max())
        else:
            assert np.array_equal(got[0], want[0]), (cid, int((got[0] != want[0]).sum()))
        assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        assert np.array_equal(got[1][..., 0], got[1][..., 2]), cid
        assert np.array_equal(got[3], want[3]), cid
        ran += 1
    assert ran >= 46


def test_node_goldens_round2(engine, golden_node_extra):
    """The second node fixture: API-only modes (reference stereoimage_generation.py:1543-1562, :1094-1120), the fill strings
    outside the combo list (GenerateStereo.py:97-99) through the node class, and depth maps of another size -- the bilinear
    resize (GenerateStereo.py:141-148, 214-220) is bit-exact, so depth maps and mask are asserted as well."""
    from conftest import extra_case_expected, extra_case_inputs
    from comfystereo_amd.GenerateStereo import StereoImageNode
    g = golden_node_extra
    node = StereoImageNode()
    for case in g.meta["cases"]:
        img, depth = extra_case_inputs(g, case)
        out = node.generate(cuda(img), cuda(depth), case["divergence"], case["separation"], case["mode"], case["balance"],
                            case["convergence"], case["exponent"], case["fill_ui"], case["edge_threshold"], case["strength"],
                            case["blur"], **case["kw"])
        got = [t.cpu().numpy() for t in out]
        want = extra_case_expected(g, case)
        cid = case["id"]
        assert list(got[0].shape) == case["shapes"]["stereo"] and list(got[3].shape) == case["shapes"]["mask"], cid
        if case["gpu"]:
            assert np.abs(got[0] - want[0]).max() <= GPU_WARP_COLOUR_TOL, (cid, np.abs(got[0] - want[0]).max())
            assert np.abs(got[1][..., 0] - want[1]).max() <= 1e-6 and np.abs(got[2][..., 0] - want[2]).max() <= 1e-6, cid
        else:
            assert np.array_equal(got[0], want[0]), (cid, int((got[0] != want[0]).sum()))
            assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        assert np.array_equal(got[3], want[3]), cid


@pytest.mark.parametrize("fill,mode", [("polylines_soft", "left-right"), ("none", "red-cyan-anaglyph"),
                                       ("naive_interpolating", "top-bottom"), ("inverse", "right-left"),
                                       ("hybrid_edge", "bottom-top"), ("polylines_sharp", "left-right")])
def test_node_vs_oracle_blur_on(engine, fill, mode):
    n, h, w = 2, 256, 384
    img = synth.image_f32(n, h, w, seed=4)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    args = (4.5, 0.0, mode, 0.0, 0.5, 2.0)
    ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
    kw = dict(depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    got = engine.generate(cuda(img), cuda(depth), *args, fill, 20.0, 20.0, True, **kw)
    want = node_oracle.generate(img, depth, *args, ui, 20.0, 20.0, True, **kw)
    for gt, wv, name in zip(got, want, ("stereo", "dl", "dr", "mask")):
        assert np.array_equal(gt.cpu().numpy(), wv), (fill, mode, name)


def test_blur_goldens(engine, golden_blur):
    g = golden_blur
    for case in g.meta["cases"]:
        cid = case["id"]
        depth = g[f"{cid}/depth_u8"].astype(np.float32)
        L, R = engine.directional_blur(cuda(depth), case["strength"], case["edge_threshold"], case["falloff"], case["vert"])
        assert np.array_equal(L.cpu().numpy().view(np.uint32), g[f"{cid}/L"].view(np.uint32)), cid
        assert np.array_equal(R.cpu().numpy().view(np.uint32), g[f"{cid}/R"].view(np.uint32)), cid


@pytest.mark.parametrize("shape", [(1, 1080, 1920), (2, 300, 517)])
@pytest.mark.parametrize("prm", [(20, 20, 2.0, 6), (21, 6, 1.0, 0), (7.5, 2.0, 0.5, 15), (3, 40, 3.0, 1)])
def test_blur_vs_oracle(engine, shape, prm):
    n, h, w = shape
    depth = np.stack([np.round(synth.blobs(h, w, seed=j) * 255) for j in range(n)]).astype(np.float32)
    L, R = engine.directional_blur(cuda(depth), *prm)
    oL, oR = oracle.blur(depth, *prm)
    assert np.array_equal(L.cpu().numpy().view(np.uint32), oL.view(np.uint32))
    assert np.array_equal(R.cpu().numpy().view(np.uint32), oR.view(np.uint32))


def test_forward_warp_goldens(engine, golden_warp):
    g = golden_warp
    for case in g.meta["cases"]:
        cid = case["id"]
        img = g[f"{cid}/img_u8"].astype(np.float32) / np.float32(255.0)
        d8 = g[f"{cid}/depth_u8"].astype(np.float32)
        depth = d8 / np.float32(255.0) if case["depth_scale"] == 1.0 else d8
        warped, mask = engine.forward_warp(cuda(img), cuda(depth), case["divergence_px"], case["separation_px"],
                                           case["exponent"], case["convergence"])
        warped, mask = warped.cpu().numpy(), mask.cpu().numpy()
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        err = np.abs(warped - g[f"{cid}/warped"])
        if case["exponent"] in (2.0, 1.0, 0.5):
            assert np.array_equal(mask, want_mask), cid
            assert_warp_colours(warped, g[f"{cid}/warped"], want_mask, cid)
        else:
            assert (mask != want_mask).mean() <= 1e-3 and np.quantile(err, 0.999) <= 1e-3, cid


def test_forward_warp_keyword_parameters(engine, golden_warp_params):
    """forward_warp_gpu(gradient_threshold=, max_stretch=) (reference :277-279, :339-340, :365): the general instantiation of
    k_gpuwarp against fixtures of the reference and, on wider rows, against the oracle; the module-level shim passes them on."""
    g = golden_warp_params
    for case in g.meta["cases"]:
        cid = case["id"]
        img = g[f"{cid}/img_u8"].astype(np.float32) / np.float32(255.0)
        d8 = g[f"{cid}/depth_u8"].astype(np.float32)
        depth = d8 / np.float32(255.0) if case["depth_scale"] == 1.0 else d8
        warped, mask = engine.forward_warp(cuda(img), cuda(depth), case["divergence_px"], case["separation_px"], case["exponent"],
                                           case["convergence"], case["gradient_threshold"], case["max_stretch"])
        warped, mask = warped.cpu().numpy(), mask.cpu().numpy()
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        assert np.array_equal(mask, want_mask), cid
        assert_warp_colours(warped, g[f"{cid}/warped"], want_mask, cid)
    from comfystereo_amd import stereoimage_generation as sig
    b, h, w = 2, 40, 1300
    img = synth.image_f32(b, h, w, seed=16).transpose(0, 3, 1, 2).copy()
    for kind, dpx, thr, ms in [("blobs", 44.8, 0.7, 8), ("noisy_ramp", -30.0, 4.0, 5), ("stepped", 60.0, 13.0, 16), ("random8", 25.0, 2.5, 1)]:
        depth = np.stack([synth.DEPTHS[kind](h, w, **({"cx": w / 2 + 40 * j} if kind == "stepped" else {"seed": j})) for j in range(b)]) * np.float32(255.0)
        warped, mask = sig.forward_warp_gpu(torch.from_numpy(img), torch.from_numpy(depth), dpx, 0.5, 2.0, 0.5, gradient_threshold=thr, max_stretch=ms)
        ow, om = oracle.forward_warp_gpu(img, depth, dpx, 0.5, 2.0, 0.5, thr, ms)
        assert np.array_equal(mask.cpu().numpy(), om), (kind, thr, ms)
        assert np.abs(warped.cpu().numpy() - ow).max() <= 2e-6, (kind, thr, ms)
    with pytest.raises(RuntimeError):   # more than 16 effective rounds: refused, not truncated
        engine.forward_warp(cuda(img), cuda(depth), 10.0, 0.0, 2.0, 0.5, 20.0, 32)


@pytest.mark.parametrize("kind", ["blobs", "stepped", "noisy_ramp", "random8"])
def test_forward_warp_vs_oracle(engine, kind):
    b, h, w = 2, 96, 1280
    img = synth.image_f32(b, h, w, seed=6).transpose(0, 3, 1, 2).copy()
    for (dpx, spx, e, conv, scale) in [(44.8, 0.0, 2.0, 0.5, 255.0), (-44.8, 3.0, 1.0, 0.3, 1.0), (-20.0, 0.0, 0.5, 0.8, 255.0)]:
        depth = np.stack([synth.DEPTHS[kind](h, w, **({"cx": w / 2 + 40 * j} if kind == "stepped" else {"seed": j}))
                          for j in range(b)]) * np.float32(scale)
        warped, mask = engine.forward_warp(cuda(img), cuda(depth), dpx, spx, e, conv)
        ow, om = oracle.forward_warp_gpu(img, depth, dpx, spx, e, conv)
        assert np.array_equal(mask.cpu().numpy(), om), (kind, dpx)
        # same arithmetic as the oracle (which mirrors torch within GPU_WARP_COLOUR_TOL): tight tolerance
        assert np.abs(warped.cpu().numpy() - ow).max() <= 2e-6, (kind, dpx, np.abs(warped.cpu().numpy() - ow).max())


def test_node_gpu_warp_vs_oracle_blur_on(engine):
    n, h, w = 3, 256, 320
    img = synth.image_f32(n, h, w, seed=5)
    depth = synth.depth_batch("blobs", n, h, w, channels=3)
    for mode in ("left-right", "red-cyan-anaglyph", "top-bottom"):
        args = (4.5, 0.3, mode, 0.2, 0.5, 2.0)
        kw = dict(depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=2)
        got = engine.generate(cuda(img), cuda(depth), *args, "gpu_warp", 20.0, 20.0, True, **kw)
        want = node_oracle.generate(img, depth, *args, "GPU Warp (Fast)", 20.0, 20.0, True, **kw)
        assert np.abs(got[0].cpu().numpy() - want[0]).max() <= 2e-6, mode
        for k in (1, 2, 3):
            assert np.array_equal(got[k].cpu().numpy(), want[k]), (mode, k)


def test_blur_wide_kernel_takes_the_two_pass_path(engine):
    """strength 120.5: the fused tile (halo 2 x 120 columns) does not fit in LDS -> two-pass kernels; same bits."""
    n, h, w = 2, 96, 900
    depth = np.stack([np.round(synth.blobs(h, w, seed=j) * 255) for j in range(n)]).astype(np.float32)
    prm = (120.5, 6.0, 1.0, 2)
    L, R = engine.directional_blur(cuda(depth), *prm)
    oL, oR = oracle.blur(depth, *prm)
    assert np.array_equal(L.cpu().numpy().view(np.uint32), oL.view(np.uint32))
    assert np.array_equal(R.cpu().numpy().view(np.uint32), oR.view(np.uint32))


def test_blur_two_pass_switch(engine, dev_switch):
    dev_switch("blur_two_pass", 1)
    depth = np.round(synth.blobs(300, 517, seed=3) * 255).astype(np.float32)
    L, R = engine.directional_blur(cuda(depth), 20, 20, 2.0, 6)
    oL, oR = oracle.blur(depth, 20, 20, 2.0, 6)
    assert np.array_equal(L.cpu().numpy().view(np.uint32), oL.view(np.uint32))
    assert np.array_equal(R.cpu().numpy().view(np.uint32), oR.view(np.uint32))


def test_polylines_row_kernel_switch(engine, dev_switch):
    """cs_debug_set(CS_DEBUG_NO_TILE, 1) forces the general row kernel for every row: same result as the tiled fast path."""
    h, w = 24, 1500
    img = synth.image_u8(h, w, seed=11)
    depth = synth.blobs(h, w, seed=4) * np.float32(255)
    a = engine.apply_stereo_divergence(cuda(img), cuda(depth), 7.0, 0.3, 2.0, "polylines_soft", 0.5).cpu().numpy()
    dev_switch("no_tile", 1)
    b = engine.apply_stereo_divergence(cuda(img), cuda(depth), 7.0, 0.3, 2.0, "polylines_soft", 0.5).cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(a, oracle.apply_stereo_divergence(img, depth, 7.0, 0.3, 2.0, "polylines_soft", 0.5))


@pytest.mark.parametrize("exponent", [2.0, 1.0])
def test_polylines_exponent_shortcuts_switch(engine, dev_switch, exponent):
    """cs_debug_set(CS_DEBUG_DBG, 17) switches the tile kernel's exact shortcuts for exponents 2.0 / 1.0 off (full powf clone for every
    pixel): identical output, and both equal the oracle."""
    h, w = 16, 1800
    img = synth.image_u8(h, w, seed=3)
    depth = synth.blobs(h, w, seed=9) * np.float32(255)
    a = engine.apply_stereo_divergence(cuda(img), cuda(depth), 6.5, 0.2, exponent, "polylines_soft", 0.4).cpu().numpy()
    dev_switch("dbg", 17)
    b = engine.apply_stereo_divergence(cuda(img), cuda(depth), 6.5, 0.2, exponent, "polylines_soft", 0.4).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle.apply_stereo_divergence(img, depth, 6.5, 0.2, exponent, "polylines_soft", 0.4))


@pytest.mark.parametrize("fill,mode", [("polylines_soft", "left-right"), ("none", "red-cyan-anaglyph"), ("hybrid_edge", "top-bottom"),
                                       ("polylines_sharp", "right-left")])
def test_compact_u8_stereoscope_expands_to_the_float_output(engine, fill, mode):
    """cs_params.flags bit 1 (what multi-GPU shards are gathered in) + cs_expand_u8 == the float32 output."""
    n, h, w = 2, 40, 644
    img = cuda(synth.image_f32(n, h, w, seed=7))
    depth = cuda(synth.depth_batch("blobs", n, h, w, channels=3))
    ref = engine.generate(img, depth, 6.0, 0.0, mode, 0.0, 0.5, 2.0, fill, 20.0, 20.0, False)
    p = engine.make_params(n, h, w, h, w, 3, fill, mode, 6.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 1.0, 0, 4)
    plan = engine.Plan(p, img.device, stereo_u8=True)
    out = plan.run(img, depth)
    assert out[0].dtype == torch.uint8
    assert torch.equal(engine.expand_u8(out[0]), ref[0])
    for k in (1, 2, 3):
        assert torch.equal(out[k], ref[k])


@pytest.mark.parametrize("w,div_px", [(3000, 240.0), (1920, -200.0), (700, 150.0), (130, 90.0)])
def test_forward_warp_wide_gaps_vs_oracle(engine, w, div_px):
    """Disocclusion gaps wider than a wave's 64-column chunk, gaps that start at column 0 (no filled column to the left) and
    rows that are mostly gap: the left-nearest filled column comes from a DPP prefix maximum per chunk plus the chunk totals
    (cs_gpuwarp.hip wave_incl_max); 1024-, 512- and 256-thread workgroups (3, 4 and 3 passes over the row)."""
    rs = np.random.RandomState(31)
    b, h = 2, 5
    img = rs.rand(b, 3, h, w).astype(np.float32)
    depth = np.zeros((b, h, w), np.float32)
    depth[:, 0] = (np.arange(w) > w // 3).astype(np.float32)                 # one step: one gap of |div_px| / 4 .. / 2 columns
    depth[:, 1] = ((np.arange(w) // 97) % 2).astype(np.float32)              # alternating plateaus: a gap at every other edge
    depth[:, 2] = np.clip(np.arange(w) / (0.3 * w), 0, 1)                    # ramp then flat: stretching, no gap
    depth[:, 3] = (rs.rand(w) > 0.5).astype(np.float32)                      # noise: mostly gaps
    depth[:, 4] = 1.0 - (np.arange(w) > 40).astype(np.float32)               # a step next to the left border
    depth[1] = depth[1, :, ::-1]
    for e, conv in ((2.0, 0.5), (1.0, 0.0), (0.5, 1.0)):
        want, wmask = oracle.forward_warp_gpu(img, depth, div_px, 1.5, e, conv)
        got, gmask = engine.forward_warp(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), div_px, 1.5, e, conv)
        assert np.array_equal(gmask.cpu().numpy(), wmask), (w, div_px, e)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-6)
