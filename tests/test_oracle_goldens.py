"""The CPU oracle against every golden vector captured from the imported reference (not gpu)."""
import hashlib

import numpy as np
import pytest

import synth
from conftest import assert_warp_colours, node_case_expected, node_case_inputs
from oracle import node_oracle, oracle

GPU_WARP_COLOUR_TOL = 1e-4  # torch's vectorised bilinear grid_sample is not bit-reproducible (SURVEY B-16)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_apply_stereo_divergence_bit_exact(golden_asd):
    g = golden_asd
    n = 0
    for case in g.meta["cases"]:
        cid = case["id"]
        img, depth = g[f"{cid}/img"], g[f"{cid}/depth"]
        for fill in g.meta["fills"]:
            got = oracle.apply_stereo_divergence(img, depth, case["divergence"], case["separation"], case["exponent"],
                                                 fill, case["convergence"])
            assert np.array_equal(got, g[f"{cid}/out/{fill}"]), (cid, fill)
            n += 1
    assert n == len(g.meta["cases"]) * 7


def test_apply_stereo_divergence_digests(golden_asd):
    for d in golden_asd.meta["digests"]:
        img = synth.image_u8(d["h"], d["w"], seed=d["img_seed"])
        depth = synth.DEPTHS[d["kind"]](d["h"], d["w"]) * np.float32(255.0)
        assert sha(img) == d["img_sha"] and sha(depth) == d["depth_sha"], "synthetic generator drifted"
        for fill, want in d["out"].items():
            got = oracle.apply_stereo_divergence(img, depth, d["divergence"], d["separation"], d["exponent"], fill,
                                                 d["convergence"])
            assert sha(got) == want, (d["kind"], fill)


def test_hidden_techniques_bit_exact(golden_asd, golden_hidden):
    """none_post / inverse_post / hybrid_edge_plus (reference dispatcher :1605-1610) against the captured outputs."""
    g, n = golden_hidden, 0
    for case in g.meta["cases"]:
        cid = case["id"]
        for fill in g.meta["fills"]:
            got = oracle.apply_stereo_divergence(golden_asd[f"{cid}/img"], golden_asd[f"{cid}/depth"], case["divergence"],
                                                 case["separation"], case["exponent"], fill, case["convergence"])
            assert np.array_equal(got, g[f"{cid}/out/{fill}"]), (cid, fill)
            n += 1
    assert n == len(g.meta["cases"]) * 3
    for d in g.meta["digests"]:
        img = synth.image_u8(d["h"], d["w"], seed=d["img_seed"])
        depth = synth.DEPTHS[d["kind"]](d["h"], d["w"]) * np.float32(255.0)
        for fill, want in d["out"].items():
            assert sha(oracle.apply_stereo_divergence(img, depth, d["divergence"], d["separation"], d["exponent"], fill,
                                                      d["convergence"])) == want, (d["kind"], fill)


def test_blur_bit_exact(golden_blur):
    g = golden_blur
    for case in g.meta["cases"]:
        cid = case["id"]
        depth = g[f"{cid}/depth_u8"].astype(np.float32)
        L, R = oracle.blur(depth, case["strength"], case["edge_threshold"], case["falloff"], case["vert"])
        assert np.array_equal(L.view(np.uint32), g[f"{cid}/L"].view(np.uint32)), cid
        assert np.array_equal(R.view(np.uint32), g[f"{cid}/R"].view(np.uint32)), cid


def test_blur_digest_540p(golden_blur):
    d = golden_blur.meta["digests"][0]
    depth = np.round(synth.DEPTHS[d["kind"]](d["h"], d["w"], seed=d["seed"]) * 255).astype(np.float32)
    assert sha(depth) == d["depth_sha"]
    L, R = oracle.blur(depth, d["strength"], d["edge_threshold"], d["falloff"], d["vert"])
    assert sha(L) == d["L"] and sha(R) == d["R"]


def test_forward_warp_gpu(golden_warp):
    g = golden_warp
    for case in g.meta["cases"]:
        cid = case["id"]
        img = g[f"{cid}/img_u8"].astype(np.float32) / np.float32(255.0)
        d8 = g[f"{cid}/depth_u8"].astype(np.float32)
        depth = d8 / np.float32(255.0) if case["depth_scale"] == 1.0 else d8
        warped, mask = oracle.forward_warp_gpu(img, depth, case["divergence_px"], case["separation_px"],
                                               case["exponent"], case["convergence"])
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        if case["exponent"] in (2.0, 1.0, 0.5):  # torch.pow is exact only there
            assert np.array_equal(mask, want_mask), cid
        else:
            assert (mask != want_mask).mean() <= 1e-3, cid
        err = np.abs(warped - g[f"{cid}/warped"])
        if case["exponent"] in (2.0, 1.0, 0.5):
            assert_warp_colours(warped, g[f"{cid}/warped"], want_mask, cid)
        else:
            assert np.quantile(err, 0.999) <= 1e-3, cid


def test_forward_warp_gpu_keyword_parameters(golden_warp_params):
    """gradient_threshold (connectivity, reference :339-340) and max_stretch (scatter rounds, :365) away from the defaults the
    reference's own call sites use: 0.6 .. 12.5 and 0 .. 20, incl. a threshold of 0 (nothing connected) and no rounds at all."""
    g = golden_warp_params
    for case in g.meta["cases"]:
        cid = case["id"]
        img = g[f"{cid}/img_u8"].astype(np.float32) / np.float32(255.0)
        d8 = g[f"{cid}/depth_u8"].astype(np.float32)
        depth = d8 / np.float32(255.0) if case["depth_scale"] == 1.0 else d8
        warped, mask = oracle.forward_warp_gpu(img, depth, case["divergence_px"], case["separation_px"], case["exponent"],
                                               case["convergence"], case["gradient_threshold"], case["max_stretch"])
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        assert np.array_equal(mask, want_mask), cid
        assert_warp_colours(warped, g[f"{cid}/warped"], want_mask, cid)


def test_scipy_depth_blur_of_the_numpy_input_path():
    """`directional_motion_blur` (reference :1346-1419; scipy.ndimage sobel / convolve1d restated in oracle/scipy_blur_oracle.py)
    against outputs of the reference: bit-exact where NumPy's float32 array power is exact (falloff 2 / 1 / 0.5), one ulp of the
    weight otherwise (NumPy's SIMD pow is not glibc's)."""
    from conftest import Golden
    from oracle import scipy_blur_oracle as sb
    g = Golden("numpy_blur.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        depth = g[f"{cid}/depth"]
        left, right = sb.directional_motion_blur(depth, case["strength"], case["edge_threshold"], case["strength"],
                                                 falloff_exponent=case["falloff"], vert_smooth_px=case["vert"])
        for got, want in ((left, g[f"{cid}/left"]), (right, g[f"{cid}/right"])):
            assert got.dtype == np.float32
            if case["exact_power"]:
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), cid
            else:
                assert np.abs(got - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), cid


def numpy_path_with_blur(apply_div, blur, img, depth, kw):
    """create_stereoimages for numpy inputs with the blur on (reference :1486-1562), from the two building blocks."""
    left_d, right_d = blur(depth, kw["depth_blur_strength"], kw["depth_blur_edge_threshold"], kw["depth_blur_strength"],
                           falloff_exponent=kw["depth_blur_falloff"], vert_smooth_px=kw["depth_blur_vert_smooth"])
    ld, rd = kw["divergence"] * (1 + kw["stereo_balance"]), kw["divergence"] * (1 - kw["stereo_balance"])
    e, f, c = kw["stereo_offset_exponent"], kw["fill_technique"], kw["convergence_point"]
    left = img if ld < 0.001 else apply_div(img, left_d, +1 * ld, -1 * kw["separation"], e, f, c)
    right = img if rd < 0.001 else apply_div(img, right_d, -1 * rd, kw["separation"], e, f, c)
    out = []
    for m in kw["modes"]:
        out.append({"left-right": lambda: np.hstack([left, right]), "right-left": lambda: np.hstack([right, left]),
                    "top-bottom": lambda: np.vstack([left, right]), "bottom-top": lambda: np.vstack([right, left]),
                    "red-cyan-anaglyph": lambda: np.dstack([left[..., :1], right[..., 1:]]), "left-only": lambda: left,
                    "only-right": lambda: right}[m]())
    return out, np.clip(left_d, 0, 255).astype(np.uint8), np.clip(right_d, 0, 255).astype(np.uint8)


def test_create_stereoimages_numpy_inputs_with_the_blur_on():
    from conftest import Golden
    from oracle import scipy_blur_oracle as sb
    g = Golden("numpy_blur.npz")
    for case in g.meta["create_stereoimages"]:
        cid = case["id"]
        outs, ml, mr = numpy_path_with_blur(oracle.apply_stereo_divergence, sb.directional_motion_blur, g[f"{cid}/img"], g[f"{cid}/depth"], case)
        for k, o in enumerate(outs):
            assert np.array_equal(o, g[f"{cid}/out{k}"]), (cid, k)
        assert np.array_equal(ml, g[f"{cid}/mod_left"]) and np.array_equal(mr, g[f"{cid}/mod_right"]), cid


def test_node_generate(golden_node):
    """StereoImageNode.generate: every UI technique x mode + variants (reference GenerateStereo.py:79-353)."""
    g = golden_node
    for case in g.meta["cases"]:
        img, depth = node_case_inputs(g, case)
        ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[case["fill"]]
        got = node_oracle.generate(img, depth, case["divergence"], case["separation"], case["mode"], case["balance"],
                                   case["convergence"], case["exponent"], ui, case["edge_threshold"], case["strength"],
                                   case["blur"], **case["kw"])
        want = node_case_expected(g, case)
        cid = case["id"]
        assert list(got[0].shape) == case["shapes"]["stereo"] and list(got[3].shape) == case["shapes"]["mask"], cid
        assert list(got[1].shape) == case["shapes"]["depth"], cid
        if cid.startswith("resize/"):
            # bilinear depth resize goes through torch CPU in both; the rest is exact
            assert np.array_equal(got[0], want[0]), cid
        elif case["fill"] == "gpu_warp":
            assert np.abs(got[0] - want[0]).max() <= GPU_WARP_COLOUR_TOL, (cid, np.abs(got[0] - want[0]).max())
            assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        else:
            assert np.array_equal(got[0], want[0]), cid
            assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        assert np.array_equal(got[3], want[3]), cid


def test_stereo_shift_oracle_matches_reference_vectors():
    """stereo_shift_torch (reference stereo_utils.py:15-88): the numpy restatement against outputs captured from the
    imported reference (latent-shaped inputs, both shift modes, exponents 1 / 2 / 0.5, a flat depth)."""
    from conftest import Golden
    from oracle import stereo_shift_oracle as so
    g = Golden("stereo_shift.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        got = so.stereo_shift(g[f"{cid}/x"], g[f"{cid}/d"], case["scale_factor"], case["shift_both"], case["exponent"])
        assert got.shape == g[f"{cid}/out"].shape
        assert np.array_equal(got, g[f"{cid}/out"]), cid


def test_node_oracle_matches_the_round2_node_vectors(golden_node_extra):
    """node_oracle.generate against the reference's outputs for the API-only modes (left-only, only-right,
    cyan-red-reverseanaglyph), the three fill strings outside the combo list and depth maps of another size."""
    from conftest import extra_case_expected, extra_case_inputs
    from oracle import node_oracle
    g = golden_node_extra
    for case in g.meta["cases"]:
        img, depth = extra_case_inputs(g, case)
        got = node_oracle.generate(img, depth, case["divergence"], case["separation"], case["mode"], case["balance"],
                                   case["convergence"], case["exponent"], case["fill_ui"], case["edge_threshold"], case["strength"],
                                   case["blur"], **case["kw"])
        want = extra_case_expected(g, case)
        cid = case["id"]
        assert list(got[0].shape) == case["shapes"]["stereo"] and list(got[3].shape) == case["shapes"]["mask"], cid
        if case["gpu"]:
            assert np.abs(got[0] - want[0]).max() <= 1e-4, cid
            assert np.allclose(got[1][..., 0], want[1], atol=1e-6) and np.allclose(got[2][..., 0], want[2], atol=1e-6), cid
        else:
            assert np.array_equal(got[0], want[0]), cid
            assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        assert np.array_equal(got[3], want[3]), cid


def _warp_1080p_inputs(case):
    h, w = case["h"], case["w"]
    img = synth.image_f32(1, h, w, seed=case["image_seed"]).transpose(0, 3, 1, 2).copy()
    return img, synth.stepped(h, w)[None] * np.float32(255.0)


def test_forward_warp_1080p_rows():
    """forward_warp_gpu at 1080p against rows captured from the reference: gap mask of the whole frame exact, colours to
    the last ulps outside the gaps (the coordinate round trip and torch.linspace are reproduced bit for bit)."""
    from conftest import Golden
    g = Golden("forward_warp_1080p.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        img, depth = _warp_1080p_inputs(case)
        warped, mask = oracle.forward_warp_gpu(img, depth, case["divergence_px"], case["separation_px"], case["exponent"],
                                               case["convergence"])
        want_mask = np.unpackbits(g[f"{cid}/mask"])[: mask.size].reshape(mask.shape).astype(bool)
        assert np.array_equal(mask, want_mask), cid
        rows = case["rows"]
        assert_warp_colours(warped[:, :, rows, :], g[f"{cid}/rows"], want_mask[:, rows, :], cid)


def test_digests_at_baseline_sizes():
    """Node outputs at BASELINE.json sizes (cfg 1: 512 x 512 naive_interpolating; cfg 2: 1080p polylines_soft; cfg 3 at a
    quarter: 1080p hybrid_edge; all with the depth blur on) against SHA-256 digests of the reference's outputs."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests.json")))
    for cid, c in dig.items():
        img = synth.image_u8(c["h"], c["w"], seed=c["image_seed"], hazards=False)[None].astype(np.float32) / np.float32(255.0)
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
        assert sha(k[0]) == c["stereo_u8"], cid
        assert sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid


def test_digest_of_the_metric_frame_at_4k():
    """The metric's own frame (BASELINE.json configs[1]: 4K, polylines_soft, left-right SBS, divergence 8, stepped depth, blur on) against
    SHA-256 digests of the REFERENCE node's outputs (tests/golden/digest_metric_4k.json, tools/make_goldens.py --only-metric-4k: a quarter
    of an hour of the pure-Python reference): uint8 codes, mask AND the float32 arrays themselves."""
    import json
    import os
    from conftest import GOLDEN
    c = json.load(open(os.path.join(GOLDEN, "digest_metric_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
    depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
    got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                               depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
    assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"]
    assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"]
    assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"]


def test_digests_of_the_other_bench_configurations_at_4k():
    """One 4K frame of every other bench configuration (BASELINE.json configs[2]: hybrid_edge + blur, configs[4]: `none` as a red-cyan
    anaglyph with its mask; naive_interpolating and polylines_sharp at the metric's size) against SHA-256 digests of the REFERENCE node's
    outputs (tests/golden/digests_4k.json, tools/make_goldens.py --only-4k): uint8 codes, mask and the float32 arrays."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert len(dig) == 4
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_order_dependent_depth_at_full_width():
    """Order-dependent rows against the REFERENCE node's own outputs (tests/golden/digests_ties.json, tools/make_goldens.py --only-ties): a 4K
    frame of depth saturated to exact 0 / 1 (exact closeness ties under every fold) and 48 rows of 8-bit noise (no reset points),
    polylines_soft and polylines_sharp, divergence 8, blur off -- where the result depends on the ORDER of the reference's active list."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests_ties.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert len(dig) == 4
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_scene8_depth_at_4k():
    """One 4K frame of scene8 depth (tools/synth.scene8: quantised smooth depth with softened object silhouettes -- what a depth estimator
    delivers, and what overflows the tile kernels' per-pixel lists) at the metric's divergence against the REFERENCE node's own outputs
    (tests/golden/digests_scene8_4k.json, tools/make_goldens.py --only-scene8-4k): polylines_soft with the blur, polylines_sharp with and
    without."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests_scene8_4k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert len(dig) == 3
    img = synth.image_f32(1, 2160, 3840, seed=1)
    depth = synth.depth_batch("scene8", 1, 2160, 3840, channels=3)
    for cid, c in dig.items():
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid


def test_digests_of_8k_frames():
    """One 8K frame (7680 x 4320), divergence 8, blur on, against the REFERENCE node's own outputs (tests/golden/digests_8k.json,
    tools/make_goldens.py --only-8k: half an hour of the pure-Python reference per case): polylines_soft side by side and naive_interpolating
    as a red-cyan anaglyph."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests_8k.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert len(dig) == 2
    img = synth.image_f32(1, 4320, 7680, seed=1)
    depth = synth.depth_batch("stepped", 1, 4320, 7680, channels=3)
    for cid, c in dig.items():
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid
        assert sha(got[0]) == c["stereo_f32"] and sha(got[1]) == c["dl_f32"] and sha(got[2]) == c["dr_f32"], cid
        del got, k


def test_digests_at_the_widths_round_6_opened():
    """Thin rows at the widths round 6 opened (anaglyphs of the forward fills beyond the row kernel's stash form, naive /
    naive_interpolating / none_post / inverse_post at their new limits, polylines_sharp at 8 192 columns) against SHA-256 digests of the
    REFERENCE node's outputs (tests/golden/digests_wide.json, tools/make_goldens.py --only-wide): the oracle is pinned there too."""
    import json
    import os
    from conftest import GOLDEN
    dig = json.load(open(os.path.join(GOLDEN, "digests_wide.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert len(dig) >= 14
    for cid, c in dig.items():
        img = synth.image_f32(1, c["h"], c["w"], seed=c["image_seed"])
        img[:, :, c["black"][0]:c["black"][1]] = 0.0
        depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
        got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                                   depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
        assert sha(k[0]) == c["stereo_u8"], cid
        assert sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"], cid
        assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"], cid


def test_node_oracle_on_scene8_depth(golden_scene8):
    """Depth of the kind the reference's own fixture maker draws (/root/reference/create_test_images.py:3-77): an 8-bit gradient with
    flat ellipses, hard and softened silhouettes (tools/synth.scene8) -- every UI technique, blur off and on, against the reference
    node's outputs (tools/make_goldens.py --only-scene8); plus the digests of one 1080p frame."""
    from conftest import scene8_case_expected, scene8_case_inputs
    g = golden_scene8
    for case in g.meta["cases"]:
        img, depth = scene8_case_inputs(g, case)
        ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[case["fill"]]
        got = node_oracle.generate(img, depth, case["divergence"], case["separation"], case["mode"], case["balance"],
                                   case["convergence"], case["exponent"], ui, case["edge_threshold"], case["strength"],
                                   case["blur"], **case["kw"])
        want = scene8_case_expected(g, case)
        cid = case["id"]
        if case["fill"] == "gpu_warp":   # (the fixture keeps every 8th row of the float32 colours)
            assert np.abs(got[0][:, ::case["row_step"]] - want[0]).max() <= GPU_WARP_COLOUR_TOL, (cid, np.abs(got[0][:, ::case["row_step"]] - want[0]).max())
        else:
            assert np.array_equal(got[0], want[0]), cid
        assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), cid
        assert np.array_equal(got[3], want[3]), cid
    c = g.meta["digest_1080p"]
    img = synth.image_u8(c["h"], c["w"], seed=c["image_seed"], hazards=False)[None].astype(np.float32) / np.float32(255.0)
    depth = synth.depth_batch(c["kind"], 1, c["h"], c["w"], channels=3)
    got = node_oracle.generate(img, depth, c["divergence"], 0.0, c["mode"], 0.0, 0.5, 2.0, c["fill_ui"], 20.0, 20.0, c["blur"],
                               depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    k = [np.round(a * 255.0).astype(np.uint8) for a in (got[0], got[1][..., 0], got[2][..., 0])]
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert sha(k[0]) == c["stereo_u8"] and sha(k[1]) == c["dl_u8"] and sha(k[2]) == c["dr_u8"]
    assert sha(np.packbits(got[3].astype(bool))) == c["mask"] and int(got[3].sum()) == c["mask_sum"]
