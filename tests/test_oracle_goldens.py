"""The CPU oracle against every golden vector captured from the imported reference (not gpu)."""
import hashlib

import numpy as np
import pytest

import synth
from conftest import node_case_expected, node_case_inputs
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
            assert err.max() <= GPU_WARP_COLOUR_TOL, (cid, err.max())
        else:
            assert np.quantile(err, 0.999) <= 1e-3, cid


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
