"""Dialect D64 on the GPU (cs_apply_stereo_divergence2 / cs_params.flags bits 3, 4): the float64 disparity chain against the
fixture generated from the reference's inner functions (tests/golden/dialect_f64.npz) and full D64 against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle, oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "dialect_f64.npz")
# the float64 chain (pinned by the fixture) runs in every technique's tile kernel since round 5; the numba typing of the polylines sweep
# (full D64) is derived -- checked against the oracle's statement of the same rules -- and runs in the point kernel since round 6
# (k_polypoint<..., DIA, SW>; the general row kernel takes the rows it flags and the geometries it does not cover)
FILLS = ("none", "naive", "naive_interpolating", "inverse", "polylines_soft", "polylines_sharp", "hybrid_edge",
         "none_post", "inverse_post", "hybrid_edge_plus")


def _gpu(img, depth, c, fill, dialect):
    from comfystereo_amd import engine
    return engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), c["divergence"],
                                          c["separation"], c["exponent"], fill, c["convergence"], dialect=dialect).cpu().numpy()


def test_f64_disparity_chain_matches_the_reference_fixture():
    z = np.load(GOLD)
    for c in json.loads(str(z["meta"]))["cases"]:
        for fill in FILLS:
            got = _gpu(z[f"{c['id']}/img"], z[f"{c['id']}/depth"], c, fill, "f64-disparity")
            np.testing.assert_array_equal(got, z[f"{c['id']}/{fill}"], err_msg=f"{c['id']}/{fill}")


def test_d64_matches_the_oracle_and_differs_from_d32_where_it_should():
    z = np.load(GOLD)
    cases = json.loads(str(z["meta"]))["cases"]
    for c in cases:
        img, depth = z[f"{c['id']}/img"], z[f"{c['id']}/depth"]
        for fill in FILLS:
            oracle.set_dialect("D64")
            try:
                want = oracle.apply_stereo_divergence(img, depth, c["divergence"], c["separation"], c["exponent"], fill, c["convergence"])
            finally:
                oracle.set_dialect("D32")
            np.testing.assert_array_equal(_gpu(img, depth, c, fill, "D64"), want, err_msg=f"{c['id']}/{fill}")
    c = cases[-1]   # the near-integer case: the dialects disagree on many pixels
    a, b = _gpu(z["5/img"], z["5/depth"], c, "none", "D32"), _gpu(z["5/img"], z["5/depth"], c, "none", "D64")
    assert int((a != b).any(-1).sum()) == c["pixels_differing_from_d32"]


@pytest.mark.parametrize("dialect", ["f64-disparity", "int64-sum", "D64"])
@pytest.mark.parametrize("fill", ["none_post", "inverse_post", "hybrid_edge_plus"])
def test_hidden_techniques_run_the_dialect(fill, dialect):
    """Round 5 (was: test_other_techniques_refuse_the_dialect): the three techniques no UI string reaches take the dialect bits too.
    none_post / inverse_post: their mapping functions are @njit (reference :1662, :1688) -- the float64 offset chain is all numba
    changes, the np.interp post-fill is plain numpy in both installs; hybrid_edge_plus = hybrid_edge + polylines_soft, both of
    which have the dialect.  HIP vs the oracle under the same setting (derived typing, like every D64 statement)."""
    from comfystereo_amd import engine
    rs = np.random.RandomState(41)
    h, w = 8, 700
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    img[:, 100:130] = 0
    depth = synth.depth_batch("blobs", 1, h, w, channels=1)[0, ..., 0].astype(np.float32)
    depth[3] = np.round(depth[3] * 5) / 5
    depth[5] = rs.randint(0, 256, w).astype(np.float32) / 255.0
    c = dict(divergence=5.0, separation=0.5, exponent=1.3, convergence=0.5)
    oracle.set_dialect(dialect)
    try:
        want = oracle.apply_stereo_divergence(img, depth, 5.0, 0.5, 1.3, fill, 0.5)
    finally:
        oracle.set_dialect("D32")
    np.testing.assert_array_equal(_gpu(img, depth, c, fill, dialect), want)


def test_gpu_warp_has_no_dialect():
    """What genuinely has no D64: gpu_warp is torch arithmetic in both installs -- the flag is refused, not ignored."""
    from comfystereo_amd import engine
    n, h, w = 1, 16, 64
    img = torch.rand((n, h, w, 3), device="cuda")
    dep = torch.rand((n, h, w, 3), device="cuda")
    engine.DIALECT = "D64"
    try:
        p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", "left-right", 5.0, 0.0, 0.0, 0.5, 2.0, False, 6.0, 6.0, 1.0, 0, 4)
        with pytest.raises(RuntimeError, match="D64"):
            engine.Plan(p, torch.device("cuda")).run(img, dep)
    finally:
        engine.DIALECT = "D32"


def test_node_path_with_the_dialect_switch():
    """engine.DIALECT -> cs_params.flags bits 3 / 4 through cs_generate (SBS + anaglyph, uint8-origin image with hazards)."""
    from comfystereo_amd import engine
    n, h, w = 2, 40, 96
    img = synth.image_f32(n, h, w, seed=9)
    depth = synth.depth_batch("random8", n, h, w, channels=3)
    engine.DIALECT = "D64"
    oracle.set_dialect("D64")
    try:
        for ui, mode in (("Fill - Naive interpolating", "left-right"), ("No fill - Reverse projection", "red-cyan-anaglyph"),
                         ("Fill - Naive", "top-bottom"), ("Fill - Polylines Soft", "left-right"),
                         ("Fill - Polylines Sharp", "red-cyan-anaglyph"), ("Fill - Polylines Soft", "bottom-top"),
                         ("Imperfect fill - Hybrid Edge", "left-right"), ("Imperfect fill - Hybrid Edge", "red-cyan-anaglyph")):
            fill = node_oracle.FILL_KEYS[ui]
            p = engine.make_params(n, h, w, h, w, 3, fill, mode, 7.0, 0.5, 0.0, 0.5, 1.3, False, 6.0, 6.0, 1.0, 0, 4)
            assert (p.flags >> 3) & 3 == 3
            got = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
            want = node_oracle.generate(img, depth, 7.0, 0.5, mode, 0.0, 0.5, 1.3, ui, 6.0, 6.0, False)
            for g, wv in zip(got, want):
                np.testing.assert_array_equal(g, wv)
    finally:
        engine.DIALECT = "D32"
        oracle.set_dialect("D32")


@pytest.mark.parametrize("dialect", ["f64-disparity", "int64-sum", "D64"])
@pytest.mark.parametrize("fill", ["polylines_soft", "polylines_sharp"])
def test_polylines_dialects_on_a_wide_noisy_row(fill, dialect):
    """1600 columns (the float64 coordinates take 8 more bytes of LDS per column), depth with occlusion folds and ties: every
    dialect bit alone and both, HIP vs the oracle under the same setting."""
    rs = np.random.RandomState(23)
    h, w = 6, 1600
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    depth = synth.depth_batch("blobs", 1, h, w, channels=1)[0, ..., 0].astype(np.float32)
    depth[2] = np.round(depth[2] * 4) / 4          # plateaus: equal |disparity| on both sides of a fold
    depth[3] = rs.randint(0, 256, w).astype(np.float32) / 255.0
    c = dict(divergence=5.0, separation=0.5, exponent=1.3, convergence=0.5)
    oracle.set_dialect(dialect)
    try:
        want = oracle.apply_stereo_divergence(img, depth, 5.0, 0.5, 1.3, fill, 0.5)
    finally:
        oracle.set_dialect("D32")
    np.testing.assert_array_equal(_gpu(img, depth, c, fill, dialect), want)


def test_polylines_dialect_width_limit():
    """The float64 coordinates need 8 w more bytes of LDS: frames the D32 row kernel still takes are refused with CS_ELIMIT."""
    from comfystereo_amd import _native, engine
    wmax = _native.lib().cs_max_width(engine.FILL["polylines_sharp"])
    img = torch.zeros((1, 2, wmax, 3), dtype=torch.uint8, device="cuda")
    dep = torch.rand((1, 2, wmax), device="cuda")
    engine.apply_stereo_divergence(img, dep, 1.0, 0.0, 1.0, "polylines_sharp", 0.5)          # D32: accepted
    with pytest.raises(RuntimeError, match="too wide"):
        engine.apply_stereo_divergence(img, dep, 1.0, 0.0, 1.0, "polylines_sharp", 0.5, dialect="D64")


@pytest.mark.parametrize("dialect", ["f64-disparity", "int64-sum", "D64"])
@pytest.mark.parametrize("exponent", [2.0, 1.0, 1.3])
def test_forward_fills_run_the_dialect_in_the_tile_kernel(dialect, exponent):
    """Round 5: none / naive / naive_interpolating / inverse / hybrid_edge of the node path under the dialect bits run in the tile
    kernels' own dialect instantiations (float64 offset chain -> int() / floor(); int64 pixel sums) instead of the whole-row kernel: rows wide enough
    for several tiles, depth with holes longer than a tile's window margin and black / (128, 128, 0) pixels (quirk Q5 is where
    the int64-sum bit changes the result), every mode family.  HIP vs the oracle under the same setting, bit for bit."""
    from comfystereo_amd import engine
    n, h, w = 2, 12, 2100
    img = synth.image_f32(n, h, w, seed=19)
    img[:, :, 300:340] = 0.0                                   # genuinely black pixels
    img[:, 3::4, 900:960] = np.array([128, 128, 0], np.float32) / 255.0   # uint8 sum wraps to 0
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    depth[1] = synth.depth_batch("random8", 1, h, w, channels=3)[0]
    engine.DIALECT = dialect
    oracle.set_dialect(dialect)
    try:
        for ui, mode in (("Fill - Naive interpolating", "left-right"), ("No fill - Reverse projection", "red-cyan-anaglyph"),
                         ("Fill - Naive", "top-bottom"), ("No fill", "right-left"), ("Fill - Naive interpolating", "cyan-red-reverseanaglyph"),
                         ("Imperfect fill - Hybrid Edge", "left-right"), ("Imperfect fill - Hybrid Edge", "bottom-top")):
            # (hybrid_edge: k_hybrid_splat_tile's dialect instantiation -- dest_x, its distance to the column and the exp argument in
            # float64, the weight sum adds in float64 under the second bit)
            fill = node_oracle.FILL_KEYS[ui]
            p = engine.make_params(n, h, w, h, w, 3, fill, mode, 6.0, 0.3, 0.1, 0.5, exponent, False, 6.0, 6.0, 1.0, 0, 4)
            plan = engine.Plan(p, torch.device("cuda"))
            got = [t.cpu().numpy() for t in plan.run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
            assert int(plan.stats()[:, 9].sum()) == 0
            want = node_oracle.generate(img, depth, 6.0, 0.3, mode, 0.1, 0.5, exponent, ui, 6.0, 6.0, False)
            for g, wv, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
                np.testing.assert_array_equal(g, wv, err_msg=f"{ui} {mode} {name}")
    finally:
        engine.DIALECT = "D32"
        oracle.set_dialect("D32")


@pytest.mark.parametrize("exponent", [2.0, 1.3])
@pytest.mark.parametrize("kind", ["stepped", "clipped", "blobs"])
@pytest.mark.parametrize("fill", ["polylines_soft", "polylines_sharp"])
def test_polylines_float64_chain_in_the_tile_kernel(fill, exponent, kind):
    """Round 5: the float64 disparity chain (dialect bit 0 alone -- the half of numba's typing that the reference itself pins,
    tests/golden/dialect_f64.npz) for the polylines techniques runs in k_polypoint's dialect instantiations (staging in float64, the
    point x rounded once; sharp: both points (float)(x64 -+ 0.45) of every source in a second LDS array); rows it flags (`clipped`:
    exact ties) go to the row kernel's dialect instantiation.  Several tiles per row, side by side and as an anaglyph.  HIP vs the
    oracle under the same setting, bit for bit."""
    from comfystereo_amd import engine
    n, h, w = 2, 10, 2100
    img = synth.image_f32(n, h, w, seed=29)
    depth = synth.depth_batch(kind, n, h, w, channels=3)
    engine.DIALECT = "f64-disparity"
    oracle.set_dialect("f64-disparity")
    try:
        for mode in ("left-right", "red-cyan-anaglyph"):
            p = engine.make_params(n, h, w, h, w, 3, fill, mode, 6.0, 0.3, 0.1, 0.5, exponent, False, 6.0, 6.0, 1.0, 0, 4)
            assert (p.flags >> 3) & 3 == 1
            plan = engine.Plan(p, torch.device("cuda"))
            got = [t.cpu().numpy() for t in plan.run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
            st = plan.stats()
            assert int(st[:, 9].sum()) == 0
            if kind != "clipped":
                assert int(st[:, 11].sum()) < n * h, "every row went back to the row kernel: the tile kernel did not take the call"
            ui = "Fill - Polylines Soft" if fill == "polylines_soft" else "Fill - Polylines Sharp"
            want = node_oracle.generate(img, depth, 6.0, 0.3, mode, 0.1, 0.5, exponent, ui, 6.0, 6.0, False)
            for g, wv, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
                np.testing.assert_array_equal(g, wv, err_msg=f"{fill} {mode} {name}")
    finally:
        engine.DIALECT = "D32"
        oracle.set_dialect("D32")


@pytest.mark.parametrize("exponent", [2.0, 1.0])
def test_exact_powers_under_the_float64_chain(dev_switch, exponent):
    """`abs(d) ** e` for e = 2 and e = 1 is exact in float64, and offsets that are whole numbers sit on int()'s boundary: depth
    levels that normalise to -0.25 / 0.25 / 0.75 and a divergence_px one ulp off a whole number (-14 % of 1 600 columns =
    -224.00000000000003) put the float64 offset a hair on the other side of the integer the float32 chain (and a pow() that is one ulp
    short: the device library's pow(0.25, 2.0)) lands on.  The dialect fuzz found the row kernels one pixel off on such a row
    (round 5); every technique, tile kernels and row kernels alike, against the oracle."""
    h, w = 3, 1600
    rs = np.random.RandomState(77)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    depth = rs.choice(np.array([0.0, 120.0, 240.0], np.float32), size=(h, w // 8)).repeat(8, axis=1)
    depth[:, :8] = 0.0; depth[:, -8:] = 240.0
    c = dict(divergence=-14.0 if exponent == 2.0 else -7.0, separation=-1.0 if exponent == 2.0 else -2.0, exponent=exponent, convergence=0.25)
    for dialect in ("f64-disparity", "D64"):
        oracle.set_dialect(dialect)
        try:
            want = {f: oracle.apply_stereo_divergence(img, depth, c["divergence"], c["separation"], exponent, f, c["convergence"]) for f in FILLS}
        finally:
            oracle.set_dialect("D32")
        d32 = oracle.apply_stereo_divergence(img, depth, c["divergence"], c["separation"], exponent, "none", c["convergence"])
        assert (want["none"] != d32).any(), "the case does not separate the dialects"
        for no_tile in (0, 1):
            dev_switch("no_tile", no_tile)
            for f in FILLS:
                np.testing.assert_array_equal(_gpu(img, depth, c, f, dialect), want[f], err_msg=f"{dialect} {f} no_tile={no_tile}")


@pytest.mark.parametrize("dialect", ["int64-sum", "D64"])
@pytest.mark.parametrize("fill", ["polylines_soft", "polylines_sharp"])
def test_numba_sweep_runs_in_the_point_kernel(fill, dialect, dev_switch):
    """Round 6: numba's typing of the polylines SWEEP (reference :1951-1991 under @njit -- from / to / length / centre, the
    segment parameter, the closeness and every colour term in float64, the sums rounded to float32 piece by piece) in the point kernel
    (k_polypoint<..., DIA, SW>: fast path, bridges, chain path, general search) instead of the general row kernel only.  4K-wide bands of
    scene8 depth (hard and softened silhouettes: folds, bridges, several points per pixel) at the metric's divergence plus a row of exact
    ties, SBS and anaglyph, exponents 2 and 1.3: HIP vs the oracle under the same setting, bit for bit -- and the same bits from the row
    kernel alone (no_tile).  That the tile kernel took the call shows in ST_TILE_REDO_ROWS: only a tile kernel flags rows, and it must
    flag some (the tie row, list overflows on the silhouettes) and not all of them."""
    from comfystereo_amd import engine
    n, h, w = 2, 40, 3840
    img = synth.image_f32(n, h, w, seed=83)
    depth = np.stack([synth.scene8(2160, w, seed=s, soften=bool(s & 1))[y0:y0 + h] for s, y0 in ((2, 740), (5, 1490))]).astype(np.float32)
    depth[0, 7] = np.where((np.arange(w) // 96) & 1, 0.75, 0.25)   # plateaus with equal |disparity| on both sides of every fold: exact ties
    depth[1, 9] = np.random.RandomState(5).randint(0, 256, w).astype(np.float32) / 255.0   # 8-bit noise: dozens of layers per pixel
    depth = depth[..., None].repeat(3, -1)
    engine.DIALECT = dialect
    oracle.set_dialect(dialect)
    try:
        for mode, e in (("left-right", 2.0), ("red-cyan-anaglyph", 1.3), ("top-bottom", 1.0)):
            ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
            want = node_oracle.generate(img, depth, 8.0, 0.0, mode, 0.0, 0.5, e, ui, 20.0, 20.0, False, batch_size=12)
            p = engine.make_params(n, h, w, h, w, 3, fill, mode, 8.0, 0.0, 0.0, 0.5, e, False, 20.0, 20.0, 1.0, 0, 12)
            assert (p.flags >> 3) & 2
            for no_tile in (0, 1):
                dev_switch("no_tile", no_tile)
                plan = engine.Plan(p, torch.device("cuda"))
                got = [t.cpu().numpy() for t in plan.run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
                st = plan.stats()
                assert int(st[:, 9].sum()) == 0
                for g, wv, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
                    np.testing.assert_array_equal(g, wv, err_msg=f"{fill}/{dialect}/{mode}/no_tile={no_tile}/{name}")
                redo = int(st[:, 11].sum())
                if no_tile:
                    assert redo == 0
                else:
                    assert 0 < redo < n * h, redo
    finally:
        engine.DIALECT = "D32"
        oracle.set_dialect("D32")
