"""Dialect D64 (numba typing) -- the part of it that can be pinned here: the float64 disparity chain.
tests/golden/dialect_f64.npz holds what the reference's own inner functions return for normalized_depth.astype(float64)
(tools/make_goldens.py --only-dialect); pixel sums there still wrap (no numba in this image), so the fixture corresponds to
the oracle's "f64-disparity" setting; the int64 pixel sums of full D64 are derived from numba's typing rules and only
checked for what they must change.  Runs without a GPU."""
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dialect_f64.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return z, json.loads(str(z["meta"]))["cases"]


def _run(z, c, fill, dialect):
    oracle.set_dialect(dialect)
    try:
        return oracle.apply_stereo_divergence(z[f"{c['id']}/img"], z[f"{c['id']}/depth"], c["divergence"], c["separation"],
                                              c["exponent"], fill, c["convergence"])
    finally:
        oracle.set_dialect("D32")


def test_f64_disparity_chain_matches_the_reference_inner_functions(gold):
    z, cases = gold
    for c in cases:
        for fill in ("none", "naive", "naive_interpolating", "inverse"):
            np.testing.assert_array_equal(_run(z, c, fill, "f64-disparity"), z[f"{c['id']}/{fill}"], err_msg=f"{c['id']}/{fill}")


def test_f64_disparity_chain_polylines(gold):
    """polylines_soft / sharp: the point coordinates leave the float64 chain and are rounded once when they are stored into
    the reference's float32 `pt` array (:1924-1934); the sweep keeps the no-numba typing.  Pinned bit for bit."""
    z, cases = gold
    differs = 0
    for c in cases:
        for fill in ("polylines_soft", "polylines_sharp"):
            got = _run(z, c, fill, "f64-disparity")
            np.testing.assert_array_equal(got, z[f"{c['id']}/{fill}"], err_msg=f"{c['id']}/{fill}")
            differs += int((got != _run(z, c, fill, "D32")).any(-1).sum())
    assert differs > 0   # (the two chains do not give the same frames)


def test_f64_disparity_chain_hybrid_edge(gold):
    """hybrid_edge: dest_x, its distance to the column and the exp argument in float64 (:1636-1644); pinned bit for bit.  The
    numba typing on top of it (float64 weight sums, derived) may move a colour by a code here and there."""
    z, cases = gold
    differs = 0
    for c in cases:
        got = _run(z, c, "hybrid_edge", "f64-disparity")
        np.testing.assert_array_equal(got, z[f"{c['id']}/hybrid_edge"], err_msg=c["id"])
        differs += int((got != _run(z, c, "hybrid_edge", "D32")).any(-1).sum())
        d = np.abs(got.astype(np.int32) - _run(z, c, "hybrid_edge", "D64").astype(np.int32)).max(-1)
        assert (d > 1).mean() < 2e-3, c["id"]
    assert differs > 0


def test_f64_disparity_chain_hidden_techniques(gold):
    """Round 5: none_post / inverse_post / hybrid_edge_plus (no UI string reaches them) under the float64 chain -- what the
    reference's own functions return for normalized_depth.astype(float64) (their mapping functions are @njit: this chain is all
    numba changes for the two `_post` techniques).  Pinned bit for bit."""
    z, cases = gold
    differs = 0
    for c in cases:
        for fill in ("none_post", "inverse_post", "hybrid_edge_plus"):
            got = _run(z, c, fill, "f64-disparity")
            np.testing.assert_array_equal(got, z[f"{c['id']}/{fill}"], err_msg=f"{c['id']}/{fill}")
            differs += int((got != _run(z, c, fill, "D32")).any(-1).sum())
    assert differs > 0


def test_numba_sweep_typing_is_close_to_the_float32_sweep(gold):
    """Full D64 for polylines (float64 sub-intervals, interpolation and products: derived from numba's typing rules, not
    pinnable here): colours differ from the float64-chain / float32-sweep frames by one code on a minority of the pixels, and
    by more on a handful (a sub-interval whose epsilon float32 absorbs picks another segment at an occlusion boundary) -- a
    sanity bound on the derived arithmetic, not a parity claim."""
    z, cases = gold
    for c in cases:
        for fill in ("polylines_soft", "polylines_sharp"):
            a, b = _run(z, c, fill, "f64-disparity").astype(np.int32), _run(z, c, fill, "D64").astype(np.int32)
            d = np.abs(a - b).max(-1)
            assert (d > 1).mean() < 2e-3, (c["id"], fill)
            assert (d > 0).mean() < 0.2, (c["id"], fill)


def test_where_the_dialects_diverge(gold):
    """Report (and bound) the divergence: ordinary depth maps give identical frames -- a float32 disparity only truncates
    differently when it lies within an ulp of an integer -- the searched near-integer case differs in a fifth of its pixels."""
    z, cases = gold
    for c in cases:
        d32, f64 = _run(z, c, "none", "D32"), _run(z, c, "none", "f64-disparity")
        n = int((d32 != f64).any(-1).sum())
        assert n == c["pixels_differing_from_d32"]
        assert (n > 500) == (c["kind"] == "near-integer")


def test_int64_sums_only_change_pixels_whose_channels_sum_to_a_multiple_of_256(gold):
    z, cases = gold
    changed = 0
    for c in cases:
        a, b = _run(z, c, "naive_interpolating", "f64-disparity"), _run(z, c, "naive_interpolating", "D64")
        none = _run(z, c, "none", "D64")
        # with int64 sums a (128,128,0) pixel is not "black": rows without such pixels (and without true holes next to
        # them) are untouched by the change of the sum
        rows_with_hazard = ((none.astype(np.int64).sum(-1) % 256 == 0) & (none.astype(np.int64).sum(-1) > 0)).any(1)
        assert (a[~rows_with_hazard] == b[~rows_with_hazard]).all()
        changed += int((a != b).any(-1).sum())
    assert changed > 0
    # the forward map itself has no sums
    for c in cases[:2]:
        np.testing.assert_array_equal(_run(z, c, "none", "f64-disparity"), _run(z, c, "none", "D64"))
