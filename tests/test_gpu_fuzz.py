"""Seeded fuzz of the HIP kernels against the oracle on inputs built to hit the order-dependent corners:
few quantised depth levels placed symmetrically around the convergence point (exact |disparity| ties between layers),
plateaus with zero disparity, single-pixel spikes, separations that push points off either border.  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
FILLS = ["none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp", "inverse", "hybrid_edge", "none_post",
         "inverse_post", "hybrid_edge_plus"]


@pytest.fixture(scope="module")
def engine():
    from comfystereo_amd import engine as e
    assert torch.cuda.is_available()
    return e


def make_case(rng):
    h = int(rng.integers(1, 5))
    w = int(rng.choice([9, 31, 64, 65, 127, 200, 333, 513, 700]))
    nlev = int(rng.integers(2, 7))
    levels = np.sort(rng.choice(np.arange(0, 256, 15), size=nlev, replace=False)).astype(np.float32)
    kind = rng.integers(0, 4)
    if kind == 0:  # piecewise constant runs
        runs = rng.integers(1, max(2, w // 6), size=w)
        idx = np.repeat(rng.integers(0, nlev, size=w), runs)[:w]
        depth = np.tile(levels[idx], (h, 1))
    elif kind == 1:  # symmetric two-level pattern around the convergence point + spikes
        depth = np.where(rng.random((h, w)) < 0.5, levels[0], levels[-1]).astype(np.float32)
    elif kind == 2:  # ramps joined by jumps
        x = np.linspace(0, 1, w, dtype=np.float64)
        depth = np.tile((np.floor(x * nlev) / nlev * 200 + 40 * ((x * 7) % 1)).astype(np.float32), (h, 1))
    else:  # noise on levels
        depth = levels[rng.integers(0, nlev, size=(h, w))]
    depth = depth.astype(np.float32).copy()
    if rng.random() < 0.3:
        depth[rng.integers(0, h), rng.integers(0, w)] = 255.0
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.random() < 0.5:
        m = rng.random((h, w)) < 0.1
        img[m] = rng.choice([[0, 0, 0], [128, 128, 0], [255, 1, 0]])
    div = float(rng.choice([-9.0, -4.5, -1.0, 0.7, 3.0, 6.5, 12.0]))
    sep = float(rng.choice([0.0, 0.0, -2.0, 1.5, 4.0]))
    e = float(rng.choice([0.3, 0.5, 1.0, 1.3, 2.0]))
    conv = float(rng.choice([0.0, 0.25, 0.5, 0.5, 0.75, 1.0]))
    return img, depth, div, sep, e, conv


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_all_fills(engine, seed):
    rng = np.random.default_rng(9000 + seed)
    for _ in range(12):
        img, depth, div, sep, e, conv = make_case(rng)
        for fill in FILLS:
            try:
                want = oracle.apply_stereo_divergence(img, depth, div, sep, e, fill, conv)
            except IndexError:  # the reference's csg scratch would overflow for this input: behaviour undefined there
                continue
            got = engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), div, sep, e,
                                                 fill, conv).cpu().numpy()
            bad = np.argwhere(got != want)
            assert bad.size == 0, (seed, fill, img.shape, div, sep, e, conv, len(bad), bad[:3].tolist())


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_dialect_d64(engine, seed):
    """The same corner-case generator under dialect D64 (float64 disparity chain + int64 pixel sums): HIP vs the oracle in the
    same dialect, bit for bit, for the techniques that have the instantiation."""
    rng = np.random.default_rng(12000 + seed)
    oracle.set_dialect("D64")
    try:
        for _ in range(12):
            img, depth, div, sep, e, conv = make_case(rng)
            for fill in FILLS:   # (round 5: every CPU technique has the dialect, the three hidden ones included)
                try:
                    want = oracle.apply_stereo_divergence(img, depth, div, sep, e, fill, conv)
                except IndexError:  # (polylines: the reference's csg scratch would overflow for this input)
                    continue
                got = engine.apply_stereo_divergence(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda(), div, sep, e,
                                                     fill, conv, dialect="D64").cpu().numpy()
                bad = np.argwhere(got != want)
                assert bad.size == 0, (seed, fill, img.shape, div, sep, e, conv, len(bad), bad[:3].tolist())
    finally:
        oracle.set_dialect("D32")


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_mesh_warp(engine, seed):
    """forward_warp_mesh on the same hostile depth maps (plateaus, spikes, noise), batches of 1-3 frames (the any-frame
    culling rule), both signs of the divergence: HIP vs its specification in the oracle."""
    rng = np.random.default_rng(15000 + seed)
    for _ in range(6):
        b = int(rng.integers(1, 4))
        cases = [make_case(rng) for _ in range(b)]
        h = max(2, min(c[1].shape[0] for c in cases) + 1)
        w = min(c[1].shape[1] for c in cases)
        dep = np.stack([np.resize(c[1][:, :w], (h, w)) for c in cases]).astype(np.float32)
        img = rng.random((b, 3, h, w), dtype=np.float32)
        _, _, div, sep, e, conv = cases[0]
        div_px, sep_px = div / 100.0 * w, sep / 100.0 * w
        want, wmask = oracle.forward_warp_mesh(img, dep, div_px, sep_px, e, conv)
        got, gmask = engine.forward_warp_mesh(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), div_px, sep_px, e, conv)
        gmask = gmask.cpu().numpy()
        # (a pixel centre within rounding of a triangle edge may be covered on one side only: float32 spans on both sides,
        # but the exponent path differs -- torch.pow semantics on the device, libm on the host -- for e not in {0.5, 1, 2})
        tol = 2e-3 if e in (0.5, 1.0, 2.0) else 2e-2
        assert (gmask != wmask).mean() <= tol, (seed, img.shape, div, sep, e, conv, float((gmask != wmask).mean()))
        same = (gmask == wmask)[:, None].repeat(3, 1)
        diff = np.abs(got.cpu().numpy() - want)
        assert (diff[same] > 1e-5).mean() <= 5 * tol, (seed, img.shape, float(diff[same].max()))
