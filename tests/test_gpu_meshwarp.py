"""The mesh-quality gpu_warp (reference stereoimage_generation.py:453-689, `forward_warp_mesh`; the reference's warp
whenever moderngl is importable, :1068-1071) against its specification in oracle/stereo_oracle.c.  No fixture of the
reference exists for this path (no OpenGL context in the image: parity unpinned, see the oracle's header), so next to the
HIP-vs-oracle comparison the tests check the properties the reference's construction guarantees."""
import numpy as np
import pytest
import torch

import synth
from oracle import node_oracle, oracle

pytestmark = pytest.mark.gpu

TOL = 2e-6   # float32 interpolation of values in 0..1: the same operations in the same order on both sides


def _inputs(b, h, w, kind, seed=3):
    rng = np.random.default_rng(seed)
    img = rng.random((b, 3, h, w), dtype=np.float32)
    dep = synth.depth_batch(kind, b, h, w, channels=1)[..., 0]
    return img, dep


@pytest.mark.parametrize("kind,div,sep,exp", [("stepped", 4.0, 0.0, 1.0), ("radial", -6.5, 1.25, 2.0), ("random8", 3.0, 0.0, 0.5),
                                              ("stepped", -9.0, -2.0, 1.0), ("radial", 12.0, 0.0, 1.7)])
def test_forward_warp_mesh_matches_the_specification(kind, div, sep, exp):
    from comfystereo_amd import engine
    img, dep = _inputs(3, 61, 97, kind)
    want, wmask = oracle.forward_warp_mesh(img, dep, div, sep, exp, 0.4)
    got, gmask = engine.forward_warp_mesh(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), div, sep, exp, 0.4)
    gmask = gmask.cpu().numpy()
    # a pixel whose centre lies within rounding of a triangle edge may be covered on one side only
    assert (gmask != wmask).mean() <= 2e-4
    same = (gmask == wmask)[:, None].repeat(3, 1)
    diff = np.abs(got.cpu().numpy() - want)
    assert (diff[same] > TOL).mean() <= 5e-4, float(diff[same].max())


def test_depth_scale_and_threshold_variants():
    from comfystereo_amd import engine
    img, dep = _inputs(2, 40, 64, "stepped")
    for d, thr in ((dep * 255.0, 1.5), (dep, 0.25), (dep, 40.0)):
        want, wmask = oracle.forward_warp_mesh(img, d, 5.0, 0.0, 1.0, 0.5, thr)
        got, gmask = engine.forward_warp_mesh(torch.from_numpy(img).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda(), 5.0, 0.0,
                                              1.0, 0.5, thr)
        assert (gmask.cpu().numpy() != wmask).mean() <= 5e-4
        ok = (gmask.cpu().numpy() == wmask)[:, None].repeat(3, 1)
        assert (np.abs(got.cpu().numpy() - want)[ok] > TOL).mean() <= 1e-3


def test_flat_depth_is_a_pure_shift():
    """Constant depth: every vertex moves by the same offset, the mesh stays regular -- the output is the source image
    resampled at a constant horizontal shift, without gaps except at the border the image moved away from."""
    from comfystereo_amd import engine
    h, w = 32, 80
    img = np.random.default_rng(0).random((1, 3, h, w), dtype=np.float32)
    dep = np.full((1, h, w), 0.3, np.float32)
    # flat depth normalises to 0 -> offset = -0.5 ** 1 * divergence_px (convergence 0.5)
    got, mask = engine.forward_warp_mesh(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), 6.0, 0.0, 1.0, 0.5)
    got, mask = got.cpu().numpy(), mask.cpu().numpy()
    assert mask[..., : w - 4].sum() == 0 and mask[..., w - 2:].all()   # shifted 3 px to the left: the right border uncovers
    # interior: the source mesh sampled at (u + 3, wy) -- barycentric inside triangle A (v00, v10, v01) or B (v11, v10, v01)
    sc, scy = (w - 1) / w, (h - 1) / h
    for k in (0, 7, h - 1):
        wy = (k + 0.5) * scy; r = int(np.floor(wy)); t = wy - r
        for px in (1, 20, w - 6):
            u = (px + 0.5) * sc + 3.0; x = int(np.floor(u)); f = u - x
            c00, c10, c01, c11 = img[0, :, r, x], img[0, :, r, x + 1], img[0, :, r + 1, x], img[0, :, r + 1, x + 1]
            left, diag, right = (1 - t) * c00 + t * c01, (1 - t) * c10 + t * c01, (1 - t) * c10 + t * c11
            if f < 1 - t:
                s_ = f / (1 - t); want = (1 - s_) * left + s_ * diag
            else:
                s_ = (f - (1 - t)) / t; want = (1 - s_) * diag + s_ * right
            np.testing.assert_allclose(got[0, :, k, px], want, atol=3e-5)


def test_nearer_surface_wins_and_gaps_fill_from_the_named_side():
    from comfystereo_amd import engine
    h, w = 16, 96
    img = np.zeros((1, 3, h, w), np.float32)
    img[:, 0, :, : w // 2] = 1.0          # left half red = far, right half green = near
    img[:, 1, :, w // 2:] = 1.0
    dep = np.zeros((1, h, w), np.float32); dep[..., w // 2:] = 1.0
    for div in (8.0, -8.0):
        got, mask = engine.forward_warp_mesh(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), div, 0.0, 1.0, 0.5)
        got, mask = got.cpu().numpy(), mask.cpu().numpy()
        # far half moves by -div/2, near half by +div/2: div > 0 opens a gap at the seam, div < 0 overlaps there
        if div > 0:
            cols = np.flatnonzero(mask[0, 5])
            assert len(cols) >= 6 and cols.min() > w // 2 - 8 and cols.max() < w // 2 + 8
            assert (got[0, 0, 5, cols] == 1.0).all() and (got[0, 1, 5, cols] == 0.0).all()   # smeared from the LEFT (red)
        else:
            assert mask[0, 5, 8:-8].sum() == 0
            seam = got[0, :, 5, w // 2 - 3: w // 2 + 3]
            assert (seam[1] == 1.0).all() and (seam[0] == 0.0).all()   # the near (green) surface wins the overlap


def test_node_path_with_the_mesh_switch():
    """cs_params.flags bit 2 through the whole node path (engine.MESH_WARP): SBS and anaglyph, blur on and off."""
    from comfystereo_amd import engine
    n, h, w = 5, 48, 80
    img = synth.image_f32(n, h, w, seed=5)
    depth = synth.depth_batch("stepped", n, h, w, channels=3)
    engine.MESH_WARP = True
    try:
        for mode, blur in (("left-right", False), ("red-cyan-anaglyph", True), ("top-bottom", False)):
            p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", mode, 5.0, 0.5, 0.1, 0.5, 1.0, blur, 6.0, 6.0, 1.0, 2, 2)
            assert p.flags & 4
            got = [t.cpu().numpy() for t in engine.Plan(p, torch.device("cuda")).run(torch.from_numpy(img).cuda(), torch.from_numpy(depth).cuda())]
            want = node_oracle.generate(img, depth, 5.0, 0.5, mode, 0.1, 0.5, 1.0, "GPU Warp (Fast)", 6.0, 6.0, blur,
                                        depth_blur_falloff=1.0, depth_blur_vert_smooth=2, batch_size=2, mesh=True)
            assert (got[3] != want[3]).mean() <= 1e-3
            assert (np.abs(got[0] - want[0]) > TOL).mean() <= 3e-3
            np.testing.assert_allclose(got[1], want[1], atol=1e-6)
            np.testing.assert_allclose(got[2], want[2], atol=1e-6)
    finally:
        engine.MESH_WARP = False
