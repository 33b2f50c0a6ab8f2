"""oracle_forward_warp_mesh (oracle/stereo_oracle.c): the specification of the mesh-quality gpu_warp (reference
stereoimage_generation.py:453-689).  PARITY UNPINNED -- the reference rasterises through moderngl / OpenGL, neither of
which exists in this image, so no fixture of the reference can be generated; these tests pin the properties the
reference's construction (mesh topology :507-520, culling :523-535, clip-space mapping :546-551, '<' depth test :601,
directional smear :664-687) guarantees for any conforming rasteriser.  Runs without a GPU."""
import numpy as np

import synth
from oracle import oracle


def _bary(img, r, t, x, f):
    c00, c10, c01, c11 = img[:, r, x], img[:, r, x + 1], img[:, r + 1, x], img[:, r + 1, x + 1]
    left, diag, right = (1 - t) * c00 + t * c01, (1 - t) * c10 + t * c01, (1 - t) * c10 + t * c11
    if f < 1 - t:
        s = f / (1 - t)
        return (1 - s) * left + s * diag
    s = (f - (1 - t)) / t
    return (1 - s) * diag + s * right


def test_zero_offset_resamples_the_mesh_at_the_pixel_centres():
    """Offsets 0 (depth == convergence everywhere is impossible after normalisation, so use separation 0 and a flat map at
    convergence 0): the output is the source mesh sampled at ((px + .5)(W-1)/W, (k + .5)(H-1)/H), no gaps."""
    h, w = 24, 40
    img = np.random.default_rng(1).random((1, 3, h, w), dtype=np.float32)
    dep = np.full((1, h, w), 0.7, np.float32)   # flat -> normalised 0; convergence 0 -> offset 0
    out, mask = oracle.forward_warp_mesh(img, dep, 7.0, 0.0, 1.0, 0.0)
    assert not mask.any()
    for k in (0, 11, h - 1):
        wy = (k + 0.5) * (h - 1) / h; r = int(np.floor(wy)); t = wy - r
        for px in (0, 13, w - 1):
            u = (px + 0.5) * (w - 1) / w; x = int(np.floor(u))
            np.testing.assert_allclose(out[0, :, k, px], _bary(img[0], r, t, x, u - x), atol=3e-5)


def test_culling_threshold_and_any_frame_rule():
    h, w = 12, 64
    img = np.random.default_rng(2).random((2, 3, h, w), dtype=np.float32)
    step = np.zeros((h, w), np.float32); step[:, w // 2:] = 1.0
    ramp = np.tile(np.linspace(0, 1, w, dtype=np.float32), (h, 1))
    # frame 0 alone: the step tears (offset jump 8 px >= 1.5) and leaves a gap for the left eye
    _, m0 = oracle.forward_warp_mesh(img[:1], step[None], 8.0, 0.0, 1.0, 0.5)
    assert m0[0, 3].sum() >= 6
    # a huge threshold keeps every triangle: the stretched triangles cover the seam
    _, m1 = oracle.forward_warp_mesh(img[:1], step[None], 8.0, 0.0, 1.0, 0.5, 100.0)
    assert m1[0, 3, 4:-4].sum() == 0
    # in one tensor with a smooth frame the seam triangles pass in THAT frame ("ANY batch item", :533-535) and are drawn in both
    _, m2 = oracle.forward_warp_mesh(img, np.stack([step, ramp]), 8.0, 0.0, 1.0, 0.5)
    assert m2[0, 3, 4:-4].sum() == 0


def test_depth_test_and_smear_direction():
    h, w = 10, 96
    img = np.zeros((1, 3, h, w), np.float32)
    img[:, 0, :, : w // 2] = 1.0
    img[:, 1, :, w // 2:] = 1.0
    dep = np.zeros((1, h, w), np.float32); dep[..., w // 2:] = 1.0
    out, mask = oracle.forward_warp_mesh(img, dep, 8.0, 0.0, 1.0, 0.5)      # left eye: gap at the seam, filled from the left
    cols = np.flatnonzero(mask[0, 4])
    assert len(cols) >= 6 and (out[0, 0, 4, cols] == 1.0).all() and (out[0, 1, 4, cols] == 0.0).all()
    out, mask = oracle.forward_warp_mesh(img, dep, -8.0, 0.0, 1.0, 0.5)     # right eye: overlap, the near surface wins
    assert mask[0, 4, 8:-8].sum() == 0
    seam = out[0, :, 4, w // 2 - 3: w // 2 + 3]
    assert (seam[1] == 1.0).all() and (seam[0] == 0.0).all()
    # right-eye gaps (at the right border here) smear from the right: nothing to take -> they stay black
    assert mask[0, 4, -3:].all() and (out[0, :, 4, -3:] == 0.0).all()


def test_scale_invariance_of_the_depth_map():
    """0..1 and 0..255 depth maps give the same mesh (:486-488 divides by 255 when any value exceeds 1)."""
    img, dep = np.random.default_rng(3).random((2, 3, 20, 48), dtype=np.float32), synth.depth_batch("stepped", 2, 20, 48, channels=1)[..., 0]
    codes = np.round(dep * 255.0).astype(np.float32)
    a, ma = oracle.forward_warp_mesh(img, codes, 5.0, 0.5, 1.0, 0.5)
    b, mb = oracle.forward_warp_mesh(img, codes / np.float32(255.0), 5.0, 0.5, 1.0, 0.5)
    assert (ma == mb).all()
    np.testing.assert_allclose(a, b, atol=1e-6)
