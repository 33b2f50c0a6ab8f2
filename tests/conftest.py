import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """An .npz fixture: `meta` (json) + flat arrays keyed 'case/field'."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.meta = json.loads(str(self.z["meta"]))

    def __getitem__(self, key):
        return self.z[key]

    def has(self, key):
        return key in self.z.files


@pytest.fixture(scope="session")
def golden_asd():
    return Golden("apply_stereo_divergence.npz")


@pytest.fixture(scope="session")
def golden_hidden():
    """The UI-unreachable techniques (none_post, inverse_post, hybrid_edge_plus) on the inputs of golden_asd."""
    return Golden("apply_stereo_divergence_hidden.npz")


@pytest.fixture(scope="session")
def golden_blur():
    return Golden("blur.npz")


@pytest.fixture(scope="session")
def golden_warp():
    return Golden("forward_warp_gpu.npz")


@pytest.fixture(scope="session")
def golden_warp_params():
    """forward_warp_gpu with gradient_threshold / max_stretch away from their defaults (round 4)."""
    return Golden("forward_warp_params.npz")


@pytest.fixture(scope="session")
def golden_node():
    return Golden("node_generate.npz")


@pytest.fixture(scope="session")
def golden_node_extra():
    """API-only modes, the fill strings outside the combo list, depth maps of another size (round 2 fixture)."""
    return Golden("node_extra.npz")


@pytest.fixture(scope="session")
def golden_scene8():
    """The reference's node on quantised depth with object silhouettes (tools/synth.scene8)."""
    return Golden("scene8.npz")


def scene8_case_inputs(g, case):
    grp = case["group"]
    depth = np.repeat((g[f"{grp}/depth_u8"].astype(np.float32) / np.float32(255.0))[..., None], 3, -1)
    return g["img_u8"].astype(np.float32) / np.float32(255.0), depth


def scene8_case_expected(g, case):
    """(stereo, depth_left [N,H,W], depth_right, mask) of a scene8.npz case; gpu_warp: the stereoscope's rows [::row_step] only."""
    cid = case["id"]
    if g.has(f"{cid}/stereo_rows"):
        stereo, dl, dr = g[f"{cid}/stereo_rows"], g[f"{cid}/dl"], g[f"{cid}/dr"]
    else:
        stereo, dl, dr = (g[f"{cid}/{k}_u8"].astype(np.float32) / np.float32(255.0) for k in ("stereo", "dl", "dr"))
    mshape = case["shapes"]["mask"]
    mask = np.unpackbits(g[f"{cid}/mask"])[: int(np.prod(mshape))].reshape(mshape).astype(np.float32)
    return stereo, dl, dr, mask


def extra_case_inputs(g, case):
    grp = case["group"]
    return g[f"{grp}/img_u8"].astype(np.float32) / np.float32(255.0), g[f"{grp}/depth"]


def extra_case_expected(g, case):
    cid = case["id"]
    if case["gpu"]:
        stereo, dl, dr = g[f"{cid}/stereo"], g[f"{cid}/dl"], g[f"{cid}/dr"]
    else:
        stereo, dl, dr = (g[f"{cid}/{k}_u8"].astype(np.float32) / np.float32(255.0) for k in ("stereo", "dl", "dr"))
    mshape = case["shapes"]["mask"]
    mask = np.unpackbits(g[f"{cid}/mask"])[: int(np.prod(mshape))].reshape(mshape).astype(np.float32)
    return stereo, dl, dr, mask


def node_case_inputs(g, case):
    """(image NHWC f32, depth NHWC f32) of a node_generate.npz case."""
    grp = case["id"].split("/")[0]
    img = g[f"{grp}/img_u8"].astype(np.float32) / np.float32(255.0)
    if g.has(f"{grp}/depth"):
        depth = g[f"{grp}/depth"]
    else:
        d = g[f"{grp}/depth_u8"].astype(np.float32) / np.float32(255.0)
        depth = np.repeat(d[..., None], 3, -1)
    return img, depth


def node_case_expected(g, case):
    """(stereo f32, depth_left f32 [N,H,W], depth_right, mask f32) of a node_generate.npz case."""
    cid = case["id"]
    if g.has(f"{cid}/stereo"):
        stereo, dl, dr = g[f"{cid}/stereo"], g[f"{cid}/dl"], g[f"{cid}/dr"]
    else:
        stereo, dl, dr = (g[f"{cid}/{k}_u8"].astype(np.float32) / np.float32(255.0) for k in ("stereo", "dl", "dr"))
    mshape = case["shapes"]["mask"]
    mask = np.unpackbits(g[f"{cid}/mask"])[: int(np.prod(mshape))].reshape(mshape).astype(np.float32)
    return stereo, dl, dr, mask


@pytest.fixture
def dev_switch():
    """Set development switches of the library (cs_debug_set) for one test; all of them are reset afterwards."""
    from comfystereo_amd import _native
    touched = []

    def set_(key, value):
        touched.append(key)
        _native.debug_set(key, value)

    yield set_
    for key in touched:
        _native.debug_set(key, 0)


def assert_warp_colours(warped, want, gap_mask, cid, channel_axis=1):
    """gpu_warp colour parity (reference stereoimage_generation.py:394-448).  The coordinate round trip, torch.linspace and
    the bilinear blend are reproduced exactly, so outside the disocclusion gaps the colours agree to the last ulps (1e-6).
    Inside a gap the source position is an interpolation weighted by torch.sqrt, which CPU torch evaluates through MKL VML:
    not correctly rounded for ~0.4 % of the rational arguments that occur, and quirk Q2 (right border = the row's RIGHTMOST
    filled column) multiplies that ulp by spans of up to W pixels -- a handful of gap pixels per frame differ by up to 1e-4."""
    err = np.abs(np.asarray(warped, dtype=np.float64) - np.asarray(want, dtype=np.float64)).max(axis=channel_axis)
    gap = np.asarray(gap_mask, dtype=bool)
    assert err.shape == gap.shape, (cid, err.shape, gap.shape)
    if (~gap).any():
        assert err[~gap].max() <= 1e-6, (cid, "outside gaps", err[~gap].max())
    assert err.max() <= 1e-4, (cid, "inside gaps", err.max())
    assert (err > 1e-6).sum() <= max(8, 0.01 * gap.sum()), (cid, "pixels beyond the last ulps", int((err > 1e-6).sum()), int(gap.sum()))
