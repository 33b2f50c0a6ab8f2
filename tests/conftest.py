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
def golden_node():
    return Golden("node_generate.npz")


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
