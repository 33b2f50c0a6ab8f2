"""The drop-in surface: node attributes, module function signatures and error behaviour mirror the reference
(GenerateStereo.py:46-80,460-466; stereoimage_generation.py:1005-1009,1422-1426).  Not gpu."""
import inspect

import pytest
import torch

from comfystereo_amd import GenerateStereo as gs
from comfystereo_amd import stereoimage_generation as sig


def test_node_protocol():
    n = gs.StereoImageNode
    assert n.RETURN_TYPES == ("IMAGE", "IMAGE", "IMAGE", "MASK")
    assert n.RETURN_NAMES == ("stereoscope", "blurred_depthmap_left", "blurred_depthmap_right", "no_fill_imperfect_mask")
    assert n.FUNCTION == "generate" and not hasattr(n, "CATEGORY")
    assert gs.NODE_CLASS_MAPPINGS == {"StereoImageNode": n}
    assert gs.NODE_DISPLAY_NAME_MAPPINGS == {"StereoImageNode": "Stereo Image Node"}
    it = n.INPUT_TYPES()
    assert list(it["required"]) == ["image", "depth_map", "modes", "fill_technique"]
    assert it["required"]["modes"][0] == ["left-right", "right-left", "top-bottom", "bottom-top", "red-cyan-anaglyph"]
    fills, opts = it["required"]["fill_technique"]
    assert fills == ['GPU Warp (Fast)', 'No fill', 'No fill - Reverse projection', 'Imperfect fill - Hybrid Edge',
                     'Fill - Naive', 'Fill - Naive interpolating', 'Fill - Polylines Soft', 'Fill - Polylines Sharp']
    assert opts["default"] == "GPU Warp (Fast)"
    want = {"divergence": (4.5, 0.05, 15, 0.01), "separation": (0, -5, 5, 0.01), "stereo_balance": (0, -0.95, 0.95, 0.05),
            "convergence_point": (0.5, 0.0, 1.0, 0.05), "stereo_offset_exponent": (2, 0.1, 2, 0.1),
            "depth_blur_edge_threshold": (20, 0.1, 60, 0.1), "depth_blur_strength": (20, 0.1, 200, 0.1),
            "depth_blur_falloff": (2.0, 0.1, 4.0, 0.1), "depth_blur_vert_smooth": (6, 0, 15, 1), "batch_size": (12, 1, 64, 1)}
    for k, (d, lo, hi, st) in want.items():
        o = it["optional"][k][1]
        assert (o["default"], o["min"], o["max"], o["step"]) == (d, lo, hi, st), k
    assert it["optional"]["depth_map_blur"] == ("BOOLEAN", {"default": True, "tooltip": it["optional"]["depth_map_blur"][1]["tooltip"]})


def test_generate_signature_matches_reference():
    params = list(inspect.signature(gs.StereoImageNode.generate).parameters.values())
    names = [p.name for p in params]
    assert names == ["self", "image", "depth_map", "divergence", "separation", "modes", "stereo_balance", "convergence_point",
                     "stereo_offset_exponent", "fill_technique", "depth_blur_edge_threshold", "depth_blur_strength",
                     "depth_map_blur", "depth_blur_falloff", "depth_blur_vert_smooth", "batch_size"]
    d = {p.name: p.default for p in params if p.default is not inspect._empty}
    assert d == {"depth_blur_falloff": 1.0, "depth_blur_vert_smooth": 0, "batch_size": 4}  # quirk Q9


def test_module_function_signatures():
    s = inspect.signature(sig.create_stereoimages)
    assert list(s.parameters) == ["original_image", "depthmap", "divergence", "separation", "modes", "stereo_balance",
                                  "stereo_offset_exponent", "fill_technique", "depth_blur_strength",
                                  "depth_blur_edge_threshold", "direction_aware_depth_blur", "return_modified_depth",
                                  "convergence_point", "depth_blur_falloff", "depth_blur_vert_smooth"]
    assert s.parameters["fill_technique"].default == "polylines_sharp" and s.parameters["stereo_offset_exponent"].default == 1.0
    g = inspect.signature(sig.create_stereoimages_gpu)
    assert list(g.parameters) == ["image_tensor", "depth_tensor", "divergence", "separation", "modes", "stereo_balance",
                                  "stereo_offset_exponent", "convergence_point", "depth_blur_strength",
                                  "depth_blur_edge_threshold", "direction_aware_depth_blur", "depth_blur_falloff",
                                  "depth_blur_vert_smooth"]


def test_empty_modes_and_unknown_mode():
    img, dep = torch.zeros(3, 4, 4), torch.zeros(4, 4)
    assert sig.create_stereoimages(img, dep, 5.0, modes=[]) == []
    assert sig.create_stereoimages_gpu(img[None], dep[None], 5.0, modes=[]) == ([], None, None, None)
    with pytest.raises(Exception, match="Unknown mode"):
        sig.create_stereoimages(img, dep, 5.0, modes=["sideways"])
    with pytest.raises(ValueError, match="Unknown mode: sideways"):
        sig.create_stereoimages_gpu(img[None], dep[None], 5.0, modes=["sideways"])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_a_gpu():
    img, dep = torch.zeros(1, 4, 4, 3), torch.zeros(1, 4, 4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gs.StereoImageNode().generate(img, dep, 4.5, 0, "left-right", 0, 0.5, 2, "Fill - Polylines Soft", 20, 20, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sig.create_stereoimages(img[0].permute(2, 0, 1), dep[0, :, :, 0], 5.0)
    from comfystereo_amd import host_pipeline
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        host_pipeline.generate_host(img, dep, 4.5, 0, "left-right", 0, 0.5, 2, "polylines_soft", 20, 20, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # the techniques no UI string reaches included
        sig.apply_stereo_divergence(torch.zeros(4, 4, 3, dtype=torch.uint8), dep[0, :, :, 0], 5.0, 0.0, 2.0, "none_post")


def test_import_allocates_nothing_and_starts_no_thread():
    """ADVICE r4 (medium) / VERDICT r4 item 8: the node module's warm-up is OPT-IN (GenerateStereo.PREWARM / COMFYSTEREO_PREWARM).
    With PREWARM None -- the default -- importing the module inside ComfyUI starts no thread, initialises no GPU and page-locks
    nothing, like the reference's module."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, types, threading\n"
        "m = types.ModuleType('comfy'); u = types.ModuleType('comfy.utils')\n"
        "class ProgressBar:\n"
        "    def __init__(self, total): pass\n"
        "    def update(self, k): pass\n"
        "u.ProgressBar = ProgressBar; m.utils = u; sys.modules['comfy'] = m; sys.modules['comfy.utils'] = u\n"
        "import torch\n"
        "from comfystereo_amd import GenerateStereo as gs, host_pipeline as hp\n"
        "assert gs._IN_COMFYUI is True and gs.PREWARM is None\n"
        "assert hp._prewarm_thread is None\n"
        "assert not [t.name for t in threading.enumerate() if t.name.startswith('comfystereo')]\n"
        "assert not torch.cuda.is_initialized()\n"
        "st = torch.cuda.host_memory_stats() if hasattr(torch.cuda, 'host_memory_stats') else {}\n"
        "assert st.get('allocated_bytes.current', 0) == 0 and st.get('reserved_bytes.current', 0) == 0, st\n"
        "print('inert')\n")
    env = {k: v for k, v in os.environ.items() if k != "COMFYSTEREO_PREWARM"}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "inert" in out.stdout, out.stderr[-2000:]


def test_prewarm_shape_is_opt_in_by_attribute_or_environment(monkeypatch):
    assert gs._prewarm_shape() is None
    monkeypatch.setenv("COMFYSTEREO_PREWARM", "32x2160x3840")
    assert gs._prewarm_shape() == (32, 2160, 3840)
    monkeypatch.setenv("COMFYSTEREO_PREWARM", "nonsense")
    assert gs._prewarm_shape() is None
    monkeypatch.setattr(gs, "PREWARM", (8, 1080, 1920))
    assert gs._prewarm_shape() == (8, 1080, 1920)


def test_pinned_pool_cap_release_with_hysteresis_and_staging_accounting(monkeypatch):
    """host_pipeline.PINNED_POOL_BYTES caps what the pipeline page-locks for itself.  PyTorch never returns cached pinned blocks, so
    the cache is released when the distinct needs served since the last release could no longer fit under the cap together -- and NOT
    on every change of shape (alternating between two shapes re-pinned gigabytes per call).  The float32 routes'
    staging copies (gpu_warp's colours when the results are pageable) are part of the accounting, and the chunk shrinks until
    the staging alone fits; generate_host and prewarm use the same helper."""
    from comfystereo_amd import host_pipeline as hp
    assert hp.PINNED_POOL_BYTES <= 8 << 30
    calls = []
    monkeypatch.setattr(hp, "_release_pinned_cache", lambda: calls.append(1))
    monkeypatch.setattr(hp, "_pinned_needs", set())
    hp._pinned_budget(4 << 30); assert not calls
    hp._pinned_budget(4 << 30); assert not calls
    hp._pinned_budget(1 << 30); assert not calls          # 4 + 1 GB: both shapes' blocks may stay cached
    for _ in range(4):                                      # alternating between the two: nothing is released
        hp._pinned_budget(4 << 30); hp._pinned_budget(1 << 30)
    assert not calls
    hp._pinned_budget(5 << 30); assert calls == [1]        # 4 + 1 + 5 GB would exceed the cap: release, start over
    hp._pinned_budget(2 << 30); assert calls == [1]
    # gpu_warp, 24 x 4K: float32 colours.  Pageable results -> each slot pins a staging copy of its chunk's stereoscope
    h, w = 2160, 3840
    fin = 4 * (h * w * 3 + h * w * 3)
    small = hp._small_bytes_per_frame(hp.ROUTES["warp"], h, 2 * w, h, w, h, w)
    f32 = 4 * h * 2 * w * 3
    fout = 4 * (h * 2 * w * 3 + 2 * h * w * 3 + h * w)
    chunk0 = hp._chunk_frames(24, fout, "gpu_warp", 4)
    chunk, staging, pin = hp._plan_pinned(24, chunk0, 4, fin, small, f32, fout, True)
    assert not pin and staging <= hp.PINNED_POOL_BYTES and chunk % 4 == 0 and chunk <= chunk0
    nr = (24 + chunk - 1) // chunk
    assert staging == (chunk * min(2, nr) + (24 % chunk if 24 % chunk and nr > 1 else 0)) * (fin + small + f32)
    # a small batch fits with its results: pinned results, no staging copies counted
    chunk, staging, pin = hp._plan_pinned(2, 2, 1, fin, small, f32, fout, True)
    assert pin and staging == 2 * (fin + small)
    # caller-provided pinned results (None): never "pin the results", no staging copies
    assert hp._plan_pinned(2, 2, 1, fin, small, f32, fout, None) == (2, 2 * (fin + small), False)
    # compact output forms per 4K frame: stereoscope codes 2 x 3 bytes per pixel, two depth codes, the SBS mask's two
    assert hp._small_bytes_per_frame(hp.ROUTES["compact"], 2160, 7680, 2160, 3840, 2160, 7680) == 2160 * 3840 * (6 + 1 + 1 + 2)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_warm_up_is_inert_without_a_gpu_and_outside_comfyui():
    """host_pipeline.prewarm (opt-in): nothing is started outside ComfyUI, and without a GPU the warm-up
    itself declines instead of raising."""
    from comfystereo_amd import host_pipeline
    assert gs.PREWARM is None and gs._IN_COMFYUI is False
    assert host_pipeline._prewarm_thread is None
    assert host_pipeline.prewarm(2, 64, 64) is False
    t = host_pipeline.prewarm_async(2, 64, 64)
    host_pipeline.prewarm_wait()
    assert not t.is_alive()
