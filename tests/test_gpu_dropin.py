"""The drop-in surface itself on the GPU: StereoImageNode.generate with CPU tensors (as ComfyUI calls it) against the
reference's captured outputs, and the two module functions GenerateStereo.py calls.  -m gpu."""
import numpy as np
import pytest
import torch
from PIL import Image

import synth
from conftest import node_case_expected, node_case_inputs
from oracle import node_oracle

pytestmark = pytest.mark.gpu


def test_node_generate_matches_reference_goldens(golden_node):
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUES, StereoImageNode
    ui = {v: k for k, v in FILL_TECHNIQUES.items()}
    node = StereoImageNode()
    n = 0
    for case in golden_node.meta["cases"]:
        img, depth = node_case_inputs(golden_node, case)
        out = node.generate(torch.from_numpy(img), torch.from_numpy(depth), case["divergence"], case["separation"], case["mode"],
                            case["balance"], case["convergence"], case["exponent"], ui[case["fill"]], case["edge_threshold"],
                            case["strength"], case["blur"], **case["kw"])
        assert all(isinstance(t, torch.Tensor) and not t.is_cuda and t.dtype == torch.float32 for t in out)
        want = node_case_expected(golden_node, case)
        got = [t.numpy() for t in out]
        if case["fill"] == "gpu_warp":
            assert np.abs(got[0] - want[0]).max() <= 1e-4, case["id"]
        else:
            assert np.array_equal(got[0], want[0]), case["id"]
        assert np.array_equal(got[1][..., 0], want[1]) and np.array_equal(got[2][..., 0], want[2]), case["id"]
        assert np.array_equal(got[3], want[3]), case["id"]
        n += 1
    assert n >= 45


def test_unknown_fill_string_falls_back_to_gpu_warp():
    from comfystereo_amd.GenerateStereo import StereoImageNode
    img = torch.from_numpy(synth.image_f32(1, 32, 64, seed=1))
    dep = torch.from_numpy(synth.depth_batch("radial", 1, 32, 64, channels=3))
    a = StereoImageNode().generate(img, dep, 4.5, 0, "left-right", 0, 0.5, 2, "no such technique", 20, 20, False)
    b = StereoImageNode().generate(img, dep, 4.5, 0, "left-right", 0, 0.5, 2, "GPU Warp (Fast)", 20, 20, False)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and a[3].shape == (1, 32, 64)


@pytest.mark.parametrize("fill", ["polylines_soft", "naive_interpolating", "hybrid_edge", "no-such-technique", "none_post",
                                  "inverse_post", "hybrid_edge_plus"])
def test_create_stereoimages_returns_pil_like_the_reference(fill):
    from comfystereo_amd import stereoimage_generation as sig
    h, w = 40, 96
    img = synth.image_f32(1, h, w, seed=2)[0]
    depth = synth.blobs(h, w, seed=1)
    args = dict(divergence=6.0, separation=0.5, modes=["left-right", "red-cyan-anaglyph"], stereo_balance=0.2,
                stereo_offset_exponent=1.3, fill_technique=fill, depth_blur_strength=8.0, depth_blur_edge_threshold=6.0,
                direction_aware_depth_blur=False, convergence_point=0.4)
    out = sig.create_stereoimages(torch.from_numpy(img).permute(2, 0, 1), torch.from_numpy(depth), **args)
    assert isinstance(out, tuple) and len(out) == 2  # (images, modified depth) when the blur is off
    images, mod = out
    assert all(isinstance(i, Image.Image) for i in images) and isinstance(mod, Image.Image)
    assert images[0].size == (2 * w, h) and images[1].size == (w, h) and mod.size == (w, h) and mod.mode == "L"
    okw = {k: v for k, v in args.items() if k not in ("divergence", "separation", "modes", "stereo_balance", "direction_aware_depth_blur")}
    if fill == "no-such-technique":  # reference falls through its dispatch: both eyes are the source image
        src = np.clip(img * np.float32(255), 0, 255).astype(np.uint8)
        assert np.array_equal(np.array(images[0]), np.hstack([src, src]))
        return
    want, ml, _ = node_oracle.create_stereoimages(img.transpose(2, 0, 1), depth, 6.0, 0.5, ["left-right", "red-cyan-anaglyph"],
                                                  0.2, direction_aware_depth_blur=False, **okw)
    assert np.array_equal(np.array(images[0]), want[0]) and np.array_equal(np.array(images[1]), want[1])
    assert np.array_equal(np.array(mod), ml)
    # blur on: three return values, and return_modified_depth=False: the list only
    out3 = sig.create_stereoimages(torch.from_numpy(img).permute(2, 0, 1), torch.from_numpy(depth), 6.0, 0.5, "left-right", 0.2, 1.3,
                                   fill, 8.0, 6.0, True)
    assert len(out3) == 3
    assert isinstance(sig.create_stereoimages(torch.from_numpy(img).permute(2, 0, 1), torch.from_numpy(depth), 6.0,
                                              return_modified_depth=False, fill_technique=fill), list)


def test_create_stereoimages_gpu_matches_oracle():
    from comfystereo_amd import stereoimage_generation as sig
    b, h, w = 3, 64, 200
    img = synth.image_f32(b, h, w, seed=3).transpose(0, 3, 1, 2).copy()
    depth = synth.depth_batch("blobs", b, h, w, channels=1)[..., 0] * np.float32(1.7)  # > 1: already "0..255-like"
    res, lo, ro, mask = sig.create_stereoimages_gpu(torch.from_numpy(img), torch.from_numpy(depth), 4.5, 0.2,
                                                    ["left-right", "top-bottom"], 0.1, 2.0, 0.5, 20.0, 20.0, False)
    want, wl, wr, wm = node_oracle.create_stereoimages_gpu(img, depth, 4.5, 0.2, ["left-right", "top-bottom"], 0.1, 2.0, 0.5, 20.0,
                                                           20.0, False)
    assert len(res) == 2 and res[0].is_cuda and tuple(res[0].shape) == (b, 3, h, 2 * w) and tuple(res[1].shape) == (b, 3, 2 * h, w)
    for r, wv in zip(res, want):
        assert np.abs(r.cpu().numpy() - wv).max() <= 2e-6
    assert mask.dtype == torch.bool and np.array_equal(mask.cpu().numpy(), wm)
    assert np.array_equal(lo.cpu().numpy(), wl) and np.array_equal(ro.cpu().numpy(), wr)  # unclamped, like the reference


@pytest.mark.parametrize("fill,n,batch", [("polylines_soft", 7, 12), ("gpu_warp", 10, 4), ("hybrid_edge", 3, 2), ("none", 1, 1)])
def test_host_pipeline_equals_the_whole_batch(monkeypatch, fill, n, batch):
    """host_pipeline.generate_host (CPU tensors in/out, chunks staged through pinned memory and overlapped) ==
    engine.generate on the whole batch at once: chunk boundaries, a shorter last chunk, buffer reuse two chunks later,
    gpu_warp chunks aligned to the reference's sub-batches."""
    from comfystereo_amd import engine, host_pipeline
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUES, StereoImageNode
    UI = {v: k for k, v in FILL_TECHNIQUES.items()}
    NODE = StereoImageNode()
    h, w = 40, 328
    img = torch.from_numpy(synth.image_f32(n, h, w, seed=21))
    dep = torch.from_numpy(synth.depth_batch("blobs", n, h, w, channels=3))
    args = (6.0, 0.2, "left-right", 0.1, 0.5, 2.0, fill, 20.0, 20.0, True, 2.0, 3, batch)
    ref = [t.cpu() for t in engine.generate(img.cuda(), dep.cuda(), *args)]
    per_frame_out = 4 * (h * 2 * w * 3 + 2 * h * w * 3 + h * 2 * w)
    monkeypatch.setattr(host_pipeline, "CHUNK_OUT_BYTES", 2 * per_frame_out + 1)  # two frames per chunk
    seen = []
    for pinned in (True, False):  # float32 boundary: results written straight into pinned tensors / staged into pageable ones
        seen.clear()
        got = host_pipeline.generate_host(img, dep, *args, progress=seen.append, pinned_outputs=pinned, compact=False)
        assert sum(seen) == n
        for g, r in zip(got, ref):
            assert not g.is_cuda and g.is_pinned() == pinned and torch.equal(g, r)
    if fill != "gpu_warp":
        # compact boundary (round 3; the default for the CPU techniques): uint8 codes over PCIe, float32 written by host threads
        monkeypatch.setattr(host_pipeline, "CHUNK_IN_BYTES", 2 * 4 * (h * w * 3 + h * w * 3) + 1)  # two frames per chunk
        for threads, pinned in ((0, True), (1, False), (3, True)):
            seen.clear()
            got = host_pipeline.generate_host(img, dep, *args, progress=seen.append, expand_threads=threads, pinned_outputs=pinned)
            assert sum(seen) == n
            for g, r in zip(got, ref):
                assert not g.is_cuda and g.is_pinned() == pinned and g.dtype == torch.float32 and torch.equal(g, r)
    else:
        with pytest.raises(ValueError):
            host_pipeline.generate_host(img, dep, *args, compact=True)
    # caller-provided result tensors: pinned ones are written directly, pageable ones through the staging
    shapes = host_pipeline.result_shapes(img.shape, "left-right", fill)
    for pin in (True, False):
        outs = tuple(torch.full(sh, -7.0).pin_memory() if pin else torch.full(sh, -7.0) for sh in shapes)
        got = host_pipeline.generate_host(img, dep, *args, out=outs)
        for g, o, r in zip(got, outs, ref):
            assert g.data_ptr() == o.data_ptr() and torch.equal(o, r)
    with pytest.raises(ValueError):
        host_pipeline.generate_host(img, dep, *args, out=(torch.empty(1),) * 4)
    node_out = NODE.generate(img, dep, 6.0, 0.2, "left-right", 0.1, 0.5, 2.0, UI[fill], 20.0, 20.0, True, 2.0, 3, batch)
    for g, r in zip(node_out, ref):
        assert torch.equal(g, r)


@pytest.mark.parametrize("ui,key", [("Fill - Post-fill", "none_post"),
                                    ("Fill - Reverse projection with Post-fill", "inverse_post"),
                                    ("Fill - Hybrid Edge with fill", "hybrid_edge_plus")])
def test_node_maps_the_strings_the_reference_keeps_out_of_its_combo_list(ui, key):
    """reference GenerateStereo.py:97-99 still translates three strings whose combo entries are commented out (:56-57):
    an API workflow passing them must get those techniques, not the gpu_warp fallback of unknown strings (:102)."""
    from comfystereo_amd.GenerateStereo import FILL_TECHNIQUES, FILL_TECHNIQUE_MAPPING, StereoImageNode
    from oracle import node_oracle
    assert ui not in FILL_TECHNIQUES and FILL_TECHNIQUE_MAPPING[ui] == key
    assert ui not in StereoImageNode.INPUT_TYPES()["required"]["fill_technique"][0]
    n, h, w = 2, 40, 200
    img = synth.image_f32(n, h, w, seed=5)
    dep = synth.depth_batch("blobs", n, h, w, channels=3)
    args = (6.0, 0.2, "left-right", 0.1, 0.5, 2.0)
    got = StereoImageNode().generate(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), *args, ui, 20.0, 20.0, False,
                                     2.0, 6, 12)
    want = node_oracle.generate(img, dep, *args, ui, 20.0, 20.0, False, depth_blur_falloff=2.0, depth_blur_vert_smooth=6,
                                batch_size=12)
    for g, w_, name in zip(got, want, ("stereoscope", "depth_left", "depth_right", "mask")):
        assert np.array_equal(g.cpu().numpy(), w_), name
    unknown = StereoImageNode().generate(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), *args, "no such fill",
                                         20.0, 20.0, False, 2.0, 6, 12)
    assert unknown[3].shape == (n, h, w)  # gpu_warp: eye-shaped mask


def test_blur_with_zero_strength_is_blur_off():
    """reference :1194 / :1050: direction-aware blur with strength <= 0 returns the depth unchanged."""
    from comfystereo_amd import engine
    n, h, w = 1, 32, 160
    img = torch.from_numpy(synth.image_f32(n, h, w, seed=8)).cuda()
    dep = torch.from_numpy(synth.depth_batch("blobs", n, h, w, channels=3)).cuda()
    for fill in ("polylines_soft", "gpu_warp"):
        a = engine.generate(img, dep, 5.0, 0.0, "left-right", 0.0, 0.5, 2.0, fill, 20.0, 0.0, True, 2.0, 6, 4)
        b = engine.generate(img, dep, 5.0, 0.0, "left-right", 0.0, 0.5, 2.0, fill, 20.0, 20.0, False, 2.0, 6, 4)
        for x, y in zip(a, b):
            assert torch.equal(x, y), fill


def test_blur_mask_width_is_independent_of_the_strength():
    """directional_motion_blur_gpu(depth, strength, threshold, blur_mask_width, ...): the weights reach
    int(blur_mask_width) pixels from an edge whatever the box width is (reference :1208-1209)."""
    from comfystereo_amd import stereoimage_generation as sig
    from oracle import oracle
    depth = np.round(synth.blobs(270, 480, seed=1) * 255).astype(np.float32)
    for s, t, mw, f, v in [(20, 20, 5, 2.0, 3), (20, 20, 33.7, 1.0, 0), (7.4, 6, 12, 2.0, 6)]:
        L, R = sig.directional_motion_blur_gpu(torch.from_numpy(depth), s, t, mw, falloff_exponent=f, vert_smooth_px=v)
        oL, oR = oracle.blur(depth, s, t, f, v, mask_width=mw)
        assert np.array_equal(L.cpu().numpy().view(np.uint32), oL.view(np.uint32))
        assert np.array_equal(R.cpu().numpy().view(np.uint32), oR.view(np.uint32))


def test_stereo_shift_torch_matches_reference_vectors_and_oracle():
    """comfystereo_amd.stereo_utils.stereo_shift_torch (cs_stereo_shift) == the reference's captured outputs, bit for bit;
    larger random cases against the oracle."""
    from conftest import Golden
    from comfystereo_amd.stereo_utils import stereo_shift_torch
    from oracle import stereo_shift_oracle as so
    g = Golden("stereo_shift.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        x, d = torch.from_numpy(g[f"{cid}/x"]), torch.from_numpy(g[f"{cid}/d"])
        got = stereo_shift_torch(x, d, case["scale_factor"], case["shift_both"], case["exponent"])
        assert not got.is_cuda and got.dtype == torch.float32   # CPU tensors in -> CPU tensor out, like the reference
        assert np.array_equal(got.numpy(), g[f"{cid}/out"]), cid
    rng = np.random.default_rng(3)
    for (b, c, h, w, sf, both, e) in [(2, 4, 96, 160, 10.0, True, 1.0), (1, 8, 33, 257, -14.0, False, 2.0)]:
        x = rng.standard_normal((b, c, h, w)).astype(np.float32)
        d = rng.random((b, h, w)).astype(np.float32)
        got = stereo_shift_torch(torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda(), sf, both, e)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), so.stereo_shift(x, d, sf, both, e))


def test_create_stereoimages_numpy_and_pil_inputs():
    """create_stereoimages with numpy / PIL inputs, depth blur off (reference stereoimage_generation.py:1486-1499, 1519-1574):
    fixtures captured from the reference (tools/make_goldens.py --only-numpy-inputs), every technique, all return forms.
    With the blur on, that input form takes the reference's scipy blur: test_create_stereoimages_numpy_inputs_with_the_blur_on."""
    from PIL import Image
    from conftest import Golden
    from comfystereo_amd import stereoimage_generation as sig
    g = Golden("create_stereoimages_numpy.npz")
    for case in g.meta["cases"]:
        img, depth = g[f"{case['group']}/img"], g[f"{case['group']}/depth"]
        res, mod = sig.create_stereoimages(Image.fromarray(img) if case["pil"] else img, depth.tolist() if case["pil"] else depth,
                                           case["divergence"], case["separation"], case["modes"], case["stereo_balance"],
                                           case["stereo_offset_exponent"], case["fill_technique"], 0.0, 6.0, False, True,
                                           case["convergence_point"])
        assert len(res) == len(case["modes"])
        for k, r in enumerate(res):
            assert isinstance(r, Image.Image) and np.array_equal(np.asarray(r), g[f"{case['id']}/out{k}"]), (case["id"], k)
        assert np.array_equal(np.asarray(mod), g[f"{case['id']}/mod"]), case["id"]
    img, depth = g["c0/img"], g["c0/depth"]
    only = sig.create_stereoimages(img, depth, 5.0, 0.0, "left-right", 0.0, 2.0, "polylines_soft", 0.0, 6.0, False, False)
    assert isinstance(only, list) and len(only) == 1
    three = sig.create_stereoimages(img, depth, 5.0, 0.0, ["left-right"], 0.0, 2.0, "none", 0.0, 6.0, True, True)
    assert len(three) == 3   # (direction-aware flag with strength 0: the reference returns the depth twice, :1378)
    with pytest.raises(Exception):
        sig.create_stereoimages(img, depth, 5.0, modes=["sideways"])


def test_scipy_depth_blur_kernels_match_the_reference():
    """cs_directional_blur_scipy (cs_scipyblur.hip) = `directional_motion_blur`, reference stereoimage_generation.py:1346-1419:
    outputs of the reference (tools/make_goldens.py --only-numpy-blur) bit for bit where NumPy's array power is exact, and the
    oracle restatement on larger frames (several workgroups per row, the even / odd box loops, every border)."""
    from conftest import Golden
    from oracle import scipy_blur_oracle as sb
    from comfystereo_amd import engine
    g = Golden("numpy_blur.npz")
    for case in g.meta["cases"]:
        cid = case["id"]
        depth = g[f"{cid}/depth"]
        left, right = engine.directional_blur_scipy(torch.from_numpy(depth).cuda(), case["strength"], case["edge_threshold"], case["strength"],
                                                    case["falloff"], case["vert"])
        for got, want in ((left.cpu().numpy(), g[f"{cid}/left"]), (right.cpu().numpy(), g[f"{cid}/right"])):
            if case["exact_power"]:
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), cid
            else:
                assert np.abs(got - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), cid
    for (h, w, kind, strength, thr, falloff, vert) in [(70, 1300, "blobs", 20.0, 20.0, 2.0, 6), (33, 777, "noisy_ramp", 21.0, 0.5, 0.5, 15),
                                                       (50, 600, "clipped", 64.0, 8.0, 1.0, 0), (12, 300, "stepped", 3.0, 2.0, 1.3, 2)]:
        depth = np.round(synth.DEPTHS[kind](h, w, **({} if kind == "stepped" else {"seed": 3})) * 255).astype(np.float32)
        left, right = engine.directional_blur_scipy(torch.from_numpy(np.stack([depth, depth[::-1].copy()])).cuda(), strength, thr, strength, falloff, vert)
        for f, d in enumerate((depth, depth[::-1].copy())):
            wl, wr = sb.directional_motion_blur(d, strength, thr, strength, falloff_exponent=falloff, vert_smooth_px=vert)
            assert np.array_equal(left[f].cpu().numpy(), wl) and np.array_equal(right[f].cpu().numpy(), wr), (kind, f)
    same_l, same_r = engine.directional_blur_scipy(torch.from_numpy(depth).cuda(), 0.0, 2.0)
    assert same_l is same_r   # (strength <= 0: the depth map itself, :1374)
    with pytest.raises(RuntimeError):   # round(0.4) = 0 taps: scipy raises in the reference
        engine.directional_blur_scipy(torch.from_numpy(depth).cuda(), 0.4, 2.0, 0.4)


def test_create_stereoimages_numpy_inputs_with_the_blur_on():
    """numpy / PIL inputs with direction_aware_depth_blur=True (reference :1486-1499): fixtures of the reference, all return forms."""
    from PIL import Image
    from conftest import Golden
    from comfystereo_amd import stereoimage_generation as sig
    g = Golden("numpy_blur.npz")
    for case in g.meta["create_stereoimages"]:
        cid = case["id"]
        img, depth = g[f"{cid}/img"], g[f"{cid}/depth"]
        res, ml, mr = sig.create_stereoimages(Image.fromarray(img) if case["pil"] else img, depth.tolist() if case["pil"] else depth,
                                              case["divergence"], case["separation"], case["modes"], case["stereo_balance"],
                                              case["stereo_offset_exponent"], case["fill_technique"], case["depth_blur_strength"],
                                              case["depth_blur_edge_threshold"], True, True, case["convergence_point"],
                                              case["depth_blur_falloff"], case["depth_blur_vert_smooth"])
        assert len(res) == len(case["modes"])
        for k, r in enumerate(res):
            assert isinstance(r, Image.Image) and np.array_equal(np.asarray(r), g[f"{cid}/out{k}"]), (cid, k)
        assert np.array_equal(np.asarray(ml), g[f"{cid}/mod_left"]) and np.array_equal(np.asarray(mr), g[f"{cid}/mod_right"]), cid
