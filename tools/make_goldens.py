#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the imported reference on seeded synthetic inputs.

Build-container only (needs /root/reference, which never travels to the GPU box).  The committed
fixtures are DATA: inputs and the reference's outputs, plus a manifest of the environment that
produced them.  Dialect: D32 (no numba, NumPy 2 promotion rules) -- the only one executable here.

  python tools/make_goldens.py            # regenerate everything (~3-4 min)
"""
import hashlib
import json
import os
import platform
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refload  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FILLS = ["none", "naive", "naive_interpolating", "polylines_soft", "polylines_sharp", "inverse", "hybrid_edge"]
UI_FILLS = {
    "gpu_warp": "GPU Warp (Fast)", "none": "No fill", "inverse": "No fill - Reverse projection",
    "hybrid_edge": "Imperfect fill - Hybrid Edge", "naive": "Fill - Naive",
    "naive_interpolating": "Fill - Naive interpolating", "polylines_soft": "Fill - Polylines Soft",
    "polylines_sharp": "Fill - Polylines Sharp",
}
MODES = ["left-right", "right-left", "top-bottom", "bottom-top", "red-cyan-anaglyph"]

# (divergence %, separation %, exponent, convergence): both signs, the hard exponents, sep != 0
ASD_PARAMS = [(5.0, 0.0, 2.0, 0.5), (-5.0, 0.0, 2.0, 0.5), (8.0, 1.3, 1.3, 0.7), (-12.0, -1.3, 0.5, 0.0),
              (3.0, 0.0, 1.0, 0.3), (-3.5, 0.4, 0.1, 1.0)]
ASD_KINDS = ["radial", "stepped", "noisy_ramp", "random8", "blobs"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_apply_stereo_divergence(sig):
    """Fixture for apply_stereo_divergence (reference :1576-1620 and every row kernel below it)."""
    H, W = 16, 96
    cases, arrays = [], {}
    for ki, kind in enumerate(ASD_KINDS):
        for pi, (div, sep, e, conv) in enumerate(ASD_PARAMS):
            if kind == "random8" and abs(div) > 6:
                div = 4.0 if div > 0 else -4.0  # keep the reference's csg scratch from overflowing
            img = synth.image_u8(H, W, seed=10 * ki + pi)
            depth = synth.DEPTHS[kind](H, W) * np.float32(255.0)
            cid = f"{kind}_{pi}"
            arrays[f"{cid}/img"] = img
            arrays[f"{cid}/depth"] = depth
            for fill in FILLS:
                arrays[f"{cid}/out/{fill}"] = sig.apply_stereo_divergence(img, depth, div, sep, e, fill, conv)
            cases.append(dict(id=cid, kind=kind, divergence=div, separation=sep, exponent=e, convergence=conv))
    # flat depth (max == min branch, :1591) and a zero-disparity case
    img = synth.image_u8(8, 40, seed=77)
    for cid, depth, div in (("flat", np.full((8, 40), 93.0, np.float32), 6.0),
                            ("tinydiv", synth.radial(8, 40) * np.float32(255), 0.01)):
        arrays[f"{cid}/img"] = img
        arrays[f"{cid}/depth"] = depth
        for fill in FILLS:
            arrays[f"{cid}/out/{fill}"] = sig.apply_stereo_divergence(img, depth, div, 0.0, 2.0, fill, 0.5)
        cases.append(dict(id=cid, kind=cid, divergence=div, separation=0.0, exponent=2.0, convergence=0.5))
    # larger frames: digests only (inputs come from tools/synth.py, also shipped)
    digests = []
    for (H2, W2, kind, div, sep, e, conv) in [(64, 512, "blobs", 6.0, 0.0, 2.0, 0.5), (64, 512, "stepped", -8.0, 0.5, 1.5, 0.4),
                                               (48, 640, "noisy_ramp", 3.5, 0.0, 0.7, 0.5)]:
        img = synth.image_u8(H2, W2, seed=5)
        depth = synth.DEPTHS[kind](H2, W2) * np.float32(255.0)
        d = dict(h=H2, w=W2, kind=kind, divergence=div, separation=sep, exponent=e, convergence=conv, img_seed=5,
                 img_sha=sha(img), depth_sha=sha(depth), out={})
        for fill in FILLS:
            d["out"][fill] = sha(sig.apply_stereo_divergence(img, depth, div, sep, e, fill, conv))
        digests.append(d)
    np.savez_compressed(os.path.join(OUT, "apply_stereo_divergence.npz"),
                        meta=json.dumps(dict(cases=cases, fills=FILLS, digests=digests)), **arrays)
    print("apply_stereo_divergence:", len(cases), "cases x", len(FILLS), "fills +", len(digests), "digest cases")


HIDDEN_FILLS = ["none_post", "inverse_post", "hybrid_edge_plus"]  # dispatcher branches no UI string reaches (:1605-1610)


def gen_hidden(sig):
    """The UI-unreachable techniques on the inputs of apply_stereo_divergence.npz (outputs only) + digests."""
    base = np.load(os.path.join(OUT, "apply_stereo_divergence.npz"))
    meta = json.loads(str(base["meta"]))
    arrays = {}
    for case in meta["cases"]:
        cid = case["id"]
        for fill in HIDDEN_FILLS:
            arrays[f"{cid}/out/{fill}"] = sig.apply_stereo_divergence(base[f"{cid}/img"], base[f"{cid}/depth"], case["divergence"],
                                                                      case["separation"], case["exponent"], fill, case["convergence"])
    digests = []
    for d in meta["digests"]:
        img = synth.image_u8(d["h"], d["w"], seed=d["img_seed"])
        depth = synth.DEPTHS[d["kind"]](d["h"], d["w"]) * np.float32(255.0)
        assert sha(img) == d["img_sha"] and sha(depth) == d["depth_sha"]
        e = dict(d, out={})
        for fill in HIDDEN_FILLS:
            e["out"][fill] = sha(sig.apply_stereo_divergence(img, depth, d["divergence"], d["separation"], d["exponent"], fill,
                                                             d["convergence"]))
        digests.append(e)
    np.savez_compressed(os.path.join(OUT, "apply_stereo_divergence_hidden.npz"),
                        meta=json.dumps(dict(cases=meta["cases"], fills=HIDDEN_FILLS, digests=digests,
                                             inputs="apply_stereo_divergence.npz")), **arrays)
    print("hidden techniques:", len(meta["cases"]), "cases x", len(HIDDEN_FILLS), "fills +", len(digests), "digest cases")


def gen_blur(sig):
    """Fixture for directional_motion_blur_gpu (reference :1171-1251) on CPU torch."""
    arrays, cases = {}, []
    specs = [(256, 256, 1, "blobs", 20, 20, 2.0, 6), (256, 256, 2, "stepped", 21, 6, 2.0, 3),
             (270, 480, 1, "blobs", 5.4, 0.5, 1.0, 15), (256, 256, 1, "noisy_ramp", 20, 20, 0.5, 0)]
    for i, (H, W, B, kind, st, thr, fo, v) in enumerate(specs):
        d8 = np.stack([np.round(synth.DEPTHS[kind](H, W, **({} if kind in ("radial", "stepped") else {"seed": j})) * 255)
                       for j in range(B)]).astype(np.uint8)
        if kind == "stepped":
            d8 = np.stack([np.round(synth.stepped(H, W, cx=W / 2 + 9 * j) * 255) for j in range(B)]).astype(np.uint8)
        depth = d8.astype(np.float32)
        t = torch.from_numpy(depth.copy())
        L, R = sig.directional_motion_blur_gpu(t if B > 1 else t[0], st, thr, st, falloff_exponent=fo, vert_smooth_px=v)
        arrays[f"c{i}/depth_u8"] = d8
        arrays[f"c{i}/L"] = L.numpy().reshape(depth.shape)
        arrays[f"c{i}/R"] = R.numpy().reshape(depth.shape)
        cases.append(dict(id=f"c{i}", strength=st, edge_threshold=thr, falloff=fo, vert=v, batch=B))
    digests = []
    for (H, W, kind, st, thr, fo, v) in [(540, 960, "blobs", 20, 20, 2.0, 6), (1080, 1920, "blobs", 20, 20, 2.0, 6)]:
        depth = np.round(synth.DEPTHS[kind](H, W, seed=1) * 255).astype(np.float32)
        L, R = sig.directional_motion_blur_gpu(torch.from_numpy(depth.copy()), st, thr, st, falloff_exponent=fo,
                                               vert_smooth_px=v)
        digests.append(dict(h=H, w=W, kind=kind, seed=1, strength=st, edge_threshold=thr, falloff=fo, vert=v,
                            depth_sha=sha(depth), L=sha(L.numpy()), R=sha(R.numpy())))
    np.savez_compressed(os.path.join(OUT, "blur.npz"), meta=json.dumps(dict(cases=cases, digests=digests)), **arrays)
    print("blur:", len(cases), "cases +", len(digests), "digests")


def gen_forward_warp(sig):
    """Fixture for forward_warp_gpu (reference :277-450) on CPU torch."""
    arrays, cases = {}, []
    specs = [(64, 96, 2, "stepped", 16.8, 0.0, 2.0, 0.5, 255.0), (64, 96, 1, "blobs", -16.8, 1.5, 2.0, 0.5, 1.0),
             (48, 160, 1, "noisy_ramp", 12.0, -1.0, 1.0, 0.3, 255.0), (32, 128, 1, "random8", -7.0, 0.0, 0.5, 0.7, 255.0),
             (48, 160, 1, "blobs", 10.0, 0.0, 1.3, 0.5, 255.0)]
    for i, (H, W, B, kind, dpx, spx, e, conv, scale) in enumerate(specs):
        d8 = np.stack([np.round((synth.DEPTHS[kind](H, W, cx=W / 2 + 7 * j) if kind in ("radial", "stepped")
                                 else synth.DEPTHS[kind](H, W, seed=j)) * 255) for j in range(B)]).astype(np.uint8)
        depth = d8.astype(np.float32) * np.float32(scale / 255.0) if scale == 1.0 else d8.astype(np.float32)
        if scale == 1.0:
            depth = d8.astype(np.float32) / np.float32(255.0)
        img8 = np.random.default_rng(40 + i).integers(0, 256, (B, 3, H, W), dtype=np.uint8)
        img = img8.astype(np.float32) / np.float32(255.0)
        wr, mr = sig.forward_warp_gpu(torch.from_numpy(img), torch.from_numpy(depth), dpx, spx, e, conv)
        arrays[f"c{i}/img_u8"] = img8
        arrays[f"c{i}/depth_u8"] = d8
        arrays[f"c{i}/warped"] = wr.numpy()
        arrays[f"c{i}/mask"] = np.packbits(mr.numpy())
        cases.append(dict(id=f"c{i}", divergence_px=dpx, separation_px=spx, exponent=e, convergence=conv,
                          depth_scale=scale, shape=[B, H, W]))
    np.savez_compressed(os.path.join(OUT, "forward_warp_gpu.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("forward_warp_gpu:", len(cases), "cases")


def gen_forward_warp_params(sig):
    """forward_warp_gpu with its two keyword parameters away from their defaults (reference :277-279: gradient_threshold --
    connectivity, :339-340; max_stretch -- scatter rounds, :365): round 4, leaves forward_warp_gpu.npz untouched."""
    arrays, cases = {}, []
    specs = [(48, 160, 2, "blobs", 14.0, 0.0, 2.0, 0.5, 255.0, 0.6, 8), (48, 160, 1, "stepped", -16.8, 1.0, 2.0, 0.5, 255.0, 3.0, 8),
             (64, 96, 1, "noisy_ramp", 12.0, -1.0, 1.0, 0.3, 255.0, 1.5, 2), (32, 128, 1, "random8", -9.0, 0.0, 0.5, 0.7, 255.0, 6.5, 12),
             (48, 160, 1, "blobs", 20.0, 0.0, 2.0, 0.5, 1.0, 2.0, 3), (40, 200, 1, "stepped", 30.0, 0.0, 2.0, 0.4, 255.0, 12.5, 20),
             (32, 96, 1, "blobs", 10.0, 0.0, 2.0, 0.5, 255.0, 1.5, 0), (32, 96, 1, "blobs", 10.0, 0.0, 2.0, 0.5, 255.0, 0.0, 8),
             (48, 160, 1, "noisy_ramp", -14.0, 0.5, 2.0, 0.5, 255.0, 1.5, 3), (48, 160, 1, "blobs", 14.0, 0.0, 2.0, 0.5, 255.0, 1.0, 8)]
    for i, (H, W, B, kind, dpx, spx, e, conv, scale, thr, ms) in enumerate(specs):
        d8 = np.stack([np.round((synth.DEPTHS[kind](H, W, cx=W / 2 + 7 * j) if kind in ("radial", "stepped")
                                 else synth.DEPTHS[kind](H, W, seed=j)) * 255) for j in range(B)]).astype(np.uint8)
        depth = d8.astype(np.float32) / np.float32(255.0) if scale == 1.0 else d8.astype(np.float32)
        img8 = np.random.default_rng(140 + i).integers(0, 256, (B, 3, H, W), dtype=np.uint8)
        img = img8.astype(np.float32) / np.float32(255.0)
        wr, mr = sig.forward_warp_gpu(torch.from_numpy(img), torch.from_numpy(depth), dpx, spx, e, conv, gradient_threshold=thr,
                                      max_stretch=ms)
        arrays[f"c{i}/img_u8"] = img8
        arrays[f"c{i}/depth_u8"] = d8
        arrays[f"c{i}/warped"] = wr.numpy()
        arrays[f"c{i}/mask"] = np.packbits(mr.numpy())
        cases.append(dict(id=f"c{i}", divergence_px=dpx, separation_px=spx, exponent=e, convergence=conv, depth_scale=scale,
                          gradient_threshold=thr, max_stretch=ms, shape=[B, H, W]))
    np.savez_compressed(os.path.join(OUT, "forward_warp_params.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("forward_warp_params:", len(cases), "cases")


def smooth_image(n, h, w, seed):
    """Low-entropy 8-bit image batch (keeps the committed fixture small); includes black pixels."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    frames = []
    for i in range(n):
        r = (x * 255 // max(w - 1, 1) + 13 * i) % 256
        g = (y * 255 // max(h - 1, 1)) % 256
        b = ((x // 8 + y // 8) % 2) * 200 + 20
        f = np.stack([r, g, b], -1).astype(np.uint8)
        f[rng.integers(0, h, 12), rng.integers(0, w, 12)] = 0
        frames.append(f)
    return np.stack(frames)


def gen_node(gs):
    """Fixture for StereoImageNode.generate (reference GenerateStereo.py:79-353): every UI technique x mode."""
    node = gs.StereoImageNode()
    arrays, cases = {}, []

    def run(cid, img8, depth, fill, mode, div=8.0, sep=0.0, bal=0.0, conv=0.5, e=2.0, thr=20.0, st=20.0, blur=False, **kw):
        image = torch.from_numpy(img8.astype(np.float32) / np.float32(255.0))
        outs = node.generate(image, torch.from_numpy(depth), div, sep, mode, bal, conv, e, UI_FILLS[fill], thr, st, blur, **kw)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        if fill == "gpu_warp":
            arrays[f"{cid}/stereo"] = stereo
            arrays[f"{cid}/dl"] = dl[..., 0].copy()
            arrays[f"{cid}/dr"] = dr[..., 0].copy()
        else:  # CPU techniques: every output value is k/255 exactly -> store k
            for name, a in (("stereo", stereo), ("dl", dl[..., 0]), ("dr", dr[..., 0])):
                k = np.round(a * 255.0).astype(np.uint8)
                assert np.array_equal(k.astype(np.float32) / np.float32(255.0), a)
                arrays[f"{cid}/{name}_u8"] = k
            assert np.array_equal(dl[..., 0], dl[..., 1]) and np.array_equal(dl[..., 0], dl[..., 2])
        assert set(np.unique(mask)) <= {0.0, 1.0}
        arrays[f"{cid}/mask"] = np.packbits(mask.astype(bool))
        cases.append(dict(id=cid, fill=fill, mode=mode, divergence=div, separation=sep, balance=bal, convergence=conv,
                          exponent=e, edge_threshold=thr, strength=st, blur=blur, kw=kw,
                          shapes=dict(stereo=list(stereo.shape), depth=list(dl.shape), mask=list(mask.shape))))

    # (i) every technique x mode, N=2, 32x64, blur off (tiny frames + blur take a different oneDNN kernel)
    img8 = np.stack([synth.image_u8(32, 64, seed=s) for s in (1, 2)])
    depth = synth.depth_batch("stepped", 2, 32, 64, channels=3)
    arrays["small/img_u8"] = img8
    arrays["small/depth"] = depth
    for fill in UI_FILLS:
        for mode in MODES:
            run(f"small/{fill}/{mode}", img8, depth, fill, mode, batch_size=12)
    # (ii) parameter variants on the same inputs: balance, separation, exponent, convergence, sub-batching (Q9/Q10)
    img8 = np.stack([synth.image_u8(24, 80, seed=s) for s in (3, 4, 5)])
    depth1 = synth.depth_batch("blobs", 3, 24, 80, channels=1)
    arrays["var/img_u8"] = img8
    arrays["var/depth"] = depth1
    run("var/poly_bal", img8, depth1, "polylines_soft", "left-right", div=6.0, bal=0.4, sep=0.5, conv=0.35, e=1.3, batch_size=2)
    run("var/none_onesided", img8, depth1, "none", "red-cyan-anaglyph", div=0.05, bal=0.95, batch_size=1)
    run("var/naive_interp", img8, depth1, "naive_interpolating", "top-bottom", div=9.0, bal=-0.3, e=0.5, conv=1.0)
    run("var/gpu_defaults", img8, depth1, "gpu_warp", "left-right", div=4.5, batch_size=2)
    run("var/hybrid", img8, depth1, "hybrid_edge", "right-left", div=7.0, sep=-0.7)
    # (iii) blur ON at 256x256 (bit-reproducible conv path), smooth image so the fixture stays small
    img8 = smooth_image(2, 256, 256, 9)
    depth8 = np.stack([np.round(synth.blobs(256, 256, seed=j) * 255) for j in range(2)]).astype(np.uint8)
    depthb = np.repeat((depth8.astype(np.float32) / np.float32(255.0))[..., None], 3, -1)
    arrays["blur/img_u8"] = img8
    arrays["blur/depth_u8"] = depth8
    kw = dict(depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    run("blur/polylines_soft", img8, depthb, "polylines_soft", "left-right", div=4.5, blur=True, **kw)
    run("blur/none_anaglyph", img8, depthb, "none", "red-cyan-anaglyph", div=4.5, blur=True, **kw)
    run("blur/gpu_warp", img8, depthb, "gpu_warp", "left-right", div=4.5, blur=True, **kw)
    # (iv) depth size != image size (bilinear resize, GenerateStereo.py:214-220) -- tolerance-checked
    img8 = np.stack([synth.image_u8(32, 64, seed=8)])
    depth_small = synth.depth_batch("radial", 1, 20, 36, channels=3)
    arrays["resize/img_u8"] = img8
    arrays["resize/depth"] = depth_small
    run("resize/none", img8, depth_small, "none", "left-right", div=5.0)
    np.savez_compressed(os.path.join(OUT, "node_generate.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("node_generate:", len(cases), "cases")


def gen_node_extra(gs):
    """Second node fixture (added in round 2, leaves node_generate.npz untouched): the three modes only an API workflow can
    pass (left-only, only-right, cyan-red-reverseanaglyph: reference stereoimage_generation.py:1543-1562, :1094-1120), the three
    fill strings the combo list leaves out (GenerateStereo.py:97-99), and depth maps whose size differs from the image's
    (bilinear resize, GenerateStereo.py:141-148 / 214-220) on both of torch's resize loops (<= 4096 output pixels and more)."""
    node = gs.StereoImageNode()
    arrays, cases = {}, []

    def run(cid, grp, fill_ui, mode, div=6.0, sep=0.2, bal=0.1, conv=0.5, e=2.0, thr=20.0, st=20.0, blur=False, **kw):
        image = torch.from_numpy(arrays[f"{grp}/img_u8"].astype(np.float32) / np.float32(255.0))
        depth = torch.from_numpy(arrays[f"{grp}/depth"])
        outs = node.generate(image, depth, div, sep, mode, bal, conv, e, fill_ui, thr, st, blur, **kw)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        gpu = fill_ui == "GPU Warp (Fast)"
        if gpu:
            arrays[f"{cid}/stereo"], arrays[f"{cid}/dl"], arrays[f"{cid}/dr"] = stereo, dl[..., 0].copy(), dr[..., 0].copy()
        else:
            for name, a in (("stereo", stereo), ("dl", dl[..., 0]), ("dr", dr[..., 0])):
                k = np.round(a * 255.0).astype(np.uint8)
                assert np.array_equal(k.astype(np.float32) / np.float32(255.0), a)
                arrays[f"{cid}/{name}_u8"] = k
        arrays[f"{cid}/mask"] = np.packbits(mask.astype(bool))
        cases.append(dict(id=cid, group=grp, fill_ui=fill_ui, gpu=gpu, mode=mode, divergence=div, separation=sep, balance=bal,
                          convergence=conv, exponent=e, edge_threshold=thr, strength=st, blur=blur, kw=kw,
                          shapes=dict(stereo=list(stereo.shape), depth=list(dl.shape), mask=list(mask.shape))))

    arrays["api/img_u8"] = np.stack([synth.image_u8(32, 72, seed=s) for s in (11, 12)])
    arrays["api/depth"] = synth.depth_batch("blobs", 2, 32, 72, channels=3)
    for mode in ("left-only", "only-right", "cyan-red-reverseanaglyph"):
        for ui in ("No fill", "Fill - Polylines Soft", "Imperfect fill - Hybrid Edge", "GPU Warp (Fast)"):
            run(f"api/{mode}/{ui}", "api", ui, mode, batch_size=12)
    for ui in ("Fill - Post-fill", "Fill - Reverse projection with Post-fill", "Fill - Hybrid Edge with fill"):
        run(f"api/hidden/{ui}", "api", ui, "left-right", batch_size=12)
    # resize: (group, image h x w, depth h x w, channels)
    for grp, (h, w), (dh, dw), ch in (("rs_small_up", (32, 64), (20, 36), 3), ("rs_big_up", (96, 160), (48, 64), 3),
                                      ("rs_big_down", (72, 96), (108, 160), 1), ("rs_odd", (45, 91), (33, 57), 3)):
        arrays[f"{grp}/img_u8"] = np.stack([synth.image_u8(h, w, seed=s) for s in (21, 22, 23)])
        arrays[f"{grp}/depth"] = synth.depth_batch("radial" if grp != "rs_big_up" else "blobs", 3, dh, dw, channels=ch)
        for ui, mode in (("No fill", "left-right"), ("Fill - Polylines Soft", "top-bottom"), ("GPU Warp (Fast)", "left-right")):
            run(f"{grp}/{ui}", grp, ui, mode, div=5.0, batch_size=2)
    np.savez_compressed(os.path.join(OUT, "node_extra.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("node_extra:", len(cases), "cases")


def gen_scene8(gs):
    """the reference's node on depth of the kind its own fixture maker draws (create_test_images.py:3-77: an 8-bit
    gradient + flat ellipses with hard silhouettes) -- tools/synth.scene8, hard and softened silhouettes -- for every UI technique,
    blur off and on, 2 x 256 x 256; plus SHA-256 digests of one 1080p frame (polylines_soft, blur on)."""
    node = gs.StereoImageNode()
    arrays, cases = {}, []
    img8 = smooth_image(1, 256, 256, 19)   # (one frame per case, and every 8th row of the float32 gpu_warp colours: a small fixture)
    arrays["img_u8"] = img8
    image = torch.from_numpy(img8.astype(np.float32) / np.float32(255.0))
    for grp, soften in (("hard", False), ("soft", True)):
        depth8 = np.stack([np.round(synth.scene8(256, 256, seed=j, soften=soften) * 255) for j in range(1)]).astype(np.uint8)
        arrays[f"{grp}/depth_u8"] = depth8
        depth = torch.from_numpy(np.repeat((depth8.astype(np.float32) / np.float32(255.0))[..., None], 3, -1))
        for fill in UI_FILLS:
            for blur in ((False, True) if grp == "soft" else (False,)):
                mode = "left-right" if fill != "none" else "red-cyan-anaglyph"
                cid = f"{grp}/{fill}/blur{int(blur)}"
                kw = dict(depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
                outs = node.generate(image, depth, 6.0, 0.0, mode, 0.0, 0.5, 2.0, UI_FILLS[fill], 20.0, 20.0, blur, **kw)
                stereo, dl, dr, mask = [o.numpy() for o in outs]
                if fill == "gpu_warp":
                    arrays[f"{cid}/stereo_rows"], arrays[f"{cid}/dl"], arrays[f"{cid}/dr"] = stereo[:, ::8].copy(), dl[..., 0].copy(), dr[..., 0].copy()
                else:
                    for name, a in (("stereo", stereo), ("dl", dl[..., 0]), ("dr", dr[..., 0])):
                        k = np.round(a * 255.0).astype(np.uint8)
                        assert np.array_equal(k.astype(np.float32) / np.float32(255.0), a)
                        arrays[f"{cid}/{name}_u8"] = k
                arrays[f"{cid}/mask"] = np.packbits(mask.astype(bool))
                cases.append(dict(id=cid, group=grp, fill=fill, row_step=8, mode=mode, divergence=6.0, separation=0.0, balance=0.0, convergence=0.5,
                                  exponent=2.0, edge_threshold=20.0, strength=20.0, blur=blur, kw=kw,
                                  shapes=dict(stereo=list(stereo.shape), depth=list(dl.shape), mask=list(mask.shape))))
                print("scene8", cid, flush=True)
    h, w = 1080, 1920
    img1 = synth.image_u8(h, w, seed=1000, hazards=False)[None]
    depth = synth.depth_batch("scene8", 1, h, w, channels=3)
    outs = node.generate(torch.from_numpy(img1.astype(np.float32) / np.float32(255.0)), torch.from_numpy(depth), 3.5, 0.0, "left-right", 0.0,
                         0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True, depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    stereo, dl, dr, mask = [o.numpy() for o in outs]
    k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
    digest = dict(h=h, w=w, kind="scene8", fill_ui="Fill - Polylines Soft", mode="left-right", divergence=3.5, blur=True, image_seed=1000,
                  stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))), mask_sum=int(mask.sum()))
    np.savez_compressed(os.path.join(OUT, "scene8.npz"), meta=json.dumps(dict(cases=cases, digest_1080p=digest)), **arrays)
    print("scene8:", len(cases), "cases; 1080p digest", digest["stereo_u8"][:16])


def gen_digests(gs):
    """SHA-256 digests of the reference's node outputs at BASELINE.json sizes: inputs come from tools/synth.py
    (seeded), so only the digests are committed.  Minutes of pure-Python reference time per case."""
    node = gs.StereoImageNode()
    out = {}
    for cid, (h, w, kind, ui, mode, div, blur) in {
            "cfg1_512_naive_interpolating": (512, 512, "radial", "Fill - Naive interpolating", "left-right", 4.5, True),
            "cfg2_1080p_polylines_soft": (1080, 1920, "stepped", "Fill - Polylines Soft", "left-right", 3.5, True),
            "cfg3q_1080p_hybrid_edge": (1080, 1920, "stepped", "Imperfect fill - Hybrid Edge", "left-right", 8.0, True)}.items():
        img8 = synth.image_u8(h, w, seed=1000, hazards=False)[None]
        depth = synth.depth_batch(kind, 1, h, w, channels=3)
        image = torch.from_numpy(img8.astype(np.float32) / np.float32(255.0))
        outs = node.generate(image, torch.from_numpy(depth), div, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind=kind, fill_ui=ui, mode=mode, divergence=div, blur=blur, image_seed=1000,
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()))
        print("digest", cid, out[cid]["stereo_u8"][:16])
    with open(os.path.join(OUT, "digests.json"), "w") as f:
        json.dump(out, f, indent=1)


WIDE_CASES = {   # id: (h, w, depth kind, UI string, mode, divergence, blur) -- the widths round 6 opened
    "naive_interp_8k_anaglyph": (8, 7680, "stepped", "Fill - Naive interpolating", "red-cyan-anaglyph", 4.0, False),
    "naive_interp_9536_anaglyph_blur": (8, 9536, "stepped", "Fill - Naive interpolating", "cyan-red-reverseanaglyph", 4.0, True),
    "naive_10240_anaglyph": (8, 10240, "stepped", "Fill - Naive", "red-cyan-anaglyph", 4.0, False),
    "inverse_9004_anaglyph": (8, 9004, "stepped", "No fill - Reverse projection", "cyan-red-reverseanaglyph", 4.0, False),
    "none_10240_anaglyph_blur": (8, 10240, "stepped", "No fill", "red-cyan-anaglyph", 4.0, True),
    "none_post_10240_anaglyph": (8, 10240, "stepped", "Fill - Post-fill", "red-cyan-anaglyph", 4.0, False),
    "inverse_post_8192_anaglyph_blur": (8, 8192, "stepped", "Fill - Reverse projection with Post-fill", "cyan-red-reverseanaglyph", 4.0, True),
    "none_post_11578_sbs": (6, 11578, "stepped", "Fill - Post-fill", "right-left", 6.0, False),
    "inverse_post_9004_sbs": (6, 9004, "stepped", "Fill - Reverse projection with Post-fill", "right-left", 6.0, False),
    "naive_11578_sbs": (6, 11578, "stepped", "Fill - Naive", "right-left", 6.0, False),
    "naive_interp_9536_sbs": (6, 9536, "stepped", "Fill - Naive interpolating", "right-left", 6.0, False),
    "sharp_8192_sbs_clipped": (6, 8192, "clipped", "Fill - Polylines Sharp", "left-right", 5.0, False),
    "sharp_8192_anaglyph_clipped": (6, 8192, "clipped", "Fill - Polylines Sharp", "red-cyan-anaglyph", 5.0, False),
    "sharp_8192_tb_scene8": (6, 8192, "scene8", "Fill - Polylines Sharp", "top-bottom", 5.0, False),
}


def gen_wide(gs):
    """SHA-256 digests of the reference's node outputs on thin rows at the widths round 6 opened (anaglyphs beyond the row kernel's stash
    form, the post-fill and naive techniques at their new limits, polylines_sharp at 8 192 columns): inputs from tools/synth.py (seeded),
    digests only.  tests/test_oracle_goldens.py checks the oracle against them on the CPU, tests/test_gpu_fullsize.py the HIP path."""
    node = gs.StereoImageNode()
    out = {}
    for cid, (h, w, kind, ui, mode, div, blur) in WIDE_CASES.items():
        img = synth.image_f32(1, h, w, seed=8)
        img[:, :, 500:560] = 0.0   # genuinely black pixels (mask, quirk Q6)
        depth = synth.depth_batch(kind, 1, h, w, channels=3)
        outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), div, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind=kind, fill_ui=ui, mode=mode, divergence=div, blur=blur, image_seed=8, black=[500, 560],
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()))
        print("wide digest", cid, out[cid]["stereo_u8"][:16], flush=True)
    with open(os.path.join(OUT, "digests_wide.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_metric_4k(gs):
    """SHA-256 digests of the REFERENCE node's outputs on the metric's own frame (BASELINE.json configs[1]: 4K, polylines_soft, left-right
    SBS, divergence 8, stepped depth, blur on; the inputs of tests/test_gpu_fullsize.py::test_4k_frame_bit_exact_vs_oracle).  The pure-Python
    reference takes a quarter of an hour for this one frame; digests only."""
    node = gs.StereoImageNode()
    h, w = 2160, 3840
    img = synth.image_f32(1, h, w, seed=1)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft",
                         20.0, 20.0, True, depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    stereo, dl, dr, mask = [o.numpy() for o in outs]
    k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
    out = dict(h=h, w=w, kind="stepped", fill_ui="Fill - Polylines Soft", mode="left-right", divergence=8.0, blur=True, image_seed=1,
               stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))), mask_sum=int(mask.sum()),
               stereo_f32=sha(stereo), dl_f32=sha(dl), dr_f32=sha(dr))
    with open(os.path.join(OUT, "digest_metric_4k.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("metric 4K digest", out["stereo_u8"][:16], flush=True)


CASES_4K = {   # the other bench configurations at full size, one frame each (bench.py CONFIGS; gpu_warp is tolerance-checked, not hashed)
    "cfg3_hybrid_edge": ("Imperfect fill - Hybrid Edge", "left-right"),
    "cfg5_none_anaglyph": ("No fill", "red-cyan-anaglyph"),
    "naive_interpolating": ("Fill - Naive interpolating", "left-right"),
    "polylines_sharp": ("Fill - Polylines Sharp", "left-right"),
}


def gen_4k(gs, only=None):
    """SHA-256 digests of the REFERENCE node's outputs on one 4K frame of every other bench configuration (BASELINE.json configs[2], [4]
    and the two fills bench.py adds; divergence 8, stepped depth, blur on; inputs as for digest_metric_4k.json).  Minutes of the
    pure-Python reference per case; `CS_GOLDEN_CASE=<id>` generates one case into digests_4k.<id>.json (cases in parallel processes,
    merged by hand into digests_4k.json: the result does not depend on the order)."""
    node = gs.StereoImageNode()
    h, w = 2160, 3840
    img = synth.image_f32(1, h, w, seed=1)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    out = {}
    for cid, (ui, mode) in CASES_4K.items():
        if only and cid != only:
            continue
        outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), 8.0, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, True,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind="stepped", fill_ui=ui, mode=mode, divergence=8.0, blur=True, image_seed=1,
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()), stereo_f32=sha(stereo), dl_f32=sha(dl), dr_f32=sha(dr))
        print("4K digest", cid, out[cid]["stereo_u8"][:16], flush=True)
    with open(os.path.join(OUT, f"digests_4k.{only}.json" if only else "digests_4k.json"), "w") as f:
        json.dump(out, f, indent=1)


CASES_TIES = {   # id: (h, w, depth kind, UI string) -- order-dependent rows at full width, blur off (ties only survive without the blur)
    "soft_clipped_4k": (2160, 3840, "clipped", "Fill - Polylines Soft"),
    "sharp_clipped_4k": (2160, 3840, "clipped", "Fill - Polylines Sharp"),
    "soft_random8_rows": (48, 3840, "random8", "Fill - Polylines Soft"),
    "sharp_random8_rows": (48, 3840, "random8", "Fill - Polylines Sharp"),
}


def gen_ties(gs, only=None):
    """SHA-256 digests of the REFERENCE node's outputs on ORDER-DEPENDENT depth at full width: a 4K frame saturated to exact 0 / 1 over
    large areas (exact closeness ties under every fold: the stretch replay) and 48 rows of 8-bit noise (no reset points: every row replayed
    whole), polylines_soft and polylines_sharp, divergence 8, left-right, blur off.  `CS_GOLDEN_CASE=<id>` as for --only-4k."""
    node = gs.StereoImageNode()
    out = {}
    for cid, (h, w, kind, ui) in CASES_TIES.items():
        if only and cid != only:
            continue
        img = synth.image_f32(1, h, w, seed=1)
        depth = synth.depth_batch(kind, 1, h, w, channels=3)
        outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, ui, 20.0, 20.0, False,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind=kind, fill_ui=ui, mode="left-right", divergence=8.0, blur=False, image_seed=1,
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()), stereo_f32=sha(stereo), dl_f32=sha(dl), dr_f32=sha(dr))
        print("tie digest", cid, out[cid]["stereo_u8"][:16], flush=True)
    with open(os.path.join(OUT, f"digests_ties.{only}.json" if only else "digests_ties.json"), "w") as f:
        json.dump(out, f, indent=1)


CASES_SCENE8_4K = {   # id: (UI string, blur) -- estimator-like depth at full size (tools/synth.scene8, softened silhouettes, seed 0)
    "soft_scene8_4k_blur": ("Fill - Polylines Soft", True),
    "sharp_scene8_4k_blur": ("Fill - Polylines Sharp", True),
    "sharp_scene8_4k": ("Fill - Polylines Sharp", False),
}


def gen_scene8_4k(gs, only=None):
    """SHA-256 digests of the REFERENCE node's outputs on one 4K frame of scene8 depth (quantised smooth depth with softened object
    silhouettes: what the tile kernels' per-pixel lists overflow on) at the metric's divergence, polylines_soft / polylines_sharp.
    `CS_GOLDEN_CASE=<id>` as for --only-4k."""
    node = gs.StereoImageNode()
    h, w = 2160, 3840
    img = synth.image_f32(1, h, w, seed=1)
    depth = synth.depth_batch("scene8", 1, h, w, channels=3)
    out = {}
    for cid, (ui, blur) in CASES_SCENE8_4K.items():
        if only and cid != only:
            continue
        outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind="scene8", fill_ui=ui, mode="left-right", divergence=8.0, blur=blur, image_seed=1,
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()), stereo_f32=sha(stereo), dl_f32=sha(dl), dr_f32=sha(dr))
        print("scene8 4K digest", cid, out[cid]["stereo_u8"][:16], flush=True)
    with open(os.path.join(OUT, f"digests_scene8_4k.{only}.json" if only else "digests_scene8_4k.json"), "w") as f:
        json.dump(out, f, indent=1)


CASES_8K = {   # id: (UI string, mode, blur) -- one 8K frame (7680 x 4320), divergence 8, stepped depth
    "soft_8k_sbs_blur": ("Fill - Polylines Soft", "left-right", True),
    "naive_interp_8k_anaglyph_blur": ("Fill - Naive interpolating", "red-cyan-anaglyph", True),
}


def gen_8k(gs, only=None):
    """SHA-256 digests of the REFERENCE node's outputs on one 8K frame (7680 x 4320): polylines_soft side by side and naive_interpolating as
    a red-cyan anaglyph (refused until round 6: the row kernel's anaglyph stash did not fit at that width), divergence 8, blur on.  Half an
    hour of the pure-Python reference per case; `CS_GOLDEN_CASE=<id>` as for --only-4k."""
    node = gs.StereoImageNode()
    h, w = 4320, 7680
    img = synth.image_f32(1, h, w, seed=1)
    depth = synth.depth_batch("stepped", 1, h, w, channels=3)
    out = {}
    for cid, (ui, mode, blur) in CASES_8K.items():
        if only and cid != only:
            continue
        outs = node.generate(torch.from_numpy(img), torch.from_numpy(depth), 8.0, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        stereo, dl, dr, mask = [o.numpy() for o in outs]
        k = [np.round(a * 255.0).astype(np.uint8) for a in (stereo, dl[..., 0], dr[..., 0])]
        out[cid] = dict(h=h, w=w, kind="stepped", fill_ui=ui, mode=mode, divergence=8.0, blur=blur, image_seed=1,
                        stereo_u8=sha(k[0]), dl_u8=sha(k[1]), dr_u8=sha(k[2]), mask=sha(np.packbits(mask.astype(bool))),
                        mask_sum=int(mask.sum()), stereo_f32=sha(stereo), dl_f32=sha(dl), dr_f32=sha(dr))
        print("8K digest", cid, out[cid]["stereo_u8"][:16], flush=True)
    with open(os.path.join(OUT, f"digests_8k.{only}.json" if only else "digests_8k.json"), "w") as f:
        json.dump(out, f, indent=1)


WARP_1080P_ROWS = [0, 1, 110, 128, 129, 257, 332, 539, 540, 746, 822, 951, 1078, 1079]


def gen_warp_1080p(sig):
    """forward_warp_gpu at 1080p (where the [-1, 1] coordinate round trip costs most, SURVEY B-16): inputs come from
    tools/synth.py (seeded); committed are the full gap mask (bit-packed) and a handful of output rows."""
    h, w = 1080, 1920
    img = synth.image_f32(1, h, w, seed=6).transpose(0, 3, 1, 2).copy()
    depth = synth.stepped(h, w)[None] * np.float32(255.0)
    arrays, cases = {}, []
    for cid, (dpx, spx, e, conv) in enumerate([(67.2, 1.5, 2.0, 0.5), (-40.0, 0.0, 1.0, 0.3)]):
        warped, mask = sig.forward_warp_gpu(torch.from_numpy(img), torch.from_numpy(depth), dpx, spx, e, conv)
        arrays[f"{cid}/rows"] = warped.numpy()[:, :, WARP_1080P_ROWS, :].copy()
        arrays[f"{cid}/mask"] = np.packbits(mask.numpy())
        cases.append(dict(id=str(cid), divergence_px=dpx, separation_px=spx, exponent=e, convergence=conv, rows=WARP_1080P_ROWS,
                          h=h, w=w, image_seed=6, depth="stepped * 255"))
    np.savez_compressed(os.path.join(OUT, "forward_warp_1080p.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("forward_warp_1080p:", len(cases), "cases")


def load_stereo_utils():
    """stereo_utils.py of the reference alone (torch, einops)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_stereo_utils", refload.REF + "/stereo_utils.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gen_stereo_shift():
    """Fixture for stereo_shift_torch (reference stereo_utils.py:15-88) on latent-shaped inputs."""
    su = load_stereo_utils()
    rng = np.random.default_rng(77)
    arrays, cases = {}, []
    for cid, (b, c, h, w, sf, both, e) in enumerate([(1, 4, 64, 64, 8.0, False, 1.0), (1, 4, 128, 128, 8.0, False, 1.0),
                                                      (2, 4, 64, 64, 12.0, True, 1.0), (1, 4, 48, 80, -9.0, False, 2.0),
                                                      (1, 3, 40, 56, 6.0, True, 0.5), (1, 4, 32, 40, 15.0, False, 1.0),
                                                      (1, 4, 16, 24, 8.0, False, 1.0)]):
        x = rng.standard_normal((b, c, h, w)).astype(np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        d = (np.sin(xx / 7.0) + np.cos(yy / 5.0) + 0.3 * rng.random((b, h, w))).astype(np.float32)
        if cid == 6:
            d[:] = 0.25  # flat depth: the reference normalises it to zeros (:40-43)
        out = su.stereo_shift_torch(torch.from_numpy(x), torch.from_numpy(d), sf, both, e).numpy()
        arrays[f"{cid}/x"], arrays[f"{cid}/d"], arrays[f"{cid}/out"] = x, d, out
        cases.append(dict(id=str(cid), scale_factor=sf, shift_both=both, exponent=e))
    np.savez_compressed(os.path.join(OUT, "stereo_shift.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("stereo_shift:", len(cases), "cases")


def gen_dialect_f64(sig):
    """Indirect pin of the float64 disparity chain of dialect D64: the reference's inner functions
    (apply_stereo_divergence_naive :1850-1910, apply_stereo_divergence_inverse :1715-1737) handed
    normalized_depth.astype(float64) -- `abs(d) ** e`, the products and int() / floor() then run in float64 as under numba,
    while sum() of a uint8 pixel still wraps (no numba here).  Depth maps span exactly 0..1, so the reference's own
    normalisation (:1587-1600) is the identity and the driver reproduces the same normalized_depth from them."""
    rng = np.random.default_rng(4242)
    arrays, cases = {}, []
    h, w = 40, 112
    for cid, (kind, div, sep, e, conv) in enumerate([("stepped", 6.0, 0.0, 2.0, 0.5), ("radial", -7.5, 1.0, 1.3, 0.4),
                                                     ("random8", 9.0, -0.5, 0.7, 0.5), ("noisy_ramp", 12.5, 0.0, 1.0, 0.6),
                                                     ("blobs", 3.3, 2.0, 0.1, 0.3), ("near-integer", 7.3, 0.0, 1.0, 0.0)]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[rng.random((h, w)) < 0.03] = (128, 128, 0)   # channel sum 256: black under the wrapping sum only
        img[rng.random((h, w)) < 0.03] = 0
        if kind == "near-integer":
            # offsets within a few float32 ulps of an integer: where the float32 and the float64 chain truncate differently
            # (searched, not constructed: the candidates are depth values whose float32 chain lands on the other side of
            # the integer than the float64 chain does)
            dpx, c32 = (div / 100.0) * w, np.float32(conv)
            base = (rng.integers(1, 6, (h, w)) / dpx).astype(np.float32) + c32
            depth = base.copy()
            for idx in np.ndindex(h, w):
                if rng.random() < 0.5:
                    continue
                for k in range(-48, 49):
                    cand = (base[idx].view(np.int32) + np.int32(k)).view(np.float32)
                    ndc = cand - c32
                    i32 = int(1.0 * (abs(ndc) ** e) * dpx + 0.0)
                    i64 = int(1.0 * (abs(np.float64(ndc)) ** e) * dpx + 0.0)
                    if i32 != i64:
                        depth[idx] = cand
                        break
        else:
            depth = synth.depth_batch(kind, 1, h, w, channels=1)[0, ..., 0].astype(np.float32)
            depth = (depth - depth.min()) / (depth.max() - depth.min())
        depth[0, 0], depth[0, 1] = 0.0, 1.0
        depth = depth.astype(np.float32)
        nd32 = depth - np.float32(conv)
        nd64 = nd32.astype(np.float64)
        div_px, sep_px = (div / 100.0) * w, (sep / 100.0) * w
        arrays[f"{cid}/img"], arrays[f"{cid}/depth"] = img, depth
        for fill in ("none", "naive", "naive_interpolating"):
            arrays[f"{cid}/{fill}"] = sig.apply_stereo_divergence_naive(img, nd64, div_px, sep_px, e, fill)
        arrays[f"{cid}/inverse"] = sig.apply_stereo_divergence_inverse(img, nd64, div_px, sep_px, e)
        # polylines: the point coordinates leave the float64 chain and are rounded once into the float32 `pt` array (:1924-1934);
        # the sweep after that keeps the no-numba typing (np.float32 array elements)
        for fill in ("polylines_soft", "polylines_sharp"):
            arrays[f"{cid}/{fill}"] = sig.apply_stereo_divergence_polylines(img, nd64, div_px, sep_px, e, fill)
        # hybrid_edge: dest_x, its distance to the column and the exp argument in float64 (:1636-1644)
        arrays[f"{cid}/hybrid_edge"] = sig.apply_stereo_divergence_hybrid_edge(img, nd64, div_px, sep_px, e)
        # (round 5) the three techniques no UI string reaches: their mapping functions run the same chain (:1662-1713)
        arrays[f"{cid}/none_post"] = sig.apply_stereo_divergence_naive_post(img, nd64, div_px, sep_px, e)
        arrays[f"{cid}/inverse_post"] = sig.apply_stereo_divergence_inverse_post(img, nd64, div_px, sep_px, e)
        arrays[f"{cid}/hybrid_edge_plus"] = sig.apply_stereo_divergence_hybrid_edge_plus(img, nd64, div_px, sep_px, e)
        # how often the two dialects disagree on this case (reported by the tests)
        d32 = sig.apply_stereo_divergence_naive(img, nd32, div_px, sep_px, e, "none")
        cases.append(dict(id=str(cid), kind=kind, divergence=div, separation=sep, exponent=e, convergence=conv,
                          pixels_differing_from_d32=int((d32 != arrays[f"{cid}/none"]).any(-1).sum())))
    np.savez_compressed(os.path.join(OUT, "dialect_f64.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("dialect_f64:", len(cases), "cases;", [c["pixels_differing_from_d32"] for c in cases], "pixels differ from D32 ('none')")


def gen_numpy_inputs(sig):
    """create_stereoimages with numpy / PIL inputs, depth blur off (reference :1486-1499, 1527-1574: no x255 scaling, no clip,
    uint8 image used as it is, depth output = clip(depth, 0, 255).astype(uint8)); every technique, several modes."""
    from PIL import Image
    rng = np.random.default_rng(77)
    arrays, cases = {}, []
    for ci, (h, w, kind) in enumerate([(40, 96, "stepped"), (33, 130, "blobs")]):
        img = synth.image_u8(h, w, seed=40 + ci)
        depth = (synth.DEPTHS[kind](h, w) * np.float32(255.0)).astype(np.float32)
        arrays[f"c{ci}/img"] = img
        arrays[f"c{ci}/depth"] = depth
        for fi, fill in enumerate(FILLS):
            modes = [["left-right", "red-cyan-anaglyph"], ["top-bottom"], ["right-left", "only-right"], ["bottom-top", "left-only"]][fi % 4]
            bal = [0.0, 0.3, -0.4][fi % 3]
            kw = dict(divergence=[5.0, 8.0][ci], separation=[0.0, 1.0][fi % 2], modes=modes, stereo_balance=bal,
                      stereo_offset_exponent=[2.0, 1.0, 1.4][fi % 3], fill_technique=fill, convergence_point=[0.5, 0.2][fi % 2])
            as_pil = fi % 2 == 1   # PIL image in, list input for the depth map
            res, mod = sig.create_stereoimages(Image.fromarray(img) if as_pil else img, depth.tolist() if as_pil else depth,
                                               kw["divergence"], kw["separation"], modes, bal, kw["stereo_offset_exponent"], fill,
                                               0.0, 6.0, False, True, kw["convergence_point"])
            cid = f"c{ci}/{fill}"
            for k, r in enumerate(res):
                arrays[f"{cid}/out{k}"] = np.asarray(r)
            arrays[f"{cid}/mod"] = np.asarray(mod)
            cases.append(dict(id=cid, group=f"c{ci}", pil=as_pil, **kw))
    np.savez_compressed(os.path.join(OUT, "create_stereoimages_numpy.npz"), meta=json.dumps(dict(cases=cases)), **arrays)
    print("create_stereoimages with numpy / PIL inputs:", len(cases), "cases")


def gen_numpy_blur(sig):
    """Round 4: the scipy depth blur of the numpy / PIL input path (`directional_motion_blur`, reference :1346-1419, called from
    create_stereoimages :1489-1494): raw blur outputs, and create_stereoimages with numpy / PIL inputs and the blur ON."""
    from PIL import Image
    arrays, cases, cs_cases = {}, [], []
    specs = [(64, 200, "blobs", 20.0, 20.0, 2.0, 6), (48, 160, "stepped", 5.4, 3.0, 1.0, 3), (40, 120, "noisy_ramp", 7.0, 0.5, 0.5, 0),
             (33, 97, "blobs", 21.0, 6.0, 2.0, 15), (48, 130, "clipped", 33.0, 12.0, 2.0, 2), (20, 64, "random8", 4.0, 2.0, 1.0, 1),
             (40, 150, "blobs", 9.6, 4.0, 1.7, 4), (36, 140, "stepped", 12.0, 8.0, 3.0, 0), (9, 40, "blobs", 2.0, 1.0, 2.0, 7),
             (30, 90, "blobs", 1.4, 1.0, 2.0, 0)]
    for i, (h, w, kind, strength, thr, falloff, vert) in enumerate(specs):
        depth = np.round((synth.DEPTHS[kind](h, w, seed=i) if kind not in ("radial", "stepped") else synth.DEPTHS[kind](h, w)) * 255).astype(np.float32)
        if i == 5:
            depth = depth / np.float32(255.0)   # a 0..1 map: the numpy path does NOT rescale it (reference :1488)
            thr = 0.004
        left, right = sig.directional_motion_blur(depth, strength, thr, strength, falloff_exponent=falloff, vert_smooth_px=vert)
        arrays[f"b{i}/depth"] = depth
        arrays[f"b{i}/left"] = np.asarray(left, dtype=np.float32)
        arrays[f"b{i}/right"] = np.asarray(right, dtype=np.float32)
        assert left.dtype == np.float32
        cases.append(dict(id=f"b{i}", strength=strength, edge_threshold=thr, falloff=falloff, vert=vert, exact_power=falloff in (2.0, 1.0, 0.5)))
    for ci, (h, w, kind, fill, modes, strength, thr, falloff, vert, as_pil) in enumerate([
            (40, 128, "blobs", "polylines_soft", ["left-right"], 20.0, 20.0, 2.0, 6, False),
            (36, 120, "stepped", "naive_interpolating", ["top-bottom", "red-cyan-anaglyph"], 6.0, 3.0, 1.0, 2, True),
            (33, 96, "blobs", "hybrid_edge", ["right-left"], 10.0, 6.0, 0.5, 0, False),
            (40, 128, "noisy_ramp", "polylines_sharp", ["left-right", "only-right"], 8.0, 2.0, 2.0, 3, True),
            (32, 100, "blobs", "none", ["bottom-top"], 5.0, 4.0, 2.0, 1, False)]):
        img = synth.image_u8(h, w, seed=60 + ci)
        depth = (synth.DEPTHS[kind](h, w) * np.float32(255.0)).astype(np.float32)
        kw = dict(divergence=[5.0, 8.0][ci % 2], separation=[0.0, 1.0][ci % 2], modes=modes, stereo_balance=[0.0, 0.3, -0.4][ci % 3],
                  stereo_offset_exponent=[2.0, 1.0][ci % 2], fill_technique=fill, convergence_point=0.5,
                  depth_blur_strength=strength, depth_blur_edge_threshold=thr, depth_blur_falloff=falloff, depth_blur_vert_smooth=vert)
        res, ml, mr = sig.create_stereoimages(Image.fromarray(img) if as_pil else img, depth.tolist() if as_pil else depth,
                                              kw["divergence"], kw["separation"], modes, kw["stereo_balance"], kw["stereo_offset_exponent"],
                                              fill, strength, thr, True, True, 0.5, falloff, vert)
        arrays[f"s{ci}/img"] = img
        arrays[f"s{ci}/depth"] = depth
        for k, r in enumerate(res):
            arrays[f"s{ci}/out{k}"] = np.asarray(r)
        arrays[f"s{ci}/mod_left"] = np.asarray(ml)
        arrays[f"s{ci}/mod_right"] = np.asarray(mr)
        cs_cases.append(dict(id=f"s{ci}", pil=as_pil, **kw))
    np.savez_compressed(os.path.join(OUT, "numpy_blur.npz"), meta=json.dumps(dict(cases=cases, create_stereoimages=cs_cases)), **arrays)
    print("scipy depth blur (numpy / PIL inputs):", len(cases), "blur cases,", len(cs_cases), "create_stereoimages cases")


def main():
    os.makedirs(OUT, exist_ok=True)
    refload.quiet()
    torch.manual_seed(0)
    sig = refload.load_sig()
    gs = refload.load_node()
    manifest = dict(dialect="D32 (numba absent, NumPy NEP-50 scalar promotion)", python=platform.python_version(),
                    numpy=np.__version__, torch=torch.__version__, torch_threads=torch.get_num_threads(),
                    glibc=platform.libc_ver()[1], machine=platform.machine(),
                    reference="Dobidop/ComfyStereo v2.1.3 @ /root/reference (read-only)",
                    generator="tools/make_goldens.py")
    if "--only-hidden" in sys.argv:  # (added after the first fixtures were committed: leaves them untouched)
        gen_hidden(sig)
        return
    if "--only-stereo-shift" in sys.argv:
        gen_stereo_shift()
        return
    if "--only-node-extra" in sys.argv:
        gen_node_extra(gs)
        return
    if "--only-warp-1080p" in sys.argv:
        gen_warp_1080p(sig)
        return
    if "--only-digests" in sys.argv:
        gen_digests(gs)
        return
    if "--only-numpy-inputs" in sys.argv:
        gen_numpy_inputs(sig)
        return
    if "--only-numpy-blur" in sys.argv:
        gen_numpy_blur(sig)
        return
    if "--only-warp-params" in sys.argv:
        gen_forward_warp_params(sig)
        return
    if "--only-dialect" in sys.argv:
        gen_dialect_f64(sig)
        return
    if "--only-scene8" in sys.argv:
        gen_scene8(gs)
        return
    if "--only-wide" in sys.argv:
        gen_wide(gs)
        return
    if "--only-metric-4k" in sys.argv:
        gen_metric_4k(gs)
        return
    if "--only-4k" in sys.argv:
        gen_4k(gs, os.environ.get("CS_GOLDEN_CASE") or None)
        return
    if "--only-scene8-4k" in sys.argv:
        gen_scene8_4k(gs, os.environ.get("CS_GOLDEN_CASE") or None)
        return
    if "--only-8k" in sys.argv:
        gen_8k(gs, os.environ.get("CS_GOLDEN_CASE") or None)
        return
    if "--only-ties" in sys.argv:
        gen_ties(gs, os.environ.get("CS_GOLDEN_CASE") or None)
        return
    gen_apply_stereo_divergence(sig)
    gen_hidden(sig)
    gen_blur(sig)
    gen_forward_warp(sig)
    gen_node(gs)
    gen_stereo_shift()
    gen_node_extra(gs)
    gen_warp_1080p(sig)
    gen_digests(gs)
    gen_dialect_f64(sig)
    gen_numpy_inputs(sig)
    gen_forward_warp_params(sig)
    gen_numpy_blur(sig)
    gen_scene8(gs)
    gen_wide(gs)
    gen_metric_4k(gs)
    gen_4k(gs)
    gen_ties(gs)
    gen_scene8_4k(gs)
    gen_8k(gs)
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("fixtures written to", OUT, "total bytes", tot)


if __name__ == "__main__":
    main()
