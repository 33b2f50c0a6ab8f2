"""The C-ABI library loads on a GPU-less host and exports every symbol include/comfystereo_amd.h declares;
argument validation that needs no device work behaves as documented (not gpu)."""
import ctypes
import os
import re

import pytest

from comfystereo_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "comfystereo_amd.h")).read()
    return sorted(set(re.findall(r"CS_API\s+[\w\s\*]+?\b(cs_\w+)\s*\(", hdr)))


def test_library_is_built_and_exports_every_declared_symbol():
    L = _native.lib()
    syms = declared_symbols()
    assert len(syms) >= 14 and set(syms) == set(_native.EXPORTS)
    for s in syms:
        assert hasattr(L, s), s
    assert L.cs_version() == _native.ABI_VERSION == 4


def test_enums_match_header():
    hdr = open(os.path.join(ROOT, "include", "comfystereo_amd.h")).read()
    for key, name in (("none", "CS_FILL_NONE"), ("polylines_soft", "CS_FILL_POLYLINES_SOFT"), ("gpu_warp", "CS_FILL_GPU_WARP"),
                      ("hybrid_edge", "CS_FILL_HYBRID_EDGE")):
        assert int(re.search(name + r"\s*=\s*(\d+)", hdr).group(1)) == _native.FILL[key]
    for key, name in (("left-right", "CS_MODE_LEFT_RIGHT"), ("red-cyan-anaglyph", "CS_MODE_RED_CYAN_ANAGLYPH"),
                      ("cyan-red-reverseanaglyph", "CS_MODE_CYAN_RED_REVERSEANAGLYPH")):
        assert int(re.search(name + r"\s*=\s*(\d+)", hdr).group(1)) == _native.MODE[key]


def test_params_struct_layout():
    assert ctypes.sizeof(_native.Params) == 12 * 4 + 8 * 8
    assert _native.Params.divergence.offset == 48


def test_host_side_queries_without_gpu():
    L = _native.lib()
    from comfystereo_amd import engine
    p = engine.make_params(4, 2160, 3840, 2160, 3840, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, True, 20.0,
                           20.0, 2.0, 6, 12)
    assert engine.output_shape(p) == (2160, 7680, 2160, 7680)
    p.mode = _native.MODE["top-bottom"]
    assert engine.output_shape(p) == (4320, 3840, 4320, 3840)
    p.mode, p.fill = _native.MODE["red-cyan-anaglyph"], _native.FILL["gpu_warp"]
    assert engine.output_shape(p) == (2160, 3840, 2160, 3840)
    assert L.cs_workspace_bytes(ctypes.byref(p)) > 4 * 2160 * 3840 * 4
    for f in range(8):
        assert L.cs_max_width(f) >= 3840, f  # every technique handles a 4K row in LDS
    p.mode = 99
    oh = ctypes.c_int()
    assert L.cs_output_shape(ctypes.byref(p), ctypes.byref(oh), None, None, None) == _native.CS_EINVAL
    assert b"Unknown mode" in L.cs_last_error()


def test_null_pointers_are_rejected_before_any_device_work():
    L = _native.lib()
    assert L.cs_apply_stereo_divergence(None, None, 1, 8, 8, 1.0, 0.0, 1.0, 0, 0.5, None, None, 0, None) == _native.CS_EINVAL
    assert L.cs_directional_blur(None, 1, 8, 8, 5.0, 6.0, 5.0, 1.0, 0, None, None, None, 0, None) == _native.CS_EINVAL
    with pytest.raises(ValueError):
        from comfystereo_amd import engine
        engine.make_params(1, 8, 8, 8, 8, 3, "none", "sideways", 1, 0, 0, 0.5, 1, False, 0, 6, 1, 0, 1)


def test_too_wide_frames_are_rejected_with_elimit():
    """Frames wider than the LDS-resident row kernels accept fail with CS_ELIMIT before any device work."""
    import ctypes
    L = _native.lib()
    from comfystereo_amd import engine
    wmax = L.cs_max_width(_native.FILL["gpu_warp"])
    p = engine.make_params(1, 8, wmax + 1, 8, wmax + 1, 3, "gpu_warp", "left-right", 4.5, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0,
                           2.0, 6, 12)
    fake = ctypes.c_void_p(16)
    rc = L.cs_generate(ctypes.byref(p), fake, fake, fake, fake, fake, fake, fake, 1 << 40, None)
    assert rc == _native.CS_ELIMIT and b"too wide" in L.cs_last_error()
    p.w = 64
    p.depth_w = 64
    assert L.cs_generate(ctypes.byref(p), fake, fake, fake, fake, fake, fake, fake, 16, None) == _native.CS_EWORKSPACE


def test_max_width_params_is_the_predicate_cs_generate_applies():
    """cs_max_width_mode reports the side-by-side limit for anaglyph polylines, but a call whose halo the tile kernels
    do not take, or that carries the full-D64 flag, is refused between the row kernel's own anaglyph form and that limit.
    cs_max_width_params (ABI 4) searches the SAME predicate cs_generate applies: at its answer the call passes the width gate
    (and fails on the 16-byte workspace), one column beyond it the call is CS_ELIMIT -- for every technique, mode and dialect flag."""
    L = _native.lib()
    from comfystereo_amd import engine
    fake = ctypes.c_void_p(16)
    seen_lower = 0
    for fill in sorted(_native.FILL):
        for mode in ("left-right", "red-cyan-anaglyph", "top-bottom"):
            for flags, div, conv in ((0, 3.0, 0.5), (8, 3.0, 0.5), (24, 3.0, 0.5), (0, 15.0, 1.0), (8, 15.0, 1.0)):
                p = engine.make_params(1, 4, 64, 4, 64, 3, fill, mode, div, 0.0, 0.0, conv, 2.0, False, 20.0, 20.0, 1.0, 0, 12)
                p.flags = flags
                wmax = L.cs_max_width_params(ctypes.byref(p))
                assert 3840 <= wmax <= L.cs_max_width_mode(p.fill, p.mode), (fill, mode, flags)
                seen_lower += wmax < L.cs_max_width_mode(p.fill, p.mode)
                for w, want in ((wmax, (_native.CS_EWORKSPACE, _native.CS_EINVAL)), (wmax + 1, (_native.CS_ELIMIT,))):
                    p.w = p.depth_w = w
                    rc = L.cs_generate(ctypes.byref(p), fake, fake, fake, fake, fake, fake, fake, 16, None)
                    assert rc in want, (fill, mode, flags, div, w, rc, L.cs_last_error())
    assert seen_lower > 0   # (the cases the advisor named exist: a pre-validation with cs_max_width_mode alone would have passed them)
    assert L.cs_max_width_params(None) == 0


def test_debug_switches_are_explicit_and_release_builds_reject_the_phase_cutoffs():
    """cs_debug_set is the only way to reach the development switches (the library never reads the environment); a
    release build refuses the CS_DEBUG_DBG values that would leave outputs unwritten."""
    L = _native.lib()
    assert L.cs_debug_set(_native.DEBUG["dbg"], 17) == _native.CS_OK
    assert L.cs_debug_set(_native.DEBUG["dbg"], 0) == _native.CS_OK
    assert L.cs_debug_set(_native.DEBUG["dbg"], 12) == _native.CS_EINVAL and b"CS_DEV" in L.cs_last_error()
    assert L.cs_debug_set(99, 1) == _native.CS_EINVAL
    src = os.path.join(ROOT, "comfystereo_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(src, f)).read(), f


def test_host_expansion_and_copy_need_no_gpu():
    """cs_host_expand_u8 / cs_host_copy (the compact node boundary's host half, cs_host.hip): plain host code.
    out = codes / 255.0f by true division (np2tensor / convertResult, reference GenerateStereo.py:41-44, 365-378), replicated
    over the channels of a depth map, or the 0 / 1 mask flags (:355-361)."""
    import ctypes
    import numpy as np
    from comfystereo_amd import _native
    L = _native.lib()
    rng = np.random.default_rng(5)
    for count in (0, 1, 7, 4099, 1 << 20):
        codes = rng.integers(0, 256, count, dtype=np.uint8)
        for threads in (1, 3, 0):
            out = np.full(count, -1.0, np.float32)
            assert L.cs_host_expand_u8(codes.ctypes.data, out.ctypes.data, count, 1, 0, threads) == 0 or count == 0
            assert np.array_equal(out.view(np.uint32), (codes.astype(np.float32) / np.float32(255.0)).view(np.uint32))
            out3 = np.full(3 * count, -1.0, np.float32)
            assert L.cs_host_expand_u8(codes.ctypes.data, out3.ctypes.data, count, 3, 0, threads) == 0 or count == 0
            assert np.array_equal(out3.reshape(-1, 3), np.repeat((codes.astype(np.float32) / np.float32(255.0))[:, None], 3, 1))
            m = np.full(count, -1.0, np.float32)
            assert L.cs_host_expand_u8(codes.ctypes.data, m.ctypes.data, count, 1, 1, threads) == 0 or count == 0
            assert np.array_equal(m, (codes != 0).astype(np.float32))
        for off in (1, 2, 3):   # destinations that are not 16-byte aligned (the streaming stores need an aligned body)
            buf = np.full(3 * count + 8, -1.0, np.float32)
            o1 = buf[off:off + count]
            assert L.cs_host_expand_u8(codes.ctypes.data, o1.ctypes.data, count, 1, 0, 3) == 0 or count == 0
            assert np.array_equal(o1, codes.astype(np.float32) / np.float32(255.0)) and buf[off + count] == -1.0 and buf[off - 1] == -1.0
            buf[:] = -1.0
            o3 = buf[off:off + 3 * count]
            assert L.cs_host_expand_u8(codes.ctypes.data, o3.ctypes.data, count, 3, 0, 3) == 0 or count == 0
            assert np.array_equal(o3.reshape(-1, 3), np.repeat((codes.astype(np.float32) / np.float32(255.0))[:, None], 3, 1))
            assert buf[off + 3 * count] == -1.0 and buf[off - 1] == -1.0
        src = rng.integers(0, 256, count * 3 + 5, dtype=np.uint8)
        dst = np.zeros_like(src)
        assert L.cs_host_copy(dst.ctypes.data, src.ctypes.data, src.size, 4) == 0
        assert np.array_equal(dst, src)
    one = np.zeros(4, np.float32)
    assert L.cs_host_expand_u8(None, one.ctypes.data, 4, 1, 0, 1) != 0          # null pointer
    assert L.cs_host_expand_u8(one.ctypes.data, one.ctypes.data, 4, 5, 0, 1) != 0   # replicate out of range
    assert L.cs_host_copy(None, one.ctypes.data, 4, 1) != 0


def test_edge_threshold_equals_the_division():
    """k_gray_edges replaces `clamp(|g| / den, 0, 1) > 0.5` (reference stereoimage_generation.py:1213-1222) by `|g| > t` with
    t = cs_test_edge_threshold(den): for every float in a window of +-64 ulps around t (and a spread of others) the two
    predicates agree in float32 arithmetic; den <= 0, inf and nan report "keep the division" (negative).  No GPU."""
    import numpy as np
    from comfystereo_amd import _native
    L = _native.lib()
    rs = np.random.RandomState(5)
    dens = np.concatenate([np.float32([200.0, 30.0, 5.0, 0.1, 1e-3, 1.0, 255.0, 1e6, 3.3e-5]),
                           np.exp(rs.uniform(-20, 20, 300)).astype(np.float32)])
    for den in dens:
        t = np.float32(L.cs_test_edge_threshold(float(den)))
        assert t > 0
        around = (t.view(np.int32) + np.arange(-64, 65, dtype=np.int32)).view(np.float32)
        g = np.concatenate([around, np.float32([0.0, np.inf, np.nan]), (rs.uniform(0, 4, 64).astype(np.float32) * t)])
        with np.errstate(invalid="ignore", over="ignore"):
            want = np.clip(g / den, np.float32(0), np.float32(1)) > np.float32(0.5)   # (np.clip passes NaN through; NaN > 0.5 is False)
            got = g > t
        assert np.array_equal(got, want), float(den)
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        assert L.cs_test_edge_threshold(bad) < 0


def test_graft_entry_build_checks_the_current_abi_version():
    """__graft_entry__.build() (the driver's "does it build" check) must compare cs_version() with _native.ABI_VERSION, not
    with a literal: round 4 bumped the ABI to 3 and a literal 2 would have failed the build check on a correct tree."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "_native.ABI_VERSION" in src
    assert not re.search(r"cs_version\(\)\s*==\s*\d", src)


def test_width_limits_of_round_6():
    """Host-side width logic (no GPU): the anaglyph modes' limit is the side-by-side modes' for every technique but hybrid_edge_plus
    (an anaglyph beyond the row kernel's stash form runs side by side into scratch + composition: the workspace grows by that
    scratch exactly there), every technique but gpu_warp and hybrid_edge_plus takes 8 192 columns, and the numbers the documentation
    quotes are the library's."""
    L = _native.lib()
    from comfystereo_amd import engine
    sbs, ana = _native.MODE["left-right"], _native.MODE["red-cyan-anaglyph"]
    quoted = {"none": 11578, "naive": 11578, "naive_interpolating": 9536, "polylines_soft": 8414, "polylines_sharp": 8206, "inverse": 9004,
              "hybrid_edge": 9412, "gpu_warp": 7763, "none_post": 11578, "inverse_post": 9004, "hybrid_edge_plus": 6394}
    for fill, want in quoted.items():
        f = _native.FILL[fill]
        assert L.cs_max_width_mode(f, sbs) == want, fill
        if fill != "hybrid_edge_plus":
            assert L.cs_max_width_mode(f, ana) == want, fill
            assert L.cs_max_width(f) == want, fill
        if fill not in ("gpu_warp", "hybrid_edge_plus"):
            assert want >= 8192, fill
    # naive_interpolating: an 8K anaglyph fits the stash form; near the limit an anaglyph asks for the side-by-side scratch (n * h * 2 w * 3 bytes)
    def ws(w, mode):
        p = engine.make_params(2, 16, w, 16, w, 3, "naive_interpolating", mode, 4.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
        return L.cs_workspace_bytes(ctypes.byref(p))
    assert ws(7680, "red-cyan-anaglyph") == ws(7680, "left-right")
    assert ws(9536, "red-cyan-anaglyph") - ws(9536, "left-right") >= 2 * 16 * 2 * 9536 * 3
