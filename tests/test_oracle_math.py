"""The libm clones of the oracle, bit-for-bit against the live libm (glibc 2.35 FMA variant) -- not gpu."""
import ctypes
import ctypes.util
import platform

import numpy as np
import pytest

from oracle import oracle

libm = ctypes.CDLL(ctypes.util.find_library("m"))
libm.powf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
libm.exp.restype = ctypes.c_double
libm.exp.argtypes = [ctypes.c_double]

pin = pytest.mark.skipif(platform.libc_ver()[1] != "2.35" or platform.machine() != "x86_64",
                         reason="the clones are pinned to glibc 2.35 x86-64 (FMA variant)")


@pin
def test_powf_clone_matches_libm():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(40000, dtype=np.float32), np.float32([0.0, 1.0, 0.5, 0.25, 1e-30, 1e-40, 3.5])])
    for e in [i / 10 for i in range(1, 21)] + [2.7, 3.0]:
        y = np.float32(e)
        for x in xs[:: 1 if e in (2.0, 1.0, 0.5) else 8]:
            a, b = libm.powf(float(x), float(y)), L.oracle_powf(float(x), float(y))
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (x, y)


@pin
def test_exp_clone_matches_libm():
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([-rng.random(20000) * 330.0, -(rng.random(20000, dtype=np.float32).astype(np.float64) ** 2) * 2,
                         [0.0, -0.0, -0.5, -1.0, -1e-300]])
    for x in xs:
        a, b = libm.exp(float(x)), L.oracle_exp(float(x))
        assert np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64), x


def test_powf_known_answers():
    """Stored known answers (glibc 2.35): powf(x, 2) is NOT x*x in general."""
    L = oracle.lib()
    assert L.oracle_powf(0.0, 2.0) == 0.0 and L.oracle_powf(1.0, 1.3) == 1.0 and L.oracle_powf(0.25, 0.5) == 0.5
    assert L.oracle_powf(0.5, 2.0) == 0.25


def test_powf_square_vectors():
    """tests/golden/powf_square.npz (tools/make_powf_square_vectors.py): arguments where powf(x, 2) != x * x."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "powf_square.npz"))
    L = oracle.lib()
    got = np.array([L.oracle_powf(float(v), 2.0) for v in g["x"]], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), g["powf_x_2"].view(np.uint32))
    assert np.all(g["x"] * g["x"] != g["powf_x_2"])
