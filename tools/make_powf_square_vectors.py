#!/usr/bin/env python3
"""Known-answer vectors for the exponent-2 shortcut of the disparity power (cs_polytile.hip, `square`).

glibc's powf(x, 2) differs from the correctly rounded x*x for 6061 mantissas per binade (the exact product lies within
0.0017 ulp of a rounding midpoint and powf's 2^-35-ish relative error tips it the other way).  This script sweeps one
binade of the oracle's powf clone (oracle/oracle_math.h, bit-exact with the libm the reference runs on), collects
arguments where the two differ -- the inputs the shortcut must route through the full routine -- and stores them with
powf's answers in tests/golden/powf_square.npz.

  python tools/make_powf_square_vectors.py      (needs gcc; ~2 s)
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <stdint.h>
#include <string.h>
#include "oracle_math.h"
int sweep(uint32_t lo, uint32_t hi, uint32_t stride, float* xs, float* ys, int cap) {
    int n = 0;
    for (uint32_t u = lo; u < hi && n < cap; u += stride) {
        float x; memcpy(&x, &u, 4);
        float r = om_powf(x, 2.0f), f = x * x;
        if (memcmp(&r, &f, 4)) { xs[n] = x; ys[n] = r; n++; }
    }
    return n;
}
'''


def main():
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        so = os.path.join(td, "s.so")
        open(c, "w").write(SRC)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "oracle"), c, "-o", so, "-lm"])
        lib = ctypes.CDLL(so)
        xs_all, ys_all = [], []
        # binades [2^-1, 1), [2^-2, 2^-1), [2^-4, 2^-3), [2^-8, 2^-7): 64 vectors each
        for e in (126, 125, 123, 119):
            xs = np.zeros(64, np.float32)
            ys = np.zeros(64, np.float32)
            n = lib.sweep(ctypes.c_uint32(e << 23), ctypes.c_uint32((e + 1) << 23), ctypes.c_uint32(1),
                          xs.ctypes.data_as(ctypes.c_void_p), ys.ctypes.data_as(ctypes.c_void_p), 64)
            assert n == 64
            xs_all.append(xs)
            ys_all.append(ys)
        x = np.concatenate(xs_all)
        y = np.concatenate(ys_all)
        assert np.all(x * x != y)
        out = os.path.join(ROOT, "tests", "golden", "powf_square.npz")
        np.savez(out, x=x, powf_x_2=y)
        print("wrote", out, x.shape)


if __name__ == "__main__":
    main()
