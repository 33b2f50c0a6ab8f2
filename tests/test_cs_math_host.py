"""comfystereo_amd/csrc/cs_math.h (the routines the HIP kernels use) compiled for the HOST and pinned against the
live libm -- runs without a GPU.  The device build of the same header is checked in tests/test_gpu_parity.py."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <stdio.h>
#include <math.h>
#include <initializer_list>
#include "cs_math.h"
int main() {
    static const csm::PowfTables T = CS_POWF_TABLES_INIT;
    long bad = 0, n = 0;
    for (int e = 1; e <= 20; e++) {
        float y = (float)(e / 10.0);
        for (uint32_t u = 0; u <= 0x3f800000u; u += 4099) {
            float x = csm::u2f(u);
            if (csm::f2u(powf(x, y)) != csm::f2u(csm::powf_exact(x, y, &T))) bad++;
            n++;
        }
    }
    for (uint32_t u = 0; u <= 0x40a00000u; u += 257) {
        float d = csm::u2f(u); float a = -(d * d) / 2.0f;
        if (csm::d2u(exp((double)a)) != csm::d2u(csm::exp_exact((double)a, cs_exp_tab))) bad++;
        // the branch-free form the hybrid_edge kernels use for their Gaussian weights (0, denormal and tiny arguments included)
        if (csm::d2u(exp((double)a)) != csm::d2u(csm::exp_exact_small((double)a, cs_exp_tab))) bad++;
        n++;
    }
    for (int i = 0; i < 400000; i++) {   // (-(diff^2) / 200 of edge_aware_gap_fill: doubles in [-330, 0], and tiny ones)
        double x = -330.0 * (double)i / 400000.0, t = -ldexp(1.0 + i * 1e-6, -40 - (i % 1000));
        if (csm::d2u(exp(x)) != csm::d2u(csm::exp_exact_small(x, cs_exp_tab))) bad++;
        if (csm::d2u(exp(t)) != csm::d2u(csm::exp_exact_small(t, cs_exp_tab))) bad++;
        n++;
    }
    for (int i = 0; i < 2000000; i++) {
        double x = -330.0 * (double)i / 2000000.0;
        if (csm::d2u(exp(x)) != csm::d2u(csm::exp_exact(x, cs_exp_tab))) bad++;
        n++;
    }
    // the two spatial weights of edge_aware_gap_fill hard-coded in cs_rowwarp.hip (CS_EXP_M05, CS_EXP_M10)
    if (csm::d2u(csm::exp_exact(-0.5, cs_exp_tab)) != csm::d2u(0x1.368b2fc6f960ap-1) || csm::d2u(exp(-0.5)) != csm::d2u(0x1.368b2fc6f960ap-1)) bad++;
    if (csm::d2u(csm::exp_exact(-1.0, cs_exp_tab)) != csm::d2u(0x1.78b56362cef38p-2) || csm::d2u(exp(-1.0)) != csm::d2u(0x1.78b56362cef38p-2)) bad++;
    // exponent shortcuts of the tile kernel: two full binades + a strided sweep of all finite x >= 0
    long risky_n = 0;
    auto check_sq = [&](uint32_t u) {
        float x = csm::u2f(u);
        bool risky;
        float sq = csm::square_or_flag(x, risky);
        if (risky) risky_n++;
        else if (csm::f2u(powf(x, 2.0f)) != csm::f2u(sq)) bad++;
        if (csm::f2u(powf(x, 1.0f)) != u) bad++;
        n++;
    };
    for (uint32_t u = 0x3f000000u; u < 0x3f800000u; u++) check_sq(u);
    if (risky_n > (1 << 23) / 200) bad++;  // the shortcut must stay a shortcut (0.4 % of a binade is flagged)
    for (uint32_t u = 0x35800000u; u < 0x36000000u; u++) check_sq(u);
    for (uint32_t u = 0; u < 0x7f800000u; u += 1021) check_sq(u);
    // closed forms of the per-pixel float64 constants used by the tile kernel's fast path (cs_polytile.hip, fast_px)
    for (int col = 0; col < (1 << 24); col++) {
        double from_d = (double)col + 1e-7, to_d = (double)(col + 1) - 1e-7, sig = to_d - from_d;
        float ff = col == 0 ? 0x1.ad7f2ap-24f : (col == 1 ? 0x1.000002p+0f : (float)col);
        float tf = col == 0 ? 0x1.fffffcp-1f : (col == 1 ? 0x1.fffffep+0f : (float)(col + 1));
        if ((float)from_d != ff || (float)to_d != tf || (float)(from_d + 0.5 * sig) != (float)col + 0.5f ||
            (float)sig != 0x1.fffffap-1f || sig == 0.0) bad++;
        n++;
    }
    for (int k = 0; k < 256; k++) { if (csm::f2u(csm::code_over_255((float)k)) != csm::f2u((float)k / 255.0f)) bad++; n++; }
    float q[] = {65025.f, 300.7f, -3.2f, 255.9f, 256.f, 1e10f};
    int want[] = {1, 44, 253, 255, 0, 0};
    for (int i = 0; i < 6; i++) if (csm::f32_to_u8_wrap(q[i]) != want[i]) bad++;
    for (float f : {-3.5f, -0.0f, 0.0f, 1e-30f, 2.0f}) if (csm::ord2f(csm::f2ord(f)) != f) bad++;
    if (!(csm::f2ord(-1.0f) < csm::f2ord(-0.5f) && csm::f2ord(-0.5f) < csm::f2ord(0.25f))) bad++;
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
'''


@pytest.mark.skipif(platform.libc_ver()[1] != "2.35" or platform.machine() != "x86_64",
                    reason="pinned to glibc 2.35 x86-64 (FMA ifunc variant)")
def test_device_math_header_matches_libm_on_host(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", *fma, "-I", os.path.join(ROOT, "comfystereo_amd", "csrc"),
                           str(src), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    n, bad = map(int, out.stdout.split())
    assert out.returncode == 0 and bad == 0 and n > 40_000_000
