// cs_math.h -- libm-exact scalar math for the gfx950 kernels (also compiles for the host so the
// CPU test-suite can pin it against the live libm without a GPU).
//
// The reference's float32 disparity is `abs(d) ** exponent` on a numpy scalar, i.e. glibc powf
// (reference stereoimage_generation.py:1637,1677,1698,1724,1865,1926) and its hybrid_edge weights
// are `math.exp`, i.e. glibc exp (reference :1644,1766,1768).  Integer pixel indices are derived
// from those values, so the kernels need the SAME bits, not "an accurate pow".  glibc 2.35's
// routines are the published ARM optimized-routines algorithms; on an FMA-capable x86-64 the ifunc
// variant fuses every a*b+c.  The functions below restate that variant operation by operation
// with explicit fma(); the translation unit must be compiled with -ffp-contract=off so nothing
// else is fused.  gfx950 executes the f64 ops natively (v_fma_f64) -- about 12 per powf.
//
// Table parameters: cs_powf tables are template-loaded from LDS on device (`Tab`), see PowfTables.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CS_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define CS_HD static inline
#endif

#include "cs_exp_table.h"

namespace csm {

CS_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
CS_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
CS_HD uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
CS_HD double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

// log2 table: for i in 0..15, c near the centre of [0x1.66p-1*2^(i/16)...): {1/c, log2(c)}.
// exp2 table: bits(2^(i/32)) - (i << 47).  Values as in glibc 2.35 __powf_log2_data/__exp2f_data.
struct PowfTables {
    double invc[16];
    double logc[16];
    uint64_t exp2t[32];
};

#define CS_POWF_TABLES_INIT                                                                                      \
    {                                                                                                            \
        {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0, 0x1.3c995b0b80385p+0,                 \
         0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,                 \
         0x1.0953f419900a7p+0, 0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1,                 \
         0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1},                \
        {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,             \
         -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7af0p-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,             \
         -0x1.a6f9db6475fcep-5, 0x0.0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3, 0x1.e840b4ac4e4d2p-3,      \
         0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2},                                      \
        {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,             \
         0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,             \
         0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,             \
         0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,             \
         0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,             \
         0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,             \
         0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,             \
         0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull}             \
    }

// powf for x >= 0 (the path only ever raises |d|), any finite y.  Negative x is mapped through
// |x| (callers never pass it).  `T` points at a PowfTables (LDS copy on device, static on host).
CS_HD float powf_exact(float x, float y, const PowfTables* T) {
    uint32_t ix = f2u(x) & 0x7fffffffu, iy = f2u(y);
    if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {  // y is 0, inf or nan
        if (2u * iy == 0) return 1.0f;
        if (ix == 0x3f800000u) return 1.0f;
        if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
        if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
        return y * y;
    }
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x is 0, subnormal, inf or nan
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {
            float x2 = x * x;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        ix = f2u(u2f(ix) * 0x1p23f) & 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2(x): x = 2^k * z, z in [0x1.66p-1, 0x1.66p0); r = z/c - 1 with c from the table row of z
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    double z = (double)u2f(ix - top);
    double k = (double)((int32_t)top >> 23);
    double r = fma(z, T->invc[i], -1.0);
    double y0 = T->logc[i] + k;
    double r2 = r * r;
    double hi = fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    double mid = fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    double r4 = r2 * r2;
    double q = fma(0x1.71547652ab82bp+0, r, y0);
    q = fma(mid, r2, q);
    double ylogx = (double)y * fma(hi, r4, q);
    if (((d2u(ylogx) >> 47) & 0xffff) >= (d2u(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return u2f(0x7f800000u);
        if (ylogx <= -150.0) return 0.0f;
    }
    // 2^ylogx: split off n/32, polynomial on the remainder, scale by the table entry
    const double SH = 0x1.8p+47;
    double kd = ylogx + SH;
    uint64_t ki = d2u(kd);
    kd -= SH;
    double rr = ylogx - kd;
    double s = u2d(T->exp2t[ki & 31u] + (ki << 47));
    double c01 = fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    double rr2 = rr * rr;
    double c2 = fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    double res = fma(c01, rr2, c2) * s;
    return (float)res;
}

// Same function, shaped for SIMT: the main path runs unconditionally on a sanitised operand (no divergent
// branches: they cost scalar-unit instructions on every wave), and lanes that need the special cases (x zero /
// subnormal / inf / nan, |y*log2 x| >= 126, y zero / inf / nan) are redone by powf_exact() behind ONE
// wave-uniform branch.  `any_special` tells whether any lane of the wave took that path (device: __any).
#if defined(__HIPCC__)
__device__ __forceinline__ float powf_exact_simt(float x, float y, const PowfTables* T) {
    uint32_t ix = f2u(x) & 0x7fffffffu;
    const uint32_t iy = f2u(y);
    bool special = (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) || (2u * iy - 1u >= 2u * 0x7f800000u - 1u);
    ix = special ? 0x3f800000u : ix;
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    double z = (double)u2f(ix - top);
    double k = (double)((int32_t)top >> 23);
    double r = fma(z, T->invc[i], -1.0);
    double y0 = T->logc[i] + k;
    double r2 = r * r;
    double hi = fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    double mid = fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    double r4 = r2 * r2;
    double q = fma(0x1.71547652ab82bp+0, r, y0);
    q = fma(mid, r2, q);
    double ylogx = (double)y * fma(hi, r4, q);
    special = special || (((d2u(ylogx) >> 47) & 0xffff) >= (d2u(126.0) >> 47));
    const double SH = 0x1.8p+47;
    double kd = ylogx + SH;
    uint64_t ki = d2u(kd);
    kd -= SH;
    double rr = ylogx - kd;
    double s = u2d(T->exp2t[ki & 31u] + (ki << 47));
    double c01 = fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    double rr2 = rr * rr;
    double c2 = fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    float res = (float)(fma(c01, rr2, c2) * s);
    if (__any(special)) {
        if (special) res = powf_exact(x, y, T);
    }
    return res;
}
#endif

// Shortcuts of powf for the two exponents the node is used with most (the widget default 2.0, and 1.0):
//  * powf(x, 1) == x for every finite x >= 0;
//  * powf(x, 2) == x * x (the correctly rounded product) unless the exact product lies within 2^-9 ulp of a rounding
//    midpoint: glibc's powf has a relative error of about 2^-35, and in an exhaustive sweep of the clone over all
//    finite x >= 0 the largest distance from the midpoint at which the two differ is 0.0017 ulp (888704 units of
//    2^-29 ulp; the margin used here is 2^20 units).  Outside 2^-60 <= x <= 2^60 (subnormal / overflowing products)
//    the argument counts as risky, x == 0 is exact.  Risky arguments (0.4 %) must take the full routine.
// tests: tests/test_oracle_math.py (criterion vs the oracle's clone), tests/test_gpu_parity.py (full binade on the GPU)
CS_HD float square_or_flag(float ax, bool& risky) {
    const double p2 = (double)ax * (double)ax;  // exact: 24 + 24 bits
    int low = (int)(uint32_t)(d2u(p2) & 0x1fffffffull) - 0x10000000;
    low = low < 0 ? -low : low;
    risky = (low < (1 << 20) || ax < 0x1p-60f || ax > 0x1p60f) && ax != 0.0f;
    return ax * ax;
}

// exp for |x| < 512 (pinned); larger magnitudes saturate to 0 / inf without the libm corner cases.
// `tab` is the 256-entry {tail, scale bits} table (cs_exp_tab, or an LDS copy).
CS_HD double exp_exact(double x, const unsigned long long* tab) {
    uint32_t abstop = (uint32_t)(d2u(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
        if (abstop >= 0x7ffu) return (d2u(x) == 0xfff0000000000000ull) ? 0.0 : 1.0 + x;
        return (d2u(x) >> 63) ? 0.0 : u2d(0x7ff0000000000000ull);
    }
    double kd = fma(x, CS_EXP_INVLN2N, CS_EXP_SHIFT);
    uint64_t ki = d2u(kd);
    kd -= CS_EXP_SHIFT;
    double r = fma(kd, CS_EXP_NEGLN2HIN, x);
    r = fma(kd, CS_EXP_NEGLN2LON, r);
    uint32_t idx = 2u * (uint32_t)(ki & 127u);
    double tail = u2d(tab[idx]);
    double scale = u2d(tab[idx + 1] + (ki << 45));
    double r2 = r * r;
    double p23 = fma(r, CS_EXP_C3, CS_EXP_C2);
    double p45 = fma(r, CS_EXP_C5, CS_EXP_C4);
    double lo = fma(p23, r2, tail + r);
    double t = fma(r2 * r2, p45, lo);
    return fma(scale, t, scale);
}

// The same for a caller that KNOWS |x| < 512 and x finite (the Gaussian weights of hybrid_edge: -8 < x <= 0): the range test and
// its branches are gone.  Identical results: the main path maps x = 0 to exactly 1 (r = 0, table entry 0 = {0, 1.0}) and a tiny
// |x| < 2^-54 to fl(1 + x), which is what libm's early return computes.
CS_HD double exp_exact_small(double x, const unsigned long long* tab) {
    double kd = fma(x, CS_EXP_INVLN2N, CS_EXP_SHIFT);
    uint64_t ki = d2u(kd);
    kd -= CS_EXP_SHIFT;
    double r = fma(kd, CS_EXP_NEGLN2HIN, x);
    r = fma(kd, CS_EXP_NEGLN2LON, r);
    uint32_t idx = 2u * (uint32_t)(ki & 127u);
    double tail = u2d(tab[idx]);
    double scale = u2d(tab[idx + 1] + (ki << 45));
    double r2 = r * r;
    double p23 = fma(r, CS_EXP_C3, CS_EXP_C2);
    double p45 = fma(r, CS_EXP_C5, CS_EXP_C4);
    double lo = fma(p23, r2, tail + r);
    double t = fma(r2 * r2, p45, lo);
    return fma(scale, t, scale);
}

// numpy float32 -> uint8 astype on x86-64 (cvttss2si, low byte): out-of-range -> 0x80000000 -> 0.
CS_HD uint8_t f32_to_u8_wrap(float v) {
    int32_t i = (v > -2147483904.0f && v < 2147483648.0f) ? (int32_t)v : (int32_t)0x80000000;
    return (uint8_t)(uint32_t)i;
}

// order-preserving float <-> uint32 map for integer atomics min/max
CS_HD uint32_t f2ord(float f) { uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
CS_HD float ord2f(uint32_t o) { return u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// k / 255 for an integer-valued 0 <= k <= 255 (convertResult / np2tensor, reference GenerateStereo.py:41-44): quotient by the
// rounded reciprocal plus one residual correction -- equal to the IEEE division for all 256 codes
// (tests/test_cs_math_host.py); three 2-cycle instructions on gfx950 instead of a table read with bank conflicts.
CS_HD float code_over_255(float k) {
    const float rcp = 1.0f / 255.0f;
    const float q = k * rcp;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q, k), rcp, q);
}

}  // namespace csm
