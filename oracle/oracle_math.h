/*
 * oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Bit-exact CPU restatements of the two libm routines the reference's arithmetic reaches
 * through numpy / math:
 *
 *   powf  -- `abs(d) ** stereo_offset_exponent` on a numpy float32 scalar
 *            (reference stereoimage_generation.py:1637,1677,1698,1724,1865,1926) is glibc powf.
 *   exp   -- `math.exp(...)` (reference stereoimage_generation.py:1644,1766,1768) is glibc exp.
 *
 * Neither routine's source is under /root/reference: they live in the third-party dependency
 * glibc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11 in the build container), whose powf/exp are the
 * published ARM "optimized-routines" algorithms (Szabolcs Nagy, 2017-2018):
 *   powf: log2 via a 16-entry {1/c, log2 c} table + degree-5 polynomial, exp2 via a 32-entry
 *         table + degree-3 polynomial, everything in double, one final rounding to float.
 *   exp : k = round(x*128/ln2), 128-entry {tail, 2^(k/128)} table + degree-5 polynomial.
 * The table values below were read out of the container's libm.so.6 by
 * tools/extract_libm_tables.py (which also re-derives the exp2 tables from first principles),
 * and the operation order -- in particular WHICH multiply-adds are fused -- follows the
 * x86-64 FMA ifunc variant that runs on the build container's CPU (every a*b+c in the routine
 * is one vfmadd; verified by disassembly).  tests/test_oracle_math.py pins both clones
 * bit-for-bit against the live libm.
 *
 * Compile with -ffp-contract=off: the only fused operations must be the explicit fma() calls.
 */
#ifndef CS_ORACLE_MATH_H
#define CS_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t om_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t om_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double om_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* ---- powf ------------------------------------------------------------------------------- */

static const double om_log2_invc[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0, 0x1.3c995b0b80385p+0,
    0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
    0x1.0953f419900a7p+0, 0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1,
    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1,
};
static const double om_log2_logc[16] = {
    -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,
    -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7af0p-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,
    -0x1.a6f9db6475fcep-5, 0x0.0p+0,              0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
    0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2,
};
static const double om_log2_poly[5] = {
    0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1,
    0x1.71547652ab82bp+0,
};
/* tab[i] = bits(2^(i/32)) - (i << 47) */
static const uint64_t om_exp2f_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
static const double om_exp2f_poly[3] = {
    0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1,
};
#define OM_EXP2F_SHIFT 0x1.8p+47 /* 0x1.8p52 / 32 */

/* 0: y is not an integer, 1: odd integer, 2: even integer (x < 0 handling). */
static inline int om_checkint(uint32_t iy) {
    int e = iy >> 23 & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

static inline float om_powf(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = om_asuint(x), iy = om_asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
        /* x is subnormal, zero, negative, inf or nan; or y is zero, inf or nan */
        if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
            if (2u * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && om_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {
            int yint = om_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 1u << (5 + 11); /* SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11) */
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            /* normalise a subnormal x */
            ix = om_asuint(om_asfloat(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    /* log2(x) in double */
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> (23 - 4)) % 16;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double z = (double)om_asfloat(iz);
    double r = fma(z, om_log2_invc[i], -1.0);
    double y0 = om_log2_logc[i] + (double)k;
    double r2 = r * r;
    double yy = fma(om_log2_poly[0], r, om_log2_poly[1]);
    double p = fma(om_log2_poly[2], r, om_log2_poly[3]);
    double r4 = r2 * r2;
    double q = fma(om_log2_poly[4], r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    double ylogx = (double)y * yy;
    if ((om_asuint64(ylogx) >> 47 & 0xffff) >= (om_asuint64(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) /* overflow */
            return sign_bias ? -INFINITY : INFINITY;
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
        /* (-150,-126]: falls through; the double->float rounding below yields the subnormal */
    }
    /* 2^ylogx in double, one rounding to float */
    double kd = ylogx + OM_EXP2F_SHIFT;
    uint64_t ki = om_asuint64(kd);
    kd -= OM_EXP2F_SHIFT;
    double rr = ylogx - kd;
    uint64_t t = om_exp2f_tab[ki % 32];
    t += (ki + sign_bias) << (52 - 5);
    double s = om_asdouble(t);
    double zz = fma(om_exp2f_poly[0], rr, om_exp2f_poly[1]);
    double rr2 = rr * rr;
    double ye = fma(om_exp2f_poly[2], rr, 1.0);
    ye = fma(zz, rr2, ye);
    ye = ye * s;
    return (float)ye;
}

/* ---- exp (double) ------------------------------------------------------------------------- */

#include "libm_exp_table.h"

/* Bit-exact for 2^-54 <= |x| < 512 and for |x| < 2^-54 (the only ranges the path reaches:
 * arguments are -(diff^2)/2 with |diff| < 2.5, -dsq/2 and -(dg^2)/200 with |dg| <= 255);
 * outside them the published special-case handling is followed but not pinned. */
static inline double om_exp(double x) {
    uint32_t abstop = (uint32_t)(om_asuint64(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x; /* |x| < 2^-54 */
        if (abstop >= 0x409u) {                              /* |x| >= 1024, inf, nan */
            if (om_asuint64(x) == om_asuint64(-INFINITY)) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (om_asuint64(x) >> 63) ? 0.0 : INFINITY;
        }
        abstop = 0; /* 512 <= |x| < 1024 */
    }
    double kd = fma(x, OM_EXP_INVLN2N, OM_EXP_SHIFT);
    uint64_t ki = om_asuint64(kd);
    kd -= OM_EXP_SHIFT;
    double r = fma(kd, OM_EXP_NEGLN2HIN, x);
    r = fma(kd, OM_EXP_NEGLN2LON, r);
    uint64_t idx = 2 * (ki % 128);
    uint64_t top = ki << (52 - 7);
    double tail = om_asdouble(om_exp_tab[idx]);
    uint64_t sbits = om_exp_tab[idx + 1] + top;
    double r2 = r * r;
    double a = fma(r, OM_EXP_C3, OM_EXP_C2);
    double tr = tail + r;
    double b = fma(r, OM_EXP_C5, OM_EXP_C4);
    double lo = fma(a, r2, tr);
    double r4 = r2 * r2;
    double tmp = fma(r4, b, lo);
    if (abstop == 0) {
        double scale, y;
        if ((ki & 0x80000000u) == 0) {
            sbits -= 1009ull << 52;
            scale = om_asdouble(sbits);
            return 0x1p1009 * (scale + scale * tmp);
        }
        sbits += 1022ull << 52;
        scale = om_asdouble(sbits);
        y = scale + scale * tmp;
        if (y < 1.0) {
            double hi, l2;
            l2 = scale - y + scale * tmp;
            hi = 1.0 + y;
            l2 = 1.0 - hi + y + l2;
            y = (hi + l2) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    double scale = om_asdouble(sbits);
    return fma(scale, tmp, scale);
}

#endif
