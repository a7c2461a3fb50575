// jsmath_device.hpp — Math.log10 / Math.pow as the reference's JavaScript engine evaluates them,
// for device code.  The reference's noise gate truncates their results
// (`parseInt(Math.pow(10, Math.log10(y) - 3) / 20)`, ref dist/main.js:2 @B28615), so the last bit
// decides segment boundaries at exact multiples of 20000 / 2000 / 200.  V8 evaluates both with
// its port of Sun's fdlibm (e_log.c, e_log10.c, e_pow.c; pow with V8's one deviation marked below)
// — pure IEEE double arithmetic, reproduced here operation for operation.  The translation unit
// is compiled with -ffp-contract=off.  Checked bit-for-bit against Node on the GPU tests.
//
// The fdlibm algorithms carry this notice:
// ====================================================
// Copyright (C) 1993-2004 by Sun Microsystems, Inc. All rights reserved.
// Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice
// is preserved.
// ====================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wsa { namespace jsm {

__device__ __forceinline__ int32_t hiw(double x) { return __double2hiint(x); }
__device__ __forceinline__ uint32_t low(double x) { return (uint32_t)__double2loint(x); }
__device__ __forceinline__ double mk(int32_t hi, uint32_t lo) { return __hiloint2double(hi, (int32_t)lo); }
__device__ __forceinline__ double set_hi(double x, int32_t hi) { return mk(hi, low(x)); }
__device__ __forceinline__ double clr_lo(double x) { return mk(hiw(x), 0u); }

__device__ inline double log_e(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        two54 = 1.80143985094819840000e+16,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    int32_t hx = hiw(x); uint32_t lx = low(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k -= 54; x *= two54; hx = hiw(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = set_hi(x, hx | (i ^ 0x3ff00000));
    k += (i >> 20);
    const double f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) { if (k == 0) return 0.0; const double dk = (double)k; return dk * ln2_hi + dk * ln2_lo; }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        const double dk = (double)k; return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    i = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

__device__ inline double log10(double x) {
    const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
        log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    int32_t hx = hiw(x); const uint32_t lx = low(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k -= 54; x *= two54; hx = hiw(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = (double)(k + i);
    x = set_hi(x, hx);
    const double z = y * log10_2lo + ivln10 * log_e(x);
    return z + y * log10_2hi;
}

// log10 for POSITIVE, FINITE, NORMAL x (the feature reductions' band energies: fp32 values above zero) — the same IEEE operations on the same operands as
// log10 / log_e above, arranged for a wave whose lanes all hold different arguments.  There every early return and every `k == 0` / `i > 0` arm runs
// one after the other under exec masks; here
//   * the zero / negative / subnormal / infinite / NaN arms are gone (cannot occur);
//   * log_e's `k == 0` arms are the general arms evaluated with dk = 0: 0 * ln2_hi - ((X - 0 * ln2_lo) - f) = -(X - f) = f - X, and s * (...) + 0 is s * (...)
//     — bit for bit (IEEE subtraction is antisymmetric; the one sign-of-zero case, X == f, gives +0 either way);
//   * the two forms of the tail (`i > 0`: with hfsq) are both evaluated and selected — what divergent lanes did anyway;
//   * only |f| < 2^-20 (x within 2^-20 of a power of two — exact powers of two do occur) stays a branch.
// tests/test_gpu_units.py holds it against jsm::log10 and the V8 vectors.
__device__ __forceinline__ double log10_fin(double x) {
    const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    // ---- log10: x = 2^k10 * m, m in [1, 2) (k10 >= 0) or [0.5, 1) (k10 < 0)
    int32_t hx = hiw(x);
    const int32_t k10 = (hx >> 20) - 1023;
    const int32_t i10 = (int32_t)((uint32_t)k10 >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i10) << 20);
    const double y = (double)(k10 + i10);
    // ---- log_e of that m
    int32_t k = (hx >> 20) - 1023;                    // 0 or -1
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    const double xm = mk(hx | (i ^ 0x3ff00000), low(x));
    k += (i >> 20);
    const double f = xm - 1.0;
    const double dk = (double)k;
    double le;
    if (__builtin_expect((0x000fffff & (2 + hx)) < 3, 0)) {
        if (f == 0.0) le = k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        else {
            const double R = f * f * (0.5 - 0.33333333333333333 * f);
            le = k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
        }
    } else {
        const double s = f / (2.0 + f);
        const double z = s * s;
        i = hx - 0x6147a;
        const double w = z * z;
        const int32_t j = 0x6b851 - hx;
        const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
        const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
        i |= j;
        const double R = t2 + t1;
        const double hfsq = 0.5 * f * f;
        const double khi = dk * ln2_hi, klo = dk * ln2_lo;
        const double a1 = khi - ((hfsq - (s * (hfsq + R) + klo)) - f);      // i > 0
        const double a2 = khi - ((s * (f - R) - klo) - f);
        le = i > 0 ? a1 : a2;
    }
    const double zz = y * log10_2lo + ivln10 * le;
    return zz + y * log10_2hi;
}

// x ** y for finite x > 0 and finite y with |y| < 2^31 — every call site in the hot path
// (10 ** (t - 3), 10 ** (t - 2), 10 ** (t / 3), 10 ** (dB / 20)).  Results that would be
// subnormal do not occur for these arguments.
__device__ inline double pow_pos(double x, double y) {
    const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01},
        dp_l[2] = {0.0, 1.35003920212974897128e-08};
    const double two53 = 9007199254740992.0,
        L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01,
        L3 = 3.33333329818377432918e-01, L4 = 2.72728123808534006489e-01,
        L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
        P5 = 4.13813679705723846039e-08,
        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01,
        lg2_l = -1.90465429995776804525e-09,
        cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
        cp_l = -7.02846165095275826516e-09;
    const int32_t hy = hiw(y); const uint32_t ly = low(y);
    const int32_t iy = hy & 0x7fffffff;
    if ((iy | ly) == 0) return 1.0;
    if (ly == 0) {
        if (iy == 0x3ff00000) return hy < 0 ? 1.0 / x : x;
        if (hy == 0x40000000) return x * x;
        if (hy == 0x3fe00000) return __dsqrt_rn(x);
    }
    double ax = x;
    int32_t ix = hiw(x);
    if (low(x) == 0 && ix == 0x3ff00000) return 1.0;                 // x == 1
    int32_t n = 0;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = hiw(ax); }
    n += (ix >> 20) - 0x3ff;
    int32_t j = ix & 0x000fffff;
    int32_t k;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = set_hi(ax, ix);
    double u = ax - bp[k];
    double v = 1.0 / (ax + bp[k]);
    const double ss = u * v;
    const double s_h = clr_lo(ss);
    double t_h = mk(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0u);
    double t_l = ax - (t_h - bp[k]);
    const double s_l = v * ((u - s_h * t_h) - s_h * t_l);
    double s2 = ss * ss;
    double r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = clr_lo(3.0 + s2 + r);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    double p_h = clr_lo(u + v);
    double p_l = v - (p_h - u);
    const double z_h = cp_h * p_h;
    const double z_l = cp_l * p_h + p_l * cp + dp_l[k];
    double t = (double)n;
    double t1 = clr_lo(((z_h + z_l) + dp_h[k]) + t);
    const double t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    const double y1 = clr_lo(y);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    double z = p_l + p_h;
    j = hiw(z);
    int32_t i = j & 0x7fffffff;
    if (j >= 0x40900000) return __builtin_inf();                      // overflow (not reachable from the hot path)
    if (i >= 0x4090cc00) return 0.0;                                  // underflow
    k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        t = mk(n & ~(0x000fffff >> k), 0u);
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = clr_lo(p_l + p_h);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    const double w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    // V8's port folds the correction term into the divisor (fdlibm: (z*t1)/(t1-2) - (w+z*w)).
    r = (z * t1) / ((t1 - 2.0) - (w + z * w));
    z = 1.0 - (r - z);
    j = hiw(z);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0) return 0.0;                                   // subnormal result: not reachable
    return set_hi(z, j);
}

}}  // namespace wsa::jsm
