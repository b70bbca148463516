// ssme_math.h -- Philox4x32-10 and libm-free fp64 elementary functions for gfx950.
//
// Every function is a fixed sequence of IEEE-754 binary64 operations (+ - * fma, correctly
// rounded sqrt and division) plus integer bit moves, so the same source gives the same bits
// in device code and in this library's host code (derived model constants).  The operation
// order is the specification; build with -ffp-contract=off.
//   exp   : Cody-Waite reduction by ln2 (hi/lo), Taylor degree 13 on |r| <= ln2/2 (fma Horner)
//   log   : frexp to [sqrt(1/2), sqrt(2)), s = f/(2+f), fdlibm's degree-7 even polynomial as fma Horner chains
//   sincos: exact octant reduction of 4u, then fdlibm's sine/cosine kernel polynomials on r*pi/2 as fma Horner chains
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssme_log_table.h"

#define SSME_HD __host__ __device__ __forceinline__

namespace ssme {

SSME_HD double bits2d(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; __builtin_memcpy(&d, &u, 8); return d;
#endif
}
SSME_HD uint64_t d2bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; __builtin_memcpy(&u, &d, 8); return u;
#endif
}
SSME_HD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// fma(a, b, C) with a compile-time constant addend.  hipcc lowers this Horner step to v_fmac_f64 plus two
// v_mov_b32 that re-materialise C in the accumulator VGPRs every time; v_fma_f64 takes C from an SGPR pair.
SSME_HD double dfma_c(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}
SSME_HD double dinf() { return bits2d(0x7ff0000000000000ull); }
SSME_HD double dnan() { return bits2d(0x7ff8000000000000ull); }
SSME_HD double pow2i(int n) { return bits2d((uint64_t)(n + 1023) << 52); }

// ---- Philox4x32-10 -------------------------------------------------------------------
struct u32x4 { uint32_t v0, v1, v2, v3; };

SSME_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;     // one v_mad_u64_u32 each
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    }
    return u32x4{c0, c1, c2, c3};
}

// 52 random bits (the top 52 of the 64-bit word a:b) as the mantissa of a double in [1,2)
SSME_HD double u12(uint32_t a, uint32_t b) { return bits2d(0x3ff0000000000000ull | ((((uint64_t)a << 32) | b) >> 12)); }
SSME_HD double u01_co(uint32_t a, uint32_t b) { return u12(a, b) - 1.0; }    // [0,1)   multiples of 2^-52
SSME_HD double u01_oc(uint32_t a, uint32_t b) { return 2.0 - u12(a, b); }    // (0,1]

enum { STREAM_PROP = 0, STREAM_RESAMP = 1, STREAM_RESAMP_EXTRA = 2, STREAM_GAMMA = 16 };

// ---- exp -----------------------------------------------------------------------------
SSME_HD double dmaxnum(double a, double b) { return __builtin_fmax(a, b); }   // IEEE maxNum: NaN-squashing
SSME_HD double dminnum(double a, double b) { return __builtin_fmin(a, b); }
SSME_HD double dldexp(double x, int e) { return __builtin_ldexp(x, e); }

// exp(x) * 2^sc.  The clamp squashes NaN to the lower bound (result 0); callers that must keep a
// NaN carry it through another operand.
SSME_HD double dexp_scaled(double x, int sc) {
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double SH = 6755399441055744.0;  // 1.5 * 2^52
    const double xc = dminnum(dmaxnum(x, -746.0), 710.0);
    const double kf = dfma_c(xc, LOG2E, SH) - SH;
    const int k = (int)kf;
    double r = dfma(-kf, LN2_HI, xc);
    r = dfma(-kf, LN2_LO, r);
    double q = 1.6059043836821613e-10;
    q = dfma_c(q, r, 2.08767569878681e-09);
    q = dfma_c(q, r, 2.505210838544172e-08);
    q = dfma_c(q, r, 2.755731922398589e-07);
    q = dfma_c(q, r, 2.7557319223985893e-06);
    q = dfma_c(q, r, 2.48015873015873e-05);
    q = dfma_c(q, r, 0.0001984126984126984);
    q = dfma_c(q, r, 0.001388888888888889);
    q = dfma_c(q, r, 0.008333333333333333);
    q = dfma_c(q, r, 0.041666666666666664);
    q = dfma_c(q, r, 0.16666666666666666);
    q = dfma_c(q, r, 0.5);
    const double e = dfma(r * r, q, r);
    const double p = 1.0 + e;
    return dldexp(p, k + sc);
}
SSME_HD double dexp(double x) { return dexp_scaled(x, 0); }

// The bootstrap filter's exp (its hot loop spends a third of its fp64 instructions here: logG's exp(-x) and the weight
// quantisation, per particle): exp(x) = 2^k 2^(j/64) e^r with n = rint(x 64/ln2) = 64 k + j, 2^(j/64) from a
// 64-entry double-double table and a degree-6 series for e^r, |r| <= ln2/128 (truncation < 2^-65).  15 fp64
// instructions against 22 for the Taylor-13 form above; <= 1 ulp.  Same clamp / NaN behaviour as dexp_scaled.
// (64 entries, not more: the table lives in LDS, see tools/gen_log_table.py.)
struct ExpTabEntry { double hi, lo; };
SSME_HD double dexp_scaled_t(double x, int sc, const ExpTabEntry* tab) {
#ifdef SSME_AB_TAYLOR_EXP          // timing A/B only (tools/ab_build.py): results differ from the specification
    (void)tab;
    return dexp_scaled(x, sc);
#endif
    const double INV = 92.33248261689366;                    // 64 / ln 2
    const double C_HI = 6.93147180369123816490e-01 * 0.015625, C_LO = 1.90821492927058770002e-10 * 0.015625;   // ln2 / 64, hi + lo (exact scalings)
    const double SH = 6755399441055744.0;                    // 1.5 * 2^52
    const double xc = dminnum(dmaxnum(x, -746.0), 710.0);
    const double kf = dfma_c(xc, INV, SH) - SH;
    const int n = (int)kf;
    double r = dfma(-kf, C_HI, xc);
    r = dfma(-kf, C_LO, r);
    const ExpTabEntry e = tab[n & 63];
    double q = dfma_c(r, 0.001388888888888889, 0.008333333333333333);
    q = dfma_c(q, r, 0.041666666666666664);
    q = dfma_c(q, r, 0.16666666666666666);
    q = dfma_c(q, r, 0.5);
    const double p = dfma(r * r, q, r);
    const double res = e.hi + dfma(e.hi, p, e.lo);
    return dldexp(res, (n >> 6) + sc);
}

// round-to-nearest-even of v in [0, 2^52) to an integer, by the 2^52 trick
SSME_HD uint64_t rne_u52(double v) { return d2bits(v + 4503599627370496.0) & 0x000fffffffffffffull; }

// ---- log -----------------------------------------------------------------------------
// core for positive NORMAL x; kadj is added to the binary exponent (subnormal pre-scaling)
SSME_HD double dlog_core(double x, int kadj) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ux = d2bits(x);
    uint32_t hx = (uint32_t)(ux >> 32);
    int k = (int)(hx >> 20) - 1023 + kadj;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
    k += (int)(i >> 20);
    const double m = bits2d(ux);
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * dfma_c(w, dfma_c(w, Lg6, Lg4), Lg2);
    const double t2 = z * dfma_c(w, dfma_c(w, dfma_c(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * LN2_HI - ((hfsq - dfma(s, hfsq + R, dk * LN2_LO)) - f);
}

// positive normal inputs only: every uniform in (0,1] and every Gamma-test argument is one
SSME_HD double dlog_pn(double x) { return dlog_core(x, 0); }

// full-domain log (subnormals, 0, inf, negative, NaN)
SSME_HD double dlog(double x) {
    const bool sub = (d2bits(x) >> 52) == 0;
    const double xs = sub ? x * 0x1.0p54 : x;
    double res = dlog_core(xs, sub ? -54 : 0);
    if (x == dinf()) res = x;
    if (x == 0.0) res = -dinf();
    if (x < 0.0) res = dnan();
    if (x != x) res = x;
    return res;
}

// ---- log of a uniform: table + short series, no division ------------------------------------------------------------
// The draws of the bootstrap filter's hot loop (exponential spacings, Box-Muller radius) take -log(u) of uniforms that lie
// strictly inside (0,1).  fdlibm's log spends half its instructions on the IEEE division s = f/(2+f); this one takes
// c_i ~ 1/m and l_i = -log(c_i) from a 64-entry table indexed by the top 6 mantissa bits, so that r = fma(m, c_i, -1)
// has |r| < 2^-7 and log(x) = k ln2 + l_i + log1p(r) with log1p a degree-7 series (truncation < 2^-59).
// ABSOLUTE error < 2^-51 for any positive normal x (the relative error is unbounded next to x = 1, which the callers
// never need: u <= 1 - 2^-41).  Same operation sequence in oracle/ssme_oracle.cpp.
struct LogTabEntry { double c, l; };
SSME_HD double dlog_u(double x, const LogTabEntry* tab) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const uint64_t ux = d2bits(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const int k = (int)(hx >> 20) - 1023;
    const LogTabEntry e = tab[(hx >> 14) & 63u];
    const double m = bits2d((ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double r = dfma(m, e.c, -1.0);
    double q = dfma_c(r, 1.4285714285714285e-01, -1.6666666666666666e-01);
    q = dfma_c(q, r, 2.0000000000000001e-01);
    q = dfma_c(q, r, -2.5000000000000000e-01);
    q = dfma_c(q, r, 3.3333333333333331e-01);
    q = dfma_c(q, r, -5.0000000000000000e-01);
    const double p = dfma(r * r, q, r);
    const double dk = (double)k;
    return dfma(dk, LN2_HI, e.l) + dfma(dk, LN2_LO, p);
}
// The same for the EXPONENTIAL SPACINGS of the multinomial resampler (round 3): E = -log(u) is quantised to 2^-35 the moment it
// is formed (qE = rne(E 2^35)), so the series stops at r^5 (truncation |r|^6 / 6 < 2^-44) and k ln2 is one
// fma on the rounded constant: absolute error < 2^-43, three fp64 instructions less per spacing.  Not for the Box-Muller
// radius (dlog_u above keeps its 2^-51).
SSME_HD double dlog_u32(double x, const LogTabEntry* tab) {
    const double LN2 = 6.93147180559945286227e-01;
    const uint64_t ux = d2bits(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const int k = (int)(hx >> 20) - 1023;
    const LogTabEntry e = tab[(hx >> 14) & 63u];
    const double m = bits2d((ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double r = dfma(m, e.c, -1.0);
    double q = dfma_c(r, 2.0000000000000001e-01, -2.5000000000000000e-01);
    q = dfma_c(q, r, 3.3333333333333331e-01);
    q = dfma_c(q, r, -5.0000000000000000e-01);
    const double p = dfma(r * r, q, r);
    return dfma((double)k, LN2, e.l + p);
}
// uniforms strictly inside (0,1): midpoints of a 2^-40 / 2^-32 grid, and [0,1) on a 2^-24 grid (Box-Muller angle)
SSME_HD double u01_mid40(uint32_t a, uint32_t b) {          // a: 32 bits, top 8 bits of b
    const uint64_t man = ((uint64_t)a << 20) | ((uint64_t)(b >> 24) << 12) | 0x800ull;
    return 2.0 - bits2d(0x3ff0000000000000ull | man);
}
SSME_HD double u01_lo24(uint32_t b) {                       // low 24 bits of b
    return bits2d(0x3ff0000000000000ull | ((uint64_t)(b & 0x00ffffffu) << 28)) - 1.0;
}
SSME_HD double u01_mid32(uint32_t a) {
    return 2.0 - bits2d(0x3ff0000000000000ull | ((uint64_t)a << 20) | 0x80000ull);
}

// ---- sin(2 pi u), cos(2 pi u), u in [0,1) -----------------------------------------------
SSME_HD void dsincos2pi(double u, double* sn, double* cs) {
    const double SH = 6755399441055744.0;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double t = 4.0 * u;
    const double qf = (t + SH) - SH;
    const int q = (int)qf;
    const double r = t - qf;
    const double a = r * PIO2_HI;
    const double al = dfma(r, PIO2_HI, -a) + r * PIO2_LO;
    const double z = a * a;
    const double v = z * a;
    const double rs = dfma_c(z, dfma_c(z, dfma_c(z, dfma_c(z, S6, S5), S4), S3), S2);
    const double u1 = dfma(-v, rs, 0.5 * al);
    const double u2 = dfma(z, u1, -al);
    const double u3 = dfma(-v, S1, u2);
    const double s0 = a - u3;
    const double rc = z * dfma_c(z, dfma_c(z, dfma_c(z, dfma_c(z, dfma_c(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    const double c0 = wv + (((1.0 - wv) - hz) + dfma(z, rc, -(a * al)));
    const int qq = q & 3;
    const double ss = (qq & 1) ? c0 : s0;
    const double cc = (qq & 1) ? s0 : c0;
    *sn = (qq & 2) ? -ss : ss;
    *cs = ((qq + 1) & 2) ? -cc : cc;
}

// ---- sin, cos of 2 pi k / 2^24 for the 24-bit Box-Muller angle of the hot loops (round 3) -------------------------------
// k = i 2^18 + d with the NEAREST table angle i (64 entries {sin, cos}(2 pi i / 64), nearest doubles) and a signed remainder
// |d| <= 2^17, i.e. |delta| = 2 pi |d| / 2^24 <= 0.0491: sin(delta) to delta^7 and cos(delta) - 1 to delta^8 (truncation
// < 5e-18), then the rotation with the small terms added last.  Absolute error < 3e-16 (2 ulp of 1); 21 instructions and
// one 16-byte table read against 45 for the octant-reduction form (dsincos2pi, which the Gamma draws and t = 0 kernels keep).
struct SinCosEntry { double s, c; };
SSME_HD void dsincos_k24(uint32_t k, const SinCosEntry* tab, double* sn, double* cs) {
    const double STEP = 3.74507028292392863750e-07;           // 2 pi / 2^24, nearest double
    const uint32_t kr = (k & 0x00ffffffu) + 0x20000u;      // + half a table step
    const SinCosEntry e = tab[(kr >> 18) & 63u];
    const int d = (int)(kr & 0x3ffffu) - 0x20000;          // [-2^17, 2^17)
    const double dl = (double)d * STEP;
    const double z = dl * dl;
    double sp = dfma_c(z, -1.9841269841269841e-04, 8.3333333333333332e-03);
    sp = dfma_c(sp, z, -1.6666666666666666e-01);
    const double sd = dfma(dl, sp * z, dl);                // sin(delta)
    double cp = dfma_c(z, 2.4801587301587302e-05, -1.3888888888888889e-03);
    cp = dfma_c(cp, z, 4.1666666666666664e-02);
    cp = dfma_c(cp, z, -5.0000000000000000e-01);
    const double cm = cp * z;                              // cos(delta) - 1
    *sn = e.s + dfma(e.c, sd, e.s * cm);
    *cs = e.c + dfma(-e.s, sd, e.c * cm);
}

SSME_HD double dsqrt(double x) { return __builtin_sqrt(x); }
// sqrt of a positive NORMAL double well inside the exponent range (the Box-Muller radicand -2 log u lies in [2^-40, 56]):
// the compiler's own v_rsq_f64 + two Goldschmidt / Newton steps without its range scaling and class fix-up (8 instructions
// less).  Correctly rounded like dsqrt -- the host copy IS sqrt -- which tests/test_parity_gpu.py checks bit for bit.
SSME_HD double dsqrt_pn(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = dfma(-h, g, 0.5);
    g = dfma(g, r, g);
    h = dfma(h, r, h);
    const double d0 = dfma(-g, g, x);
    g = dfma(d0, h, g);
    const double d1 = dfma(-g, g, x);
    return dfma(d1, h, g);
#else
    return __builtin_sqrt(x);
#endif
}

}  // namespace ssme
