// ssme_oracle.cpp -- CPU ORACLE for the bootstrap-particle-filter hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The product path (ssme_amd/csrc, the
// C-ABI in include/ssme_pf.h) never links, calls or falls back to anything in oracle/.
//
// PARITY STATUS: "parity unpinned".  The arithmetic of the path lives in the third-party
// header-only library `pf` (github tbrown122387/pf, NO version pin: reference
// CMakeLists.txt:12 `find_package(pf CONFIG REQUIRED)`), which is absent from
// /root/reference, as are Eigen3 and Catch2; the reference cannot be compiled here and
// none of its own tests pins a filter output (test/test_pswarm.cpp:251-252 and
// test/test_liu_west.cpp:172,198-199 assert only loglike^2 > 0 and E[42] == 42 on
// uninitialised inputs).  What IS pinned from the reference's own files:
//   * param::pack inverse transforms / log-Jacobian KATs   test/test_parameters.cpp:114-145
//   * log-mean-exp of replicates KAT                       test/test_thread_pool.cpp:39-46
//   * constant-functional expectation == 42                test/test_pswarm.cpp:252
// and from the reference RUN here: include/ssme/thread_pool.h is the one hot-path header that needs only the
// standard library; oracle/_ref builds it where it lies (oracle/ref_thread_pool_driver.cpp, `make _ref`) and
// orc_log_mean_exp / the product's log-mean-exp are checked against thread_pool<>::work itself.
// Independent anchors added by this repo: exact Kalman log-likelihood of a linear-Gaussian
// model, agreement between the two RNG modes below, Random123 Philox4x32-10 KATs.
//
// Two modes restate the same algorithm:
//  (A) "reference-faithful": std::mt19937 + std::normal_distribution +
//      std::discrete_distribution, reference operation order in logGEv.  Used for the
//      statistical cross-check and as bench.py's cpu_baseline ("port").
//  (B) "kernel-matched": Philox4x32-10 counter RNG, libm-free fp64 math, and the exact
//      fixed-point weight cdf documented in DESIGN.md section 4 (integer sums: no summation
//      tree to mirror).  The HIP kernels are compared BIT FOR BIT against this mode.
//
// Reference lines followed (all relative to /root/reference):
//   model callbacks      example/univ_svol_bootstrap_filter.h:55-103   (svol_bs)
//                        test/test_pswarm.cpp:80-134                    (svol_leverage)
//   driver loop          example/estimate_univ_svol.h:108-131          (log_like_eval)
//   SISR step, LSE, expectations, resample schedule (in-tree twin of the external
//   pf::BSFilter::filter)  include/ssme/liu_west_filter.h:1608-1761
//   multinomial resampling by sorted uniforms (exponential spacings), weight reset
//                        include/ssme/liu_west_filter.h:91-145
//   replicate aggregation (log-mean-exp)  include/ssme/thread_pool.h:263-268
//   parameter transforms include/ssme/parameters.h:317-457
// pf-internal details that cannot be read here are marked [pf-recollection].

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123).  Counter = {index, t, replicate, stream}
// ---------------------------------------------------------------------------------------
inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
        philox_round(c, k);
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// 52 random bits from two words -> [0,1) and (0,1]
inline double u12(uint32_t a, uint32_t b) {                       // 52 random bits as the mantissa of a double in [1,2)
    const uint64_t u = 0x3ff0000000000000ull | ((((uint64_t)a << 32) | b) >> 12);
    double d; std::memcpy(&d, &u, 8); return d;
}
inline double u01_co(uint32_t a, uint32_t b) { return u12(a, b) - 1.0; }          // [0,1)
inline double u01_oc(uint32_t a, uint32_t b) { return 2.0 - u12(a, b); }          // (0,1]

enum { STREAM_PROP = 0, STREAM_RESAMP = 1, STREAM_RESAMP_EXTRA = 2 };

// ---------------------------------------------------------------------------------------
// libm-free fp64 math.  Only + - * fma, IEEE sqrt and division, and integer bit moves, so
// that the device implementation (ssme_amd/csrc/device_math.h) yields identical bits.
// Built with -ffp-contract=off.  Algorithms: exp = Cody-Waite reduction + Taylor-13;
// log = fdlibm e_log.c main path; sin/cos = fdlibm k_sin.c / k_cos.c kernels on r*pi/2.
// ---------------------------------------------------------------------------------------
inline double bits_to_double(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }
inline uint64_t double_to_bits(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
inline double pow2i(int n) { return bits_to_double((uint64_t)(n + 1023) << 52); }   // n in [-1022,1023]

// exp(x) * 2^sc.  Clamp squashes NaN to the lower bound (fmax/fmin = IEEE maxNum/minNum).
double o_exp_scaled(double x, int sc) {
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double SH = 6755399441055744.0;  // 1.5 * 2^52
    const double xc = std::fmin(std::fmax(x, -746.0), 710.0);
    const double kf = std::fma(xc, LOG2E, SH) - SH;
    const int k = (int)kf;
    double r = std::fma(-kf, LN2_HI, xc);
    r = std::fma(-kf, LN2_LO, r);
    double q = 1.6059043836821613e-10;            // 1/13!
    q = std::fma(q, r, 2.08767569878681e-09);     // 1/12!
    q = std::fma(q, r, 2.505210838544172e-08);    // 1/11!
    q = std::fma(q, r, 2.755731922398589e-07);    // 1/10!
    q = std::fma(q, r, 2.7557319223985893e-06);   // 1/9!
    q = std::fma(q, r, 2.48015873015873e-05);     // 1/8!
    q = std::fma(q, r, 0.0001984126984126984);    // 1/7!
    q = std::fma(q, r, 0.001388888888888889);     // 1/6!
    q = std::fma(q, r, 0.008333333333333333);     // 1/5!
    q = std::fma(q, r, 0.041666666666666664);     // 1/4!
    q = std::fma(q, r, 0.16666666666666666);      // 1/3!
    q = std::fma(q, r, 0.5);                      // 1/2!
    const double e = std::fma(r * r, q, r);
    const double p = 1.0 + e;
    return std::ldexp(p, k + sc);
}
double o_exp(double x) { return o_exp_scaled(x, 0); }

double o_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0.0) return -std::numeric_limits<double>::infinity();
    if (x == std::numeric_limits<double>::infinity()) return x;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = 0;
    uint64_t ux = double_to_bits(x);
    if ((ux >> 52) == 0) { x = x * 0x1.0p54; k -= 54; ux = double_to_bits(x); }
    uint32_t hx = (uint32_t)(ux >> 32);
    k += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
    k += (int)(i >> 20);
    const double m = bits_to_double(ux);          // in [sqrt(2)/2, sqrt(2))
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * std::fma(w, std::fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * std::fma(w, std::fma(w, std::fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * LN2_HI - ((hfsq - std::fma(s, hfsq + R, dk * LN2_LO)) - f);
}

// log of a uniform strictly inside (0,1): 64-entry table + degree-7 series, no division (mirror of ssme_math.h: dlog_u).
// The bootstrap filter's hot-loop draws (exponential spacings, Box-Muller radius) use it; absolute error < 2^-51.
#include "ssme_log_table.h"
struct LogTabEntry { double c, l; };
static const LogTabEntry LOG_TABLE[SSME_LOG_TABLE_SIZE] = {SSME_LOG_TABLE_ROWS};
double o_log_u(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const uint64_t ux = double_to_bits(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const int k = (int)(hx >> 20) - 1023;
    const LogTabEntry e = LOG_TABLE[(hx >> 14) & 63u];
    const double m = bits_to_double((ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double r = std::fma(m, e.c, -1.0);
    double q = std::fma(r, 1.4285714285714285e-01, -1.6666666666666666e-01);
    q = std::fma(q, r, 2.0000000000000001e-01);
    q = std::fma(q, r, -2.5000000000000000e-01);
    q = std::fma(q, r, 3.3333333333333331e-01);
    q = std::fma(q, r, -5.0000000000000000e-01);
    const double p = std::fma(r * r, q, r);
    const double dk = (double)k;
    return std::fma(dk, LN2_HI, e.l) + std::fma(dk, LN2_LO, p);
}
// log of a 32-bit uniform for the exponential spacings, which are quantised to 2^-35 as they are formed: series to r^5, one fma
// for k ln2 (mirror of ssme_math.h: dlog_u32; absolute error < 2^-43)
double o_log_u32(double x) {
    const double LN2 = 6.93147180559945286227e-01;
    const uint64_t ux = double_to_bits(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const int k = (int)(hx >> 20) - 1023;
    const LogTabEntry e = LOG_TABLE[(hx >> 14) & 63u];
    const double m = bits_to_double((ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double r = std::fma(m, e.c, -1.0);
    double q = std::fma(r, 2.0000000000000001e-01, -2.5000000000000000e-01);
    q = std::fma(q, r, 3.3333333333333331e-01);
    q = std::fma(q, r, -5.0000000000000000e-01);
    const double p = std::fma(r * r, q, r);
    return std::fma((double)k, LN2, e.l + p);
}
// sin, cos of 2 pi k / 2^24 for the 24-bit Box-Muller angle (mirror of ssme_math.h: dsincos_k24): nearest of 64 table angles,
// sin / cos - 1 of the remainder by short series, rotation
struct SinCosEntry { double s, c; };
static const SinCosEntry SINCOS_TABLE[SSME_SINCOS_TABLE_SIZE] = {SSME_SINCOS_TABLE_ROWS};
void o_sincos_k24(uint32_t k, double* sn, double* cs) {
    const double STEP = 3.74507028292392863750e-07;
    const uint32_t kr = (k & 0x00ffffffu) + 0x20000u;
    const SinCosEntry e = SINCOS_TABLE[(kr >> 18) & 63u];
    const int d = (int)(kr & 0x3ffffu) - 0x20000;
    const double dl = (double)d * STEP;
    const double z = dl * dl;
    double sp = std::fma(z, -1.9841269841269841e-04, 8.3333333333333332e-03);
    sp = std::fma(sp, z, -1.6666666666666666e-01);
    const double sd = std::fma(dl, sp * z, dl);
    double cp = std::fma(z, 2.4801587301587302e-05, -1.3888888888888889e-03);
    cp = std::fma(cp, z, 4.1666666666666664e-02);
    cp = std::fma(cp, z, -5.0000000000000000e-01);
    const double cm = cp * z;
    *sn = e.s + std::fma(e.c, sd, e.s * cm);
    *cs = e.c + std::fma(-e.s, sd, e.c * cm);
}
// The bootstrap filter's exp (mirror of ssme_math.h: dexp_scaled_t): 64-entry double-double table of 2^(j/64) + degree-6 series
struct ExpTabEntry { double hi, lo; };
static const ExpTabEntry EXP_TABLE[SSME_EXP_TABLE_SIZE] = {SSME_EXP_TABLE_ROWS};
double o_exp_scaled_t(double x, int sc) {
    const double INV = 92.33248261689366;
    const double C_HI = 6.93147180369123816490e-01 * 0.015625, C_LO = 1.90821492927058770002e-10 * 0.015625;
    const double SH = 6755399441055744.0;
    const double xc = std::fmin(std::fmax(x, -746.0), 710.0);
    const double kf = std::fma(xc, INV, SH) - SH;
    const int n = (int)kf;
    double r = std::fma(-kf, C_HI, xc);
    r = std::fma(-kf, C_LO, r);
    const ExpTabEntry e = EXP_TABLE[n & 63];
    double q = std::fma(r, 0.001388888888888889, 0.008333333333333333);
    q = std::fma(q, r, 0.041666666666666664);
    q = std::fma(q, r, 0.16666666666666666);
    q = std::fma(q, r, 0.5);
    const double p = std::fma(r * r, q, r);
    const double res = e.hi + std::fma(e.hi, p, e.lo);
    return std::ldexp(res, (n >> 6) + sc);
}
double o_exp_t(double x) { return o_exp_scaled_t(x, 0); }

// uniforms strictly inside (0,1) on a 2^-40 / 2^-32 midpoint grid; [0,1) on a 2^-24 grid (Box-Muller angle)
inline double u01_mid40(uint32_t a, uint32_t b) {
    const uint64_t man = ((uint64_t)a << 20) | ((uint64_t)(b >> 24) << 12) | 0x800ull;
    return 2.0 - bits_to_double(0x3ff0000000000000ull | man);
}
inline double u01_lo24(uint32_t b) { return bits_to_double(0x3ff0000000000000ull | ((uint64_t)(b & 0x00ffffffu) << 28)) - 1.0; }
inline double u01_mid32(uint32_t a) { return 2.0 - bits_to_double(0x3ff0000000000000ull | ((uint64_t)a << 20) | 0x80000ull); }

// sin(2*pi*u), cos(2*pi*u) for u in [0,1)
void o_sincos2pi(double u, double* sn, double* cs) {
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
    const double r = t - qf;                      // exact, in [-0.5, 0.5]
    const double a = r * PIO2_HI;
    const double al = std::fma(r, PIO2_HI, -a) + r * PIO2_LO;
    const double z = a * a;
    // sine kernel (fdlibm k_sin polynomial as an fma Horner chain)
    const double v = z * a;
    const double rs = std::fma(z, std::fma(z, std::fma(z, std::fma(z, S6, S5), S4), S3), S2);
    const double u1 = std::fma(-v, rs, 0.5 * al);
    const double u2 = std::fma(z, u1, -al);
    const double u3 = std::fma(-v, S1, u2);
    const double s0 = a - u3;
    // cosine kernel (fdlibm k_cos polynomial as an fma Horner chain)
    const double rc = z * std::fma(z, std::fma(z, std::fma(z, std::fma(z, std::fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    const double c0 = wv + (((1.0 - wv) - hz) + std::fma(z, rc, -(a * al)));
    switch (q & 3) {
        case 0: *sn = s0;  *cs = c0;  break;
        case 1: *sn = c0;  *cs = -s0; break;
        case 2: *sn = -s0; *cs = -c0; break;
        default: *sn = -c0; *cs = s0; break;
    }
}

const double HALF_LOG_2PI = 0.91893853320467274178;
const double NEG_INF = -std::numeric_limits<double>::infinity();
const double POS_INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------------------------------
// Models (kernel-matched form).  theta is UNTRANSFORMED, as the reference's model ctors
// receive it after pack::get_untrans_params (univ_svol_bootstrap_filter.h:55-61).
// ---------------------------------------------------------------------------------------
enum { MODEL_SVOL = 0, MODEL_SVOL_LEVERAGE = 1, MODEL_LIN_GAUSS = 2, MODEL_USER0 = 3 };
enum { RESAMP_MULTINOMIAL = 0, RESAMP_SYSTEMATIC = 1, RESAMP_STRATIFIED = 2, RESAMP_MULTINOMIAL_IID = 3 };

// MODEL_USER0: the device's model extension point (ssme_amd/csrc/model_api.h) mirrored by CALLBACKS -- the test that defines a
// user model for the device restates it independently (in Python, through ctypes) and hands the restatement in here
typedef double (*user_prop_fn)(double x, double zn, double zcov);
typedef double (*user_logg_fn)(double y, double x);
// ... with a vector state / observation (model_api.h: dim_x, dim_y <= 4): q1Samp, fSamp and logGEv on arrays
typedef void (*user_initv_fn)(const double* zn, double* x0);
typedef void (*user_propv_fn)(const double* x, const double* zn, double zcov, double* xn);
typedef double (*user_loggv_fn)(const double* y, const double* x);
enum { STREAM_XDIM = 96 };         // counter stream of state component d >= 1: STREAM_XDIM + d (component 0: the pair's draw)

struct ModelConst {          // derived once per replicate on the host, in this op order
    int model;
    double a0, a1, a2, a3, a4, a5;  // meaning depends on model, see derive()
    int bad;                        // 1 -> logG == -inf (e.g. beta <= 0)
    user_prop_fn u_prop = nullptr;  // MODEL_USER0 only
    user_logg_fn u_logg = nullptr;
    int dx = 1, dy = 1;             // MODEL_USER0 with a vector state / observation
    user_initv_fn v_init = nullptr;
    user_propv_fn v_prop = nullptr;
    user_loggv_fn v_logg = nullptr;
};

ModelConst derive(int model, const double* th) {
    ModelConst c{};
    c.model = model;
    if (model == MODEL_SVOL) {          // th = (beta, phi, sigma)
        const double beta = th[0], phi = th[1], sigma = th[2];
        c.a0 = phi;                                   // phi
        c.a1 = sigma;                                 // sigma
        c.a2 = sigma / std::sqrt(1.0 - phi * phi);    // stationary sd (q1Samp, :65-70)
        c.a3 = o_log(beta);                           // log beta
        c.a4 = 1.0 / (beta * beta);                   // beta^-2
        c.bad = !(beta > 0.0);
    } else if (model == MODEL_SVOL_LEVERAGE) {   // th = (phi, mu, sigma, rho)
        const double phi = th[0], mu = th[1], sigma = th[2], rho = th[3];
        c.a0 = phi; c.a1 = mu;
        c.a2 = sigma / std::sqrt(1.0 - phi * phi);    // stationary sd (test_pswarm.cpp:80-86)
        c.a3 = sigma * std::sqrt(1.0 - phi * phi);    // transition sd (:95)
        c.a4 = rho * sigma;                           // leverage coefficient (:94)
        c.bad = 0;
    } else {                              // linear Gaussian: th = (phi, sigma, tau)
        const double phi = th[0], sigma = th[1], tau = th[2];
        c.a0 = phi; c.a1 = sigma;
        c.a2 = sigma / std::sqrt(1.0 - phi * phi);
        c.a3 = o_log(tau);
        c.a4 = 1.0 / tau;
        c.bad = !(tau > 0.0);
    }
    return c;
}

// q1Samp: stationary draw (univ_svol_bootstrap_filter.h:65-70, test_pswarm.cpp:80-86)
inline double m_init(const ModelConst& c, double zn) { return zn * c.a2; }

// fSamp (univ_svol_bootstrap_filter.h:74-79; test_pswarm.cpp:90-97)
inline double m_prop(const ModelConst& c, double x, double zn, double zcov) {
    if (c.model == MODEL_USER0) return c.u_prop(x, zn, zcov);
    if (c.model == MODEL_SVOL_LEVERAGE) {
        const double e = o_exp_t(-0.5 * x);
        const double mean = (c.a1 + c.a0 * (x - c.a1)) + (c.a4 * zcov) * e;
        return mean + zn * c.a3;
    }
    return c.a0 * x + zn * c.a1;
}

// logGEv (univ_svol_bootstrap_filter.h:83-86; test_pswarm.cpp:101-108).  Kernel form of
// evalUnivNorm(y, 0, s, log=true) with s = beta*exp(x/2):
//   -log s - 0.5 log 2pi - 0.5 (y/s)^2  =  -(log beta + x/2) - 0.5 log 2pi - 0.5 y^2 beta^-2 e^-x
// evalUnivNorm returns -inf when s <= 0 [pf-recollection]; mirrored by `bad` and by the
// underflow guard (s == 0 in fp64 when log s < -745.13).
inline double m_logg(const ModelConst& c, double y, double x) {
    if (c.model == MODEL_USER0) return c.bad ? NEG_INF : c.u_logg(y, x);
    if (c.model == MODEL_LIN_GAUSS) {
        if (c.bad) return NEG_INF;
        const double d = (y - x) * c.a4;
        return (-c.a3 - HALF_LOG_2PI) - 0.5 * (d * d);
    }
    double logb = 0.0, ib2 = 1.0;
    if (c.model == MODEL_SVOL) { if (c.bad) return NEG_INF; logb = c.a3; ib2 = c.a4; }
    const double hl = logb + 0.5 * x;
    if (hl < -745.1332191019412) return NEG_INF;
    const double e = o_exp_t(-x);
    const double q = (y * y) * ib2;
    return (-hl - HALF_LOG_2PI) - 0.5 * (q * e);
}

// ---------------------------------------------------------------------------------------
// Exact fixed-point weight cdf with per-tile scales.
//   tile b (2048 particles):  m_b = max logw,  q_i = rne( exp(logw_i - m_b) * 2^41 )
//                             loc_j = sum_{i<=j in tile} q_i      (uint64, EXACT, monotone)
//                             A_b   = loc_last
//   across tiles:             m = max_b m_b,  A'_b = rint( A_b * exp(m_b - m) * 2^(rg-41) ),
//                             rg = 52 - ceil(log2(Npad)),  T'_b = sum_{c<=b} A'_c (exact),  S' = T'_last
//   every integer stays below 2^53, so the device carries them EXACTLY in fp64 registers
//   log-sum-exp:              m + log(S' * 2^-rg)
//   ancestor of target tau in [0, S']:
//       b* = min(#{b : T'_b < tau}, B-1);  d = tau - (T'_b* - A'_b*)
//       tl = ceil( (double)d * ((double)A_b* / (double)A'_b*) );  j* = min(#{j : loc_j < tl}, 2047)
//       anc = min(b* * 2048 + j*, N-1)
// Integer sums are associative: no summation tree to mirror; the ancestor is an integer COUNT,
// independent of the kernel's scan and search strategy.  No global max pass is needed, so the
// device runs ONE kernel per filter step.
// ---------------------------------------------------------------------------------------
constexpr int TILE = 2048;
constexpr int TILE_SHIFT = 41;                // tile-local fixed point: q <= 2^41, tile sums <= 2^52 (exact in fp64)
constexpr int E_SHIFT = 35;                   // exponential spacings: qE = round(E * 2^35)
constexpr double TWO52 = 4503599627370496.0;
constexpr uint64_t MASK52 = (1ull << 52) - 1;

inline int ceil_log2(int n) { int k = 0; while ((1ll << k) < n) ++k; return k; }
// round-to-nearest-even of v in [0, 2^52) via the 2^52 trick (same two ops on the device)
inline uint64_t rne_u64(double v) { return (uint64_t)std::rint(v); }     // v in [0, 2^52): integer-valued double
inline uint64_t tau_to_u64(double tau) {
    double c = std::ceil(tau);
    c = std::fmin(std::fmax(c, 0.0), 9.2e18);          // NaN -> 0
    return (uint64_t)c;
}
inline uint64_t rint_to_u64(double v) {
    double c = std::rint(v);
    c = std::fmin(std::fmax(c, 0.0), 9.2e18);          // NaN -> 0
    return (uint64_t)c;
}

enum { STREAM_GAMMA = 16 };

// ---------------------------------------------------------------------------------------
// Kernel-matched filter (mode B), one replicate.
// ---------------------------------------------------------------------------------------
struct Filter {
    int model, N, resamp, rs;
    uint32_t key[2];
    uint32_t rep;
    ModelConst mc;
    int B, Npad, rshift;
    int t;
    std::vector<double> x, xprev, logw;
    std::vector<double> xv[4], xvprev[4];     // vector states: components 1 .. dx-1 (component 0 is x)
    std::vector<uint64_t> loc, A, Ap, Tincl;  // tile-local sums, tile sums (tile scale), rescaled tile sums, their prefixes
    std::vector<double> mb;                   // per-tile max log-weight
    std::vector<uint32_t> anc;
    double m, prev, loglik, last_ll;
    uint64_t Sint;
    // Bootstrap filter (k_filter_step): ONE Philox call per particle pair and time step, counter (pair, t, filter,
    // STREAM_PROP): words 0-1 -> Box-Muller (radius uniform 40 bits, angle 24 bits), words 2-3 -> the pair's two
    // exponential spacings (32 bits each); logs by o_log_u, weight exps by the table form.  The Liu-West filter's two draws
    // use the same construction on their own counter streams (pair_stream).  false: 52-bit uniforms, o_log, Taylor exp
    // (the round-1 arithmetic; kept for the accuracy comparisons of tests/test_oracle_cpu.py).
    bool bootstrap_draws = false;
    int pair_stream = STREAM_PROP;     // the counter's stream word of that one call (the Liu-West draws use their own)
    // particles per tile (2048, 1024 or 512): weights are fixed point relative to their TILE's maximum and the multinomial
    // resampler draws one Gamma variate per tile, so the tile size is part of the specification.  The device's rule when
    // the caller does not choose (pf_api.hip: default_tile): 2048 for N <= 2048 and N > 2^18, 512 in between.
    int tile = 2048;

    void init(int model_, int N_, int resamp_, int rs_, uint64_t seed, uint32_t rep_, const double* th) {
        model = model_; N = N_; resamp = resamp_; rs = rs_ < 1 ? 1 : rs_;
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32); rep = rep_;
        mc = derive(model, th);
        B = (N + tile - 1) / tile; Npad = B * tile;
        rshift = 52 - ceil_log2(Npad);
        x.assign(Npad, 0.0); xprev.assign(Npad, 0.0); logw.assign(Npad, 0.0); loc.assign(Npad, 0);
        A.assign(B, 0); Ap.assign(B, 0); Tincl.assign(B, 0); mb.assign(B, 0.0);
        anc.assign(Npad, 0u);
        reset();
    }
    void reset() { t = 0; loglik = 0.0; last_ll = 0.0; prev = o_log((double)N); m = 0; Sint = 0; }

    // standard normal for particle i at time tt (Box-Muller on the pair i>>1)
    double normal(int i, int tt) const {
        const uint32_t ctr[4] = {(uint32_t)(i >> 1), (uint32_t)tt, rep, (uint32_t)(bootstrap_draws ? pair_stream : STREAM_PROP)};
        uint32_t o[4]; philox4x32_10(ctr, key, o);
        if (bootstrap_draws) {
            const double rad = std::sqrt(-2.0 * o_log_u(u01_mid40(o[0], o[1])));
            double sn, cs; o_sincos_k24(o[1], &sn, &cs);
            return (i & 1) ? rad * sn : rad * cs;
        }
        const double u1 = u01_oc(o[0], o[1]), u2 = u01_co(o[2], o[3]);
        const double rad = std::sqrt(-2.0 * o_log(u1));
        double sn, cs; o_sincos2pi(u2, &sn, &cs);
        return (i & 1) ? rad * sn : rad * cs;
    }
    // exponential spacing E_i of the multinomial resampler at time tt
    double spacing(int i, int tt, int stream) const {
        if (bootstrap_draws) {
            const uint32_t ctr[4] = {(uint32_t)(i >> 1), (uint32_t)tt, rep, (uint32_t)pair_stream};
            uint32_t o[4]; philox4x32_10(ctr, key, o);
            return -o_log_u32(u01_mid32(o[2 + (i & 1)]));
        }
        uint32_t wa, wb; resamp_words(i, tt, &wa, &wb, stream);
        return -o_log(u01_oc(wa, wb));
    }
    void resamp_words(int i, int tt, uint32_t* a, uint32_t* b, int stream = STREAM_RESAMP) const {
        const uint32_t ctr[4] = {(uint32_t)(i >> 1), (uint32_t)tt, rep, (uint32_t)stream};
        uint32_t o[4]; philox4x32_10(ctr, key, o);
        *a = o[2 * (i & 1)]; *b = o[2 * (i & 1) + 1];
    }
    void extra_words(int tt, uint32_t o[4], int stream = STREAM_RESAMP_EXTRA) const {
        const uint32_t ctr[4] = {0u, (uint32_t)tt, rep, (uint32_t)stream};
        philox4x32_10(ctr, key, o);
    }
    // Gamma(shape) draw for tile b at time tt: Marsaglia & Tsang (2000), counter-driven attempts
    double gamma_draw(int b, int tt, double shape, int stream_base = STREAM_GAMMA) const {
        const double d = shape - 0.3333333333333333;
        const double c = 1.0 / std::sqrt(9.0 * d);
        for (int a = 0; a < 32; ++a) {
            const uint32_t c1[4] = {(uint32_t)b, (uint32_t)tt, rep, (uint32_t)(stream_base + 2 * a)};
            const uint32_t c2[4] = {(uint32_t)b, (uint32_t)tt, rep, (uint32_t)(stream_base + 2 * a + 1)};
            uint32_t o1[4], o2[4];
            philox4x32_10(c1, key, o1); philox4x32_10(c2, key, o2);
            const double rad = std::sqrt(-2.0 * o_log(u01_oc(o1[0], o1[1])));
            double sn, cs; o_sincos2pi(u01_co(o1[2], o1[3]), &sn, &cs);
            const double xn = rad * cs;
            const double v = 1.0 + c * xn;
            if (v > 0.0) {
                const double v3 = (v * v) * v;
                const double lhs = o_log(u01_oc(o2[0], o2[1]));
                const double rhs = ((0.5 * (xn * xn) + d) - d * v3) + d * o_log(v3);
                if (lhs < rhs) return d * v3;
            }
        }
        return d;
    }

    // integer targets for the ancestors consumed at time tt (drawn against the cdf of step tt-1)
    void targets(int tt, std::vector<uint64_t>& tau, int s_spacing = STREAM_RESAMP, int s_extra = STREAM_RESAMP_EXTRA,
                 int s_gamma = STREAM_GAMMA) const {
        tau.assign(N, 0);
        const double Sd = (double)Sint;
        if (resamp == RESAMP_MULTINOMIAL) {
            // multinomial by sorted uniforms = exponential spacings (liu_west_filter.h:105-139):
            //   U_(i) = sum_{j<=i} E_j / sum_{j<=N+1} E_j.
            // Per tile the spacings are generated as Gamma_b * (E_j / sum_tile E), with
            // Gamma_b ~ Gamma(n_b) drawn directly: the normalised spacings are Dirichlet(1..1) and
            // independent of their sum, so the joint law of the U_(i) is unchanged.
            std::vector<double> gam(B), pgam(B);
            double run = 0.0;
            for (int b = 0; b < B; ++b) {
                const int nb = std::min(tile, N - b * tile);
                gam[b] = gamma_draw(b, tt, (double)nb, s_gamma);
                pgam[b] = run;
                run = run + gam[b];
            }
            uint32_t o[4]; extra_words(tt, o, s_extra);
            const double G = run + (-o_log(u01_oc(o[0], o[1])));
            const double scale = Sd / G;
            for (int b = 0; b < B; ++b) {
                const int nb = std::min(tile, N - b * tile);
                std::vector<uint64_t> locE(nb);
                uint64_t s = 0;
                for (int j = 0; j < nb; ++j) {
                    const double E = spacing(b * tile + j, tt, s_spacing);
                    s += (uint64_t)std::rint(E * 34359738368.0 /* 2^35 */);
                    locE[j] = s;
                }
                const double ratio = gam[b] / (double)s;
                for (int j = 0; j < nb; ++j) {
                    const double t1 = ratio * (double)locE[j];
                    const double t2 = pgam[b] + t1;
                    tau[b * tile + j] = tau_to_u64(t2 * scale);
                }
            }
        } else if (resamp == RESAMP_SYSTEMATIC) {
            uint32_t o[4]; extra_words(tt, o);
            const double u0 = u01_co(o[0], o[1]);
            const double scale = Sd / (double)N;
            for (int i = 0; i < N; ++i) tau[i] = tau_to_u64(((double)i + u0) * scale);
        } else if (resamp == RESAMP_STRATIFIED) {
            const double scale = Sd / (double)N;
            for (int i = 0; i < N; ++i) { uint32_t a, b; resamp_words(i, tt, &a, &b); tau[i] = tau_to_u64(((double)i + u01_co(a, b)) * scale); }
        } else {
            for (int i = 0; i < N; ++i) { uint32_t a, b; resamp_words(i, tt, &a, &b); tau[i] = tau_to_u64(u01_co(a, b) * Sd); }
        }
    }

    // ancestor of an integer target (see the header of this section)
    int search(uint64_t tau) const {
        int b = (int)(std::lower_bound(Tincl.begin(), Tincl.end(), tau) - Tincl.begin());   // #{T'_b < tau}
        if (b > B - 1) b = B - 1;
        const uint64_t d = tau - (Tincl[b] - Ap[b]);                     // unsigned, as on the device
        const double ratio = (double)A[b] / (double)Ap[b];
        const uint64_t tl = tau_to_u64((double)d * ratio);
        const uint64_t* tl_cdf = &loc[(size_t)b * tile];
        int j = (int)(std::lower_bound(tl_cdf, tl_cdf + tile, tl) - tl_cdf);   // #{loc_j < tl}
        if (j > tile - 1) j = tile - 1;
        return std::min(b * tile + j, N - 1);
    }

    // the exp of the weight arithmetic: the bootstrap filter's table form, or the Taylor form the Liu-West kernels use
    double xexp(double v, int sc) const { return bootstrap_draws ? o_exp_scaled_t(v, sc) : o_exp_scaled(v, sc); }

    // logw[0..N) -> per-tile maxima, tile-local exact cdf, rescaled tile sums, their prefixes; returns S' 2^-rg
    double build_cdf() {
        // per-tile NaN-propagating max, tile-local exact cdf
        for (int b = 0; b < B; ++b) {
            double mx = NEG_INF; bool nan = false;
            for (int j = 0; j < tile; ++j) {
                const int i = b * tile + j;
                if (i >= N) break;
                if (logw[i] != logw[i]) nan = true; else if (logw[i] > mx) mx = logw[i];
            }
            mb[b] = nan ? std::numeric_limits<double>::quiet_NaN() : mx;
            uint64_t s = 0;
            for (int j = 0; j < tile; ++j) {
                const int i = b * tile + j;
                if (i < N) s += rne_u64(xexp(logw[i] - mb[b], TILE_SHIFT));
                loc[i] = s;
            }
            A[b] = s;
        }
        // across tiles: global max (NaN if any tile is NaN: any NaN log-weight -> NaN log-likelihood,
        // as the reference's sums), rescaled integer tile sums and their exact prefixes
        {
            double mx = NEG_INF; bool nan = false;
            for (int b = 0; b < B; ++b) { if (mb[b] != mb[b]) nan = true; else if (mb[b] > mx) mx = mb[b]; }
            m = nan ? std::numeric_limits<double>::quiet_NaN() : mx;
        }
        uint64_t run = 0;
        for (int b = 0; b < B; ++b) {
            Ap[b] = rint_to_u64((double)A[b] * xexp(mb[b] - m, rshift - TILE_SHIFT));
            run += Ap[b]; Tincl[b] = run;
        }
        Sint = run;
        return Sint ? std::ldexp((double)Sint, -rshift) : std::numeric_limits<double>::quiet_NaN();
    }

    // standard normal of state component d >= 1 for particle i at time tt: words 0-1 of the pair's call on stream STREAM_XDIM + d
    double normal_dim(int i, int tt, int d) const {
        const uint32_t ctr[4] = {(uint32_t)(i >> 1), (uint32_t)tt, rep, (uint32_t)(STREAM_XDIM + d)};
        uint32_t o[4]; philox4x32_10(ctr, key, o);
        const double rad = std::sqrt(-2.0 * o_log_u(u01_mid40(o[0], o[1])));
        double sn, cs; o_sincos_k24(o[1], &sn, &cs);
        return (i & 1) ? rad * sn : rad * cs;
    }
    // MODEL_USER0 with a vector state / observation (model_api.h): the scalar step with arrays in the model calls
    double step_vec(const double* yv, double zcov) {
        const int DX = mc.dx;
        const bool resampled_prev = (t > 0) && (t % rs == 0);
        std::vector<double> lw_old(N, 0.0);
        for (int d = 1; d < DX; ++d) { if ((int)xv[d].size() != Npad) { xv[d].assign(Npad, 0.0); xvprev[d].assign(Npad, 0.0); } }
        if (t > 0) {
            xprev.swap(x);
            for (int d = 1; d < DX; ++d) xvprev[d].swap(xv[d]);
            if (resampled_prev) {
                std::vector<uint64_t> tau; targets(t, tau);
                for (int i = 0; i < N; ++i) anc[i] = (uint32_t)search(tau[i]);
            } else {
                for (int i = 0; i < N; ++i) lw_old[i] = logw[i];
            }
        }
        for (int i = 0; i < N; ++i) {
            double zin[4], xin[4] = {0, 0, 0, 0}, xo[4];
            zin[0] = normal(i, t);
            for (int d = 1; d < DX; ++d) zin[d] = normal_dim(i, t, d);
            if (t == 0) mc.v_init(zin, xo);
            else {
                const int j = resampled_prev ? (int)anc[i] : i;
                xin[0] = xprev[j];
                for (int d = 1; d < DX; ++d) xin[d] = xvprev[d][j];
                mc.v_prop(xin, zin, zcov, xo);
            }
            x[i] = xo[0];
            for (int d = 1; d < DX; ++d) xv[d][i] = xo[d];
            logw[i] = lw_old[i] + (mc.bad ? NEG_INF : mc.v_logg(yv, xo));
        }
        const double Sd = build_cdf();
        const double lse = m + o_log(Sd);
        last_ll = lse - prev;
        loglik += last_ll;
        const bool resample_now = ((t + 1) % rs == 0);
        prev = resample_now ? o_log((double)N) : lse;
        ++t;
        return last_ll;
    }

    double step(double y, double zcov) {
        const bool resampled_prev = (t > 0) && (t % rs == 0);
        std::vector<double> lw_old(N, 0.0);
        if (t == 0) {
            for (int i = 0; i < N; ++i) x[i] = m_init(mc, normal(i, 0));
        } else {
            xprev.swap(x);
            if (resampled_prev) {
                std::vector<uint64_t> tau; targets(t, tau);
                for (int i = 0; i < N; ++i) anc[i] = (uint32_t)search(tau[i]);
                for (int i = 0; i < N; ++i) x[i] = m_prop(mc, xprev[anc[i]], normal(i, t), zcov);
            } else {
                for (int i = 0; i < N; ++i) { lw_old[i] = logw[i]; x[i] = m_prop(mc, xprev[i], normal(i, t), zcov); }
            }
        }
        for (int i = 0; i < N; ++i) logw[i] = lw_old[i] + m_logg(mc, y, x[i]);
        const double Sd = build_cdf();
        const double lse = m + o_log(Sd);
        last_ll = lse - prev;
        loglik += last_ll;
        const bool resample_now = ((t + 1) % rs == 0);
        prev = resample_now ? o_log((double)N) : lse;
        ++t;
        return last_ll;
    }
};


// ---------------------------------------------------------------------------------------
// Mode A: reference-faithful scalar filter (mt19937, <random>), templated on float type.
// Follows liu_west_filter.h:1608-1761 (SISR step) with the model callbacks of
// univ_svol_bootstrap_filter.h:55-103 / test_pswarm.cpp:80-134; resampler =
// std::discrete_distribution multinomial [pf-recollection of pf::resamplers::mn_resampler]
// or the in-tree sorted-uniform resampler (liu_west_filter.h:91-145) when fast != 0.
// Heap storage (the reference's std::array-on-stack pattern overflows at N >= 2^18).
// ---------------------------------------------------------------------------------------
template <typename F>
F evalUnivNormLog(F x, F mu, F sigma) {     // pf::rveval::evalUnivNorm(...,true) [pf-recollection]
    if (sigma > F(0)) {
        const F d = (x - mu) / sigma;
        return -std::log(sigma) - F(HALF_LOG_2PI) - F(0.5) * d * d;
    }
    return -std::numeric_limits<F>::infinity();
}

template <typename F>
double ref_run(int model, const double* th, int N, const double* y, const double* zc, int T,
               uint32_t seed, int fast, double* per_step) {
    std::mt19937 gen(seed), rgen(seed ^ 0x9E3779B9u);
    std::normal_distribution<F> nd(F(0), F(1));
    std::uniform_real_distribution<F> ud(F(0), F(1));
    std::vector<F> x(N), xn(N), lw(N, F(0)), w(N);
    F loglik = 0;
    const F p0 = (F)th[0], p1 = (F)th[1], p2 = (F)th[2], p3 = (F)(model == MODEL_SVOL_LEVERAGE ? th[3] : 0.0);
    for (int t = 0; t < T; ++t) {
        const F yt = (F)y[t];
        const F zt = zc ? (F)zc[t] : F(0);
        std::vector<F> old = lw;
        F mold = *std::max_element(old.begin(), old.end());
        for (int i = 0; i < N; ++i) {
            F xs, lg;
            if (model == MODEL_SVOL) {               // th = beta, phi, sigma
                if (t == 0) xs = nd(gen) * p2 / std::sqrt(F(1) - p1 * p1);
                else        xs = p1 * x[i] + nd(gen) * p2;
                lg = evalUnivNormLog<F>(yt, F(0), p0 * std::exp(F(0.5) * xs));
            } else if (model == MODEL_SVOL_LEVERAGE) {   // th = phi, mu, sigma, rho
                if (t == 0) xs = nd(gen) * p2 / std::sqrt(F(1) - p0 * p0);
                else {
                    const F mean = p1 + p0 * (x[i] - p1) + p3 * p2 * zt * std::exp(F(-0.5) * x[i]);
                    xs = mean + nd(gen) * p2 * std::sqrt(F(1) - p0 * p0);
                }
                lg = evalUnivNormLog<F>(yt, F(0), std::exp(F(0.5) * xs));
            } else {                                 // th = phi, sigma, tau
                if (t == 0) xs = nd(gen) * p1 / std::sqrt(F(1) - p0 * p0);
                else        xs = p0 * x[i] + nd(gen) * p1;
                lg = evalUnivNormLog<F>(yt, xs, p2);
            }
            x[i] = xs;
            lw[i] += lg;          // t == 0: logMu - logQ1 cancel identically for these models
        }
        const F mx = *std::max_element(lw.begin(), lw.end());
        F s1 = 0, s2 = 0;
        for (int i = 0; i < N; ++i) { w[i] = std::exp(lw[i] - mx); s1 += w[i]; s2 += std::exp(old[i] - mold); }
        const F ll = (t == 0) ? (-std::log((F)N) + mx + std::log(s1))
                              : (mx + std::log(s1) - mold - std::log(s2));
        if (per_step) per_step[t] = (double)ll;
        loglik += ll;
        // resample every step (default schedule)
        if (!fast) {
            std::discrete_distribution<int> dd(w.begin(), w.end());
            for (int i = 0; i < N; ++i) xn[i] = x[dd(rgen)];
        } else {
            std::vector<F> E(N);
            F G = 0;
            for (int i = 0; i < N; ++i) { E[i] = -std::log(ud(rgen)); G += E[i]; }
            G -= std::log(ud(rgen));
            F uos = 0, run = w[0] / s1, less = 0;
            int idx = 0;
            for (int i = 0; i < N; ++i) {
                uos += E[i] / G;
                while (!((less < uos) && (uos <= run)) && idx < N - 1) {
                    ++idx; run += w[idx] / s1; less += w[idx - 1] / s1;
                }
                xn[i] = x[idx];
            }
        }
        x.swap(xn);
        std::fill(lw.begin(), lw.end(), F(0));
    }
    return (double)loglik;
}

// ---------------------------------------------------------------------------------------
// Liu-West filter with covariates, auxiliary-particle form: LWFilterWithCovs::filter,
// include/ssme/liu_west_filter.h:971-1159; proposal components :1184-1198; shrinkage a = (3d-1)/(2d) :960;
// model svol_lw_1_par, test/test_liu_west.cpp:82-157 (parameters phi, mu, sigma, rho with transforms
// logit, null, log, twice_fisher :70; uniform priors :150-157); transforms include/ssme/parameters.h:317-457.
// Resampling every step (the reference's default schedule rs = 1).
// ---------------------------------------------------------------------------------------
constexpr int DP = 4;
enum { TR_NULL = 0, TR_TWICE_FISHER = 1, TR_LOGIT = 2, TR_LOG = 3 };     // enum order of parameters.h:27
enum { STREAM_LW_PRIOR = 3 /* and 4 */, STREAM_LW_JIT = 5 /* and 6 */, STREAM_LW_K = 7, STREAM_LW_K_EXTRA = 8,
       STREAM_GAMMA_K = 80 };

// kernel-matched transforms (libm-free)
inline double tr_inv(int kind, double tp) {                 // parameters.h inv_trans
    switch (kind) {
        case TR_NULL: return tp;
        case TR_TWICE_FISHER: return (tp >= 0.0) ? 2.0 / (1.0 + o_exp_t(-tp)) - 1.0 : 1.0 - 2.0 / (1.0 + o_exp_t(tp));
        case TR_LOGIT: return (tp >= 0.0) ? 1.0 / (1.0 + o_exp_t(-tp)) : o_exp_t(tp) / (1.0 + o_exp_t(tp));
        default: return o_exp_t(tp);
    }
}
inline double tr_fwd(int kind, double p) {                  // parameters.h trans
    switch (kind) {
        case TR_NULL: return p;
        case TR_TWICE_FISHER: return o_log(1.0 + p) - o_log(1.0 - p);
        case TR_LOGIT: return o_log(p) - o_log(1.0 - p);
        default: return o_log(p);
    }
}

// model callbacks of svol_lw_1_par (kernel form of logGEv as in m_logg with beta = 1)
inline double lw_logg(double y, double x) {
    const double hl = 0.5 * x;
    if (hl < -745.1332191019412) return NEG_INF;
    return (-hl - HALF_LOG_2PI) - 0.5 * ((y * y) * o_exp_t(-x));
}
inline double lw_propmu(double x, double z, const double* tu) {           // test_liu_west.cpp:93-101
    double xt = tu[1] + tu[0] * (x - tu[1]);
    xt = xt + ((z * tu[3]) * tu[2]) * o_exp_t(-0.5 * x);
    return xt;
}

// the DPP reduction tree of one wave (64 lanes): returns what lane 63 holds after the inclusive scan
inline double wave_tree_sum(const double* v64) {
    double v[64];
    std::memcpy(v, v64, sizeof(v));
    for (int d = 1; d <= 8; d <<= 1) {
        double n[64];
        for (int l = 0; l < 64; ++l) n[l] = v[l] + (((l & 15) >= d) ? v[l - d] : 0.0);
        std::memcpy(v, n, sizeof(v));
    }
    {   double n[64];
        for (int l = 0; l < 64; ++l) { const int row = l >> 4; n[l] = v[l] + ((row == 1 || row == 3) ? v[row * 16 - 1] : 0.0); }
        std::memcpy(v, n, sizeof(v)); }
    {   double n[64];
        for (int l = 0; l < 64; ++l) n[l] = v[l] + ((l >= 32) ? v[31] : 0.0);
        std::memcpy(v, n, sizeof(v)); }
    return v[63];
}
// canonical sum of one tile (2048 values): fold the upper half onto the lower (u_j = v_j + v_{j+1024}), pair sums,
// wave tree per 128-element segment of u, the 8 segments in order
inline double tile_tree_sum(const double* v) {
    double tot = 0.0;
    for (int sgm = 0; sgm < 8; ++sgm) {
        double lanes[64];
        for (int l = 0; l < 64; ++l) {
            const int j = sgm * 128 + 2 * l;
            lanes[l] = (v[j] + v[j + 1024]) + (v[j + 1] + v[j + 1025]);
        }
        tot = tot + wave_tree_sum(lanes);
    }
    return tot;
}
// canonical sum over tiles: 64 lanes each add a contiguous chunk of tiles in order, then the wave tree
inline double tiles_tree_sum(const double* part, int B) {
    const int c = (B + 63) / 64;
    double lanes[64];
    for (int l = 0; l < 64; ++l) { double a = 0.0; for (int j = l * c; j < (l + 1) * c && j < B; ++j) a = a + part[j]; lanes[l] = a; }
    return wave_tree_sum(lanes);
}

struct LWFilter {
    Filter cdfA, cdfB;                 // reuse the exact-cdf machinery: A = first-stage weights, B = second-stage
    int N, Npad, B, t;
    uint32_t key[2], rep;
    int trans[DP];
    double lo[DP], hi[DP], a_shrink;
    std::vector<double> x, th[DP], xr, thr[DP], lw1;     // (x, theta) after stage 2; resampled population; first-stage log-weights
    std::vector<uint32_t> anc, kidx;
    double thetabar[DP], L[DP][DP];
    double loglik, last_ll, lse1;
    // form 0: auxiliary-particle form (LWFilterWithCovs::filter, :971-1159); form 1: SISR form (LWFilter2WithCovs::filter,
    // :2191-2343, model svol_lw_2_par: the proposal is the transition, logFEv - logQEv = 0).  rs = m_rs: resample when
    // (t + 1) % rs == 0 (:1139-1140, :2317-2318); prev = log-sum-exp of the weights the step starts from.
    int form = 0, rs = 1;
    double prev = 0.0;
    std::vector<double> lwB;           // second-stage log-weights of the last step (carried when no resampling follows)

    void init(int N_, uint64_t seed, uint32_t rep_, const int* tr, const double* lo_, const double* hi_, double delta,
              int form_ = 0, int rs_ = 1) {
        form = form_; rs = rs_ < 1 ? 1 : rs_;
        N = N_; rep = rep_;
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
        double dummy[3] = {1.0, 0.5, 0.1};
        // exact-cdf machinery with the table exp; spacings = words 2-3 of ONE call per particle pair: the resampling draw's
        // counter stream is STREAM_RESAMP, the k draw's STREAM_LW_K (whose words 0-1 are the pair's state normals of fSamp)
        cdfA.bootstrap_draws = true; cdfA.pair_stream = STREAM_LW_K;
        cdfB.bootstrap_draws = true; cdfB.pair_stream = STREAM_RESAMP;
        cdfA.init(MODEL_SVOL, N, RESAMP_MULTINOMIAL, 1, seed, rep, dummy);
        cdfB.init(MODEL_SVOL, N, RESAMP_MULTINOMIAL, 1, seed, rep, dummy);
        B = cdfA.B; Npad = cdfA.Npad;
        for (int d = 0; d < DP; ++d) { trans[d] = tr[d]; lo[d] = lo_[d]; hi[d] = hi_[d]; th[d].assign(Npad, 0.0); thr[d].assign(Npad, 0.0); }
        x.assign(Npad, 0.0); xr.assign(Npad, 0.0); lw1.assign(Npad, 0.0); anc.assign(Npad, 0); kidx.assign(Npad, 0);
        lwB.assign(Npad, 0.0);
        a_shrink = (3.0 * delta - 1.0) / (2.0 * delta);
        t = 0; loglik = 0.0; last_ll = 0.0; lse1 = 0.0; prev = o_log((double)N);
        for (int d = 0; d < DP; ++d) { thetabar[d] = 0.0; for (int e = 0; e < DP; ++e) L[d][e] = 0.0; }
    }
    void words(int idx, int tt, int stream, uint32_t o[4]) const {
        const uint32_t ctr[4] = {(uint32_t)idx, (uint32_t)tt, rep, (uint32_t)stream};
        philox4x32_10(ctr, key, o);
    }
    void normal2(int idx, int tt, int stream, double* z0, double* z1) const {
        uint32_t o[4]; words(idx, tt, stream, o);
        const double rad = std::sqrt(-2.0 * o_log(u01_oc(o[0], o[1])));
        double sn, cs; o_sincos2pi(u01_co(o[2], o[3]), &sn, &cs);
        *z0 = rad * cs; *z1 = rad * sn;
    }
    // t = 0 (k_lw_init): 52-bit Box-Muller on (pair, 0, filter, STREAM_PROP).  t >= 1: words 0-1 of the pair's STREAM_LW_K call
    double state_normal(int i, int tt) const {
        if (tt > 0) return cdfA.normal(i, tt);
        double z0, z1; normal2(i >> 1, tt, STREAM_PROP, &z0, &z1); return (i & 1) ? z1 : z0;
    }
    // the four jitter normals of particle i from ONE call: Box-Muller on (words 0-1) and on (words 2-3), 40-bit radius
    // uniform and 24-bit angle each, table log
    void jitter4(int i, int tt, double* e) const {
        uint32_t o[4]; words(i, tt, STREAM_LW_JIT, o);
        for (int h = 0; h < 2; ++h) {
            const double rad = std::sqrt(-2.0 * o_log_u(u01_mid40(o[2 * h], o[2 * h + 1])));
            double sn, cs; o_sincos_k24(o[2 * h + 1], &sn, &cs);
            e[2 * h] = rad * cs; e[2 * h + 1] = rad * sn;
        }
    }

    double step(double y, double z) {
        const double logN = o_log((double)N);
        if (t == 0) {
            // :1103-1122  prior draws, q1Samp, weights (logMuEv - logQ1Ev cancel identically)
            for (int i = 0; i < N; ++i) {
                uint32_t o1[4], o2[4]; words(i, 0, STREAM_LW_PRIOR, o1); words(i, 0, STREAM_LW_PRIOR + 1, o2);
                const double u[DP] = {u01_co(o1[0], o1[1]), u01_co(o1[2], o1[3]), u01_co(o2[0], o2[1]), u01_co(o2[2], o2[3])};
                double tu[DP];
                for (int d = 0; d < DP; ++d) { tu[d] = lo[d] + u[d] * (hi[d] - lo[d]); th[d][i] = tr_fwd(trans[d], tu[d]); }
                x[i] = state_normal(i, 0) * (tu[2] / std::sqrt(1.0 - tu[0] * tu[0]));
                cdfB.logw[i] = lw_logg(y, x[i]);
            }
            const double Sd = cdfB.build_cdf();
            const double lseB = cdfB.m + o_log(Sd);
            last_ll = lseB - prev;                     // prev = log N
            for (int i = 0; i < N; ++i) lwB[i] = cdfB.logw[i];
            prev = ((t + 1) % rs == 0) ? logN : lseB;
        } else {
            // ---- stage 1: resample (x, theta) by the previous second-stage weights (:91-145 via the exact cdf) if the
            //      schedule resampled at the end of step t-1; otherwise the population stays and its weights are carried
            const bool resampled = (t % rs) == 0;
            std::vector<uint64_t> tau;
            if (resampled) {
                cdfB.targets(t, tau);
                for (int i = 0; i < N; ++i) anc[i] = (uint32_t)cdfB.search(tau[i]);
            } else {
                for (int i = 0; i < N; ++i) anc[i] = (uint32_t)i;
            }
            std::vector<double> mom[14];
            for (auto& v : mom) v.assign(Npad, 0.0);
            for (int i = 0; i < N; ++i) {
                xr[i] = x[anc[i]];
                double tt[DP], tu[DP];
                for (int d = 0; d < DP; ++d) { tt[d] = th[d][anc[i]]; thr[d][i] = tt[d]; tu[d] = tr_inv(trans[d], tt[d]); }
                const double lw_old = resampled ? 0.0 : lwB[i];
                if (form == 0) {
                    // first-stage weight :985-991 = carried weight + logG(y | propMu) (logGEv ignores its parameter argument
                    // in this model); lw1 keeps the logG part alone, which is what stage 2 subtracts (:1041-1043)
                    lw1[i] = lw_logg(y, lw_propmu(xr[i], z, tu));
                    cdfA.logw[i] = lw_old + lw1[i];
                } else {
                    lw1[i] = lw_old;                   // SISR form: no first stage; the carried weight goes to stage 2
                }
                int q = 0;
                for (int d = 0; d < DP; ++d) mom[q++][i] = tt[d];
                for (int d = 0; d < DP; ++d) for (int e = 0; e <= d; ++e) mom[q++][i] = tt[d] * tt[e];
            }
            if (form == 0) {
                const double S1 = cdfA.build_cdf();
                lse1 = cdfA.m + o_log(S1);
            }
            // ---- proposal components :1184-1198 (theta-bar, V over the resampled population), Cholesky of (1-a^2) V
            double sums[14];
            for (int q = 0; q < 14; ++q) {
                std::vector<double> part(B);
                for (int b = 0; b < B; ++b) part[b] = tile_tree_sum(&mom[q][(size_t)b * TILE]);
                sums[q] = tiles_tree_sum(part.data(), B);
            }
            const double invN = 1.0 / (double)N;
            for (int d = 0; d < DP; ++d) thetabar[d] = sums[d] * invN;
            double Sig[DP][DP];
            const double h2 = 1.0 - a_shrink * a_shrink;
            { int q = DP; for (int d = 0; d < DP; ++d) for (int e = 0; e <= d; ++e) { Sig[d][e] = h2 * (sums[q++] * invN - thetabar[d] * thetabar[e]); } }
            for (int d = 0; d < DP; ++d) for (int e = 0; e < DP; ++e) L[d][e] = 0.0;
            for (int j = 0; j < DP; ++j) {
                double sdiag = Sig[j][j];
                for (int k = 0; k < j; ++k) sdiag = sdiag - L[j][k] * L[j][k];
                L[j][j] = (sdiag > 0.0) ? std::sqrt(sdiag) : 0.0;
                for (int i = j + 1; i < DP; ++i) {
                    double v = Sig[i][j];
                    for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
                    L[i][j] = (L[j][j] > 0.0) ? v / L[j][j] : 0.0;
                }
            }
            // ---- stage 2: k ~ Categorical(first-stage weights) :1006, jitter :1024-1027, fSamp, second-stage weight :1030-1033
            if (form == 0) {
                cdfA.targets(t, tau, STREAM_LW_K, STREAM_LW_K_EXTRA, STREAM_GAMMA_K);
                for (int i = 0; i < N; ++i) kidx[i] = (uint32_t)cdfA.search(tau[i]);
            } else {
                for (int i = 0; i < N; ++i) kidx[i] = (uint32_t)i;       // :2206-2235 every particle continues itself
            }
            for (int i = 0; i < N; ++i) {
                const int k = (int)kidx[i];
                double e[DP];
                jitter4(i, t, e);
                double tn[DP], tu[DP];
                for (int d = 0; d < DP; ++d) {
                    const double mm = a_shrink * thr[d][k] + (1.0 - a_shrink) * thetabar[d];
                    double acc = 0.0;
                    for (int q = 0; q <= d; ++q) acc = acc + L[d][q] * e[q];
                    tn[d] = mm + acc;
                    tu[d] = tr_inv(trans[d], tn[d]);
                }
                const double xk = xr[k];
                const double mean = (tu[1] + tu[0] * (xk - tu[1])) + ((z * tu[3]) * tu[2]) * o_exp_t(-0.5 * xk);     // fSamp :114-121
                const double xn = mean + state_normal(i, t) * (tu[2] * std::sqrt(1.0 - tu[3] * tu[3]));
                x[i] = xn;
                for (int d = 0; d < DP; ++d) th[d][i] = tn[d];
                cdfB.logw[i] = (form == 0) ? lw_logg(y, xn) - lw1[k] : lw1[k] + lw_logg(y, xn);
            }
            const double S2 = cdfB.build_cdf();
            const double lseB = cdfB.m + o_log(S2);
            // form 0, :1047: m1 + log(sum1) + m2 + log(sum2) - 2 m3 - 2 log(sum3); form 1, :2257-2264: lse(new) - lse(old)
            last_ll = (form == 0) ? (lseB + lse1) - 2.0 * prev : lseB - prev;
            for (int i = 0; i < N; ++i) lwB[i] = cdfB.logw[i];
            prev = ((t + 1) % rs == 0) ? logN : lseB;
        }
        loglik += last_ll;
        ++t;
        return last_ll;
    }
    // E[h] under the current second-stage weights: ids 0-3 = x, x^2, exp(x/2), 42; 4-7 = untransformed phi, mu, sigma, rho
    double expectation(int id) const {
        double den = 0.0, num = 0.0;
        for (int i = 0; i < N; ++i) {
            const double w = o_exp(cdfB.logw[i] - cdfB.m);
            const double hv = id == 0 ? x[i] : id == 1 ? x[i] * x[i] : id == 2 ? o_exp(0.5 * x[i]) : id == 3 ? 42.0 : tr_inv(trans[id - 4], th[id - 4][i]);
            den += w; num += w * hv;
        }
        return num / den;
    }
    // weighted mean of the untransformed parameters under the current second-stage weights
    void param_means(double* out) const {
        double den = 0.0, num[DP] = {0, 0, 0, 0};
        for (int i = 0; i < N; ++i) {
            const double w = o_exp(cdfB.logw[i] - cdfB.m);
            den += w;
            for (int d = 0; d < DP; ++d) num[d] += w * tr_inv(trans[d], th[d][i]);
        }
        for (int d = 0; d < DP; ++d) out[d] = num[d] / den;
    }
};

// Mode A: reference-faithful Liu-West (mt19937, <random>), double.  Follows liu_west_filter.h:971-1159 step by step,
// including the quirk at :988 (untransformed theta mixed with transformed theta-bar; harmless: logGEv ignores it).
// MVNSampler is restated with a Cholesky factor (pf uses an eigen-decomposition square root [pf-recollection];
// any square root gives the same law); k_gen = discrete_distribution on max-subtracted weights [pf-recollection].
double lw_ref_run(int N, const int* tr, const double* lo, const double* hi, double delta, const double* y, const double* z, int T,
                  uint32_t seed, double* per_step, double* means_out, int form = 0, int rs = 1) {
    if (rs < 1) rs = 1;
    std::mt19937 gen(seed), rgen(seed ^ 0x9E3779B9u), kgen(seed ^ 0x85EBCA6Bu);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    const double a = (3.0 * delta - 1.0) / (2.0 * delta);
    auto inv = [&](int kind, double tp) {
        switch (kind) {
            case TR_NULL: return tp;
            case TR_TWICE_FISHER: return (tp >= 0.0) ? 2.0 / (1.0 + std::exp(-tp)) - 1.0 : 1.0 - 2.0 / (1.0 + std::exp(tp));
            case TR_LOGIT: return (tp >= 0.0) ? 1.0 / (1.0 + std::exp(-tp)) : std::exp(tp) / (1.0 + std::exp(tp));
            default: return std::exp(tp);
        }
    };
    auto fwd = [&](int kind, double p) {
        switch (kind) {
            case TR_NULL: return p;
            case TR_TWICE_FISHER: return std::log(1.0 + p) - std::log(1.0 - p);
            case TR_LOGIT: return std::log(p) - std::log(1.0 - p);
            default: return std::log(p);
        }
    };
    auto logg = [&](double yy, double xx) { return evalUnivNormLog<double>(yy, 0.0, std::exp(0.5 * xx)); };
    std::vector<double> x(N), lw(N, 0.0);
    std::vector<std::array<double, DP>> th(N);
    double loglik = 0.0;
    for (int t = 0; t < T; ++t) {
        const double yt = y[t], zt = z ? z[t] : 0.0;
        double ll;
        if (t == 0) {
            for (int i = 0; i < N; ++i) {
                double tu[DP];
                for (int d = 0; d < DP; ++d) { tu[d] = lo[d] + ud(gen) * (hi[d] - lo[d]); th[i][d] = fwd(tr[d], tu[d]); }
                x[i] = nd(gen) * tu[2] / std::sqrt(1.0 - tu[0] * tu[0]);
                lw[i] = logg(yt, x[i]);
            }
            const double mx = *std::max_element(lw.begin(), lw.end());
            double se = 0; for (int i = 0; i < N; ++i) se += std::exp(lw[i] - mx);
            ll = -std::log((double)N) + mx + std::log(se);
        } else {
            double tb[DP] = {0, 0, 0, 0}, V[DP][DP] = {};
            for (int i = 0; i < N; ++i) for (int d = 0; d < DP; ++d) { tb[d] += th[i][d] / N; for (int e = 0; e < DP; ++e) V[d][e] += th[i][d] * th[i][e] / N; }
            double Lc[DP][DP] = {};
            const double h2 = 1.0 - a * a;
            for (int d = 0; d < DP; ++d) for (int e = 0; e < DP; ++e) V[d][e] = h2 * (V[d][e] - tb[d] * tb[e]);
            for (int j = 0; j < DP; ++j) {
                double sd = V[j][j]; for (int k = 0; k < j; ++k) sd -= Lc[j][k] * Lc[j][k];
                Lc[j][j] = sd > 0 ? std::sqrt(sd) : 0.0;
                for (int i = j + 1; i < DP; ++i) { double v = V[i][j]; for (int k = 0; k < j; ++k) v -= Lc[i][k] * Lc[j][k]; Lc[i][j] = Lc[j][j] > 0 ? v / Lc[j][j] : 0.0; }
            }
            if (form == 0) {
            std::vector<double> lw1(N), w1(N);
            double m3 = -INFINITY, m2 = -INFINITY;
            for (int i = 0; i < N; ++i) {
                if (lw[i] > m3) m3 = lw[i];
                double tu[DP]; for (int d = 0; d < DP; ++d) tu[d] = inv(tr[d], th[i][d]);
                const double mu = tu[1] + tu[0] * (x[i] - tu[1]) + zt * tu[3] * tu[2] * std::exp(-0.5 * x[i]);
                lw1[i] = lw[i] + logg(yt, mu);
                if (lw1[i] > m2) m2 = lw1[i];
            }
            for (int i = 0; i < N; ++i) w1[i] = std::exp(lw1[i] - m2);
            std::discrete_distribution<int> kd(w1.begin(), w1.end());
            std::vector<double> xo = x; auto tho = th; std::vector<double> lwo = lw;
            double m1 = -INFINITY, s1 = 0, s2 = 0, s3 = 0;
            for (int i = 0; i < N; ++i) {
                s2 += std::exp(lw1[i] - m2); s3 += std::exp(lwo[i] - m3);
                const int k = kd(kgen);
                double tn[DP], tu[DP], tuo[DP], e[DP];
                for (int d = 0; d < DP; ++d) e[d] = nd(gen);
                for (int d = 0; d < DP; ++d) { double acc = 0; for (int q = 0; q <= d; ++q) acc += Lc[d][q] * e[q]; tn[d] = a * tho[k][d] + (1.0 - a) * tb[d] + acc; tu[d] = inv(tr[d], tn[d]); tuo[d] = inv(tr[d], tho[k][d]); }
                const double xk = xo[k];
                const double xn = tu[1] + tu[0] * (xk - tu[1]) + zt * tu[3] * tu[2] * std::exp(-0.5 * xk) + nd(gen) * tu[2] * std::sqrt(1.0 - tu[3] * tu[3]);
                const double muk = tuo[1] + tuo[0] * (xk - tuo[1]) + zt * tuo[3] * tuo[2] * std::exp(-0.5 * xk);
                x[i] = xn; for (int d = 0; d < DP; ++d) th[i][d] = tn[d];
                lw[i] = logg(yt, xn) - logg(yt, muk);
                if (lw[i] > m1) m1 = lw[i];
            }
            for (int i = 0; i < N; ++i) s1 += std::exp(lw[i] - m1);
            ll = m1 + std::log(s1) + m2 + std::log(s2) - 2 * m3 - 2 * std::log(s3);
            } else {
            // LWFilter2WithCovs::filter :2196-2264 with svol_lw_2_par (test/test_liu_west.cpp:274-336): jitter, qSamp (= the
            // transition), weight += logFEv + logGEv - logQEv added and subtracted literally as the reference does
            std::vector<double> old = lw;
            const double mold = *std::max_element(old.begin(), old.end());
            for (int i = 0; i < N; ++i) {
                double tn[DP], tu[DP], e[DP];
                for (int d = 0; d < DP; ++d) e[d] = nd(gen);
                for (int d = 0; d < DP; ++d) { double acc = 0; for (int q = 0; q <= d; ++q) acc += Lc[d][q] * e[q]; tn[d] = a * th[i][d] + (1.0 - a) * tb[d] + acc; tu[d] = inv(tr[d], tn[d]); }
                const double mean = tu[1] + tu[0] * (x[i] - tu[1]) + zt * tu[3] * tu[2] * std::exp(-0.5 * x[i]);
                const double sd = tu[2] * std::sqrt(1.0 - tu[3] * tu[3]);
                const double xn = mean + nd(gen) * sd;
                lw[i] += evalUnivNormLog<double>(xn, mean, sd);
                lw[i] += logg(yt, xn);
                lw[i] -= evalUnivNormLog<double>(xn, mean, sd);
                x[i] = xn; for (int d = 0; d < DP; ++d) th[i][d] = tn[d];
            }
            const double mx = *std::max_element(lw.begin(), lw.end());
            double s1 = 0, s2 = 0;
            for (int i = 0; i < N; ++i) { s1 += std::exp(lw[i] - mx); s2 += std::exp(old[i] - mold); }
            ll = mx + std::log(s1) - mold - std::log(s2);
            }
        }
        if (per_step) per_step[t] = ll;
        loglik += ll;
        if (t == T - 1 && means_out) {
            const double mx = *std::max_element(lw.begin(), lw.end());
            double den = 0, num[DP] = {0, 0, 0, 0};
            for (int i = 0; i < N; ++i) { const double w = std::exp(lw[i] - mx); den += w; for (int d = 0; d < DP; ++d) num[d] += w * inv(tr[d], th[i][d]); }
            for (int d = 0; d < DP; ++d) means_out[d] = num[d] / den;
        }
        // resample states and parameters (:91-145), weights reset -- when the schedule says so (:1139-1140, :2317-2318)
        if ((t + 1) % rs == 0) {
            const double mx = *std::max_element(lw.begin(), lw.end());
            std::vector<double> w(N); for (int i = 0; i < N; ++i) w[i] = std::exp(lw[i] - mx);
            std::discrete_distribution<int> dd(w.begin(), w.end());
            std::vector<double> xn(N); auto tn = th;
            for (int i = 0; i < N; ++i) { const int j = dd(rgen); xn[i] = x[j]; tn[i] = th[j]; }
            x.swap(xn); th.swap(tn); std::fill(lw.begin(), lw.end(), 0.0);
        }
    }
    return loglik;
}

}  // namespace

// =======================================================================================
// C entry points (ctypes)
// =======================================================================================
extern "C" {

void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
void orc_exp(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_exp(x[i]); }
void orc_log(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_log(x[i]); }
void orc_sincos2pi(const double* u, double* s, double* c, long n) { for (long i = 0; i < n; ++i) o_sincos2pi(u[i], s + i, c + i); }
void orc_normals(uint64_t seed, uint32_t rep, int t, int n, double* out) {
    Filter f; double th[3] = {1.0, 0.5, 0.1}; f.bootstrap_draws = true; f.init(MODEL_SVOL, n, 0, 1, seed, rep, th);
    for (int i = 0; i < n; ++i) out[i] = f.normal(i, t);
}
void orc_log_u(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_log_u(x[i]); }
void orc_log_u32(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_log_u32(x[i]); }
void orc_sincos_k24(const double* k, double* s, double* c, long n) { for (long i = 0; i < n; ++i) o_sincos_k24((uint32_t)k[i], s + i, c + i); }
void orc_exp_t(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_exp_t(x[i]); }
void orc_exp_scaled_t(const double* x, int sc, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_exp_scaled_t(x[i], sc); }
void orc_exp_scaled(const double* x, int sc, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = o_exp_scaled(x[i], sc); }
// Gamma(shape) draws for tiles b = 0..n-1 at time t (Marsaglia-Tsang, counter driven)
void orc_gamma(uint64_t seed, uint32_t rep, int t, double shape, int n, double* out) {
    Filter f; double th[3] = {1.0, 0.5, 0.1}; f.init(MODEL_SVOL, 2048, 0, 1, seed, rep, th);
    for (int b = 0; b < n; ++b) out[b] = f.gamma_draw(b, t, shape);
}
// fixed-point quantisation of weights: q = rne(exp(x) * 2^sc) for x <= 0
// A'_b = rint((double)A * exp(dm) * 2^sc)
void orc_rescale(const uint64_t* A, const double* dm, int sc, uint64_t* out, long n) { for (long i = 0; i < n; ++i) out[i] = rint_to_u64((double)A[i] * o_exp_scaled_t(dm[i], sc)); }
void orc_quantize(const double* x, int sc, uint64_t* q, long n) { for (long i = 0; i < n; ++i) q[i] = rne_u64(o_exp_scaled_t(x[i], sc)); }

void* orc_pf_create(int model, int N, int resamp, int rs, uint64_t seed, uint32_t rep, const double* theta, int tile) {
    Filter* f = new Filter(); f->bootstrap_draws = true; f->tile = tile; f->init(model, N, resamp, rs, seed, rep, theta); return f;
}
// a filter whose model is given by callbacks (MODEL_USER0): init_sd = the sd of the t = 0 draw x_0 = init_sd * z; bad: logG = -inf
void* orc_pf_create_user(int N, int resamp, int rs, uint64_t seed, uint32_t rep, double init_sd, int bad, user_prop_fn prop, user_logg_fn logg, int tile) {
    Filter* f = new Filter(); f->bootstrap_draws = true; f->tile = tile;
    const double th[3] = {1.0, 0.5, 0.1};                  // placeholder for init(): the constants are replaced below
    f->init(MODEL_SVOL, N, resamp, rs, seed, rep, th);
    f->model = MODEL_USER0;
    f->mc = ModelConst{};
    f->mc.model = MODEL_USER0; f->mc.a2 = init_sd; f->mc.bad = bad; f->mc.u_prop = prop; f->mc.u_logg = logg;
    return f;
}
// ... with a vector state / observation: dx, dy <= 4; y of orc_pf_run_series_vec is T rows of dy values
void* orc_pf_create_user_vec(int N, int resamp, int rs, uint64_t seed, uint32_t rep, int dx, int dy, int bad, user_initv_fn init, user_propv_fn prop,
                             user_loggv_fn logg, int tile) {
    Filter* f = new Filter(); f->bootstrap_draws = true; f->tile = tile;
    const double th[3] = {1.0, 0.5, 0.1};
    f->init(MODEL_SVOL, N, resamp, rs, seed, rep, th);
    f->model = MODEL_USER0;
    f->mc = ModelConst{};
    f->mc.model = MODEL_USER0; f->mc.bad = bad; f->mc.dx = dx; f->mc.dy = dy; f->mc.v_init = init; f->mc.v_prop = prop; f->mc.v_logg = logg;
    return f;
}
double orc_pf_run_series_vec(void* h, const double* y, const double* z, int T, double* per_step) {
    Filter* f = (Filter*)h; f->reset();
    for (int t = 0; t < T; ++t) { const double l = f->step_vec(y + (size_t)t * f->mc.dy, z ? z[t] : 0.0); if (per_step) per_step[t] = l; }
    return f->loglik;
}
// component d (1 .. dx-1) of the particles after the last step (component 0: orc_pf_state)
void orc_pf_state_dim(void* h, int d, double* x) { Filter* f = (Filter*)h; std::memcpy(x, f->xv[d].data(), sizeof(double) * f->N); }
void orc_pf_destroy(void* h) { delete (Filter*)h; }
void orc_pf_reset(void* h) { ((Filter*)h)->reset(); }
double orc_pf_step(void* h, double y, double z) { return ((Filter*)h)->step(y, z); }
double orc_pf_loglik(void* h) { return ((Filter*)h)->loglik; }
double orc_pf_run_series(void* h, const double* y, const double* z, int T, double* per_step) {
    Filter* f = (Filter*)h; f->reset();
    for (int t = 0; t < T; ++t) { const double l = f->step(y[t], z ? z[t] : 0.0); if (per_step) per_step[t] = l; }
    return f->loglik;
}
// state after the last step: particles (pre-resampling), log-weights, tile-local integer cdf,
// ancestors used by the last step, integer tile sums, scalars {m, (double)S_int, r}
void orc_pf_state(void* h, double* x, double* logw, uint64_t* loc, uint32_t* anc, uint64_t* A, double* mb, double* scal) {
    Filter* f = (Filter*)h;
    if (x) std::memcpy(x, f->x.data(), sizeof(double) * f->N);
    if (logw) std::memcpy(logw, f->logw.data(), sizeof(double) * f->N);
    if (loc) std::memcpy(loc, f->loc.data(), sizeof(uint64_t) * f->N);
    if (anc) std::memcpy(anc, f->anc.data(), sizeof(uint32_t) * f->N);
    if (A) std::memcpy(A, f->A.data(), sizeof(uint64_t) * f->B);
    if (mb) std::memcpy(mb, f->mb.data(), sizeof(double) * f->B);
    if (scal) { scal[0] = f->m; scal[1] = (double)f->Sint; scal[2] = (double)f->rshift; }
}
uint64_t orc_pf_sum_int(void* h) { return ((Filter*)h)->Sint; }
// weighted expectation of a built-in functional, pre-resampling (liu_west_filter.h:1662-1683)
// kind: 0 = x, 1 = x^2, 2 = exp(x/2) (volatility), 3 = constant 42 (test_pswarm.cpp:252 KAT)
double orc_pf_expectation(void* h, int kind) {
    Filter* f = (Filter*)h;
    double num = 0, den = 0;
    for (int i = 0; i < f->N; ++i) {
        const double w = o_exp(f->logw[i] - f->m);
        const double xv = f->x[i];
        const double hv = kind == 0 ? xv : kind == 1 ? xv * xv : kind == 2 ? o_exp(0.5 * xv) : 42.0;
        num += hv * w; den += w;
    }
    return num / den;
}

double orc_ref_run_series(int model, const double* theta, int N, const double* y, const double* z, int T,
                          uint32_t seed, int use_float, int fast_resampler, double* per_step) {
    return use_float ? ref_run<float>(model, theta, N, y, z, T, seed, fast_resampler, per_step)
                     : ref_run<double>(model, theta, N, y, z, T, seed, fast_resampler, per_step);
}

// ---- Liu-West ----
void* orc_lw_create(int N, uint64_t seed, uint32_t rep, const int* trans, const double* lo, const double* hi, double delta, int form, int rs) {
    LWFilter* f = new LWFilter(); f->init(N, seed, rep, trans, lo, hi, delta, form, rs); return f;
}
double orc_lw_expectation(void* h, int id) { return ((LWFilter*)h)->expectation(id); }
void orc_lw_destroy(void* h) { delete (LWFilter*)h; }
double orc_lw_step(void* h, double y, double z) { return ((LWFilter*)h)->step(y, z); }
double orc_lw_loglik(void* h) { return ((LWFilter*)h)->loglik; }
void orc_lw_param_means(void* h, double* out) { ((LWFilter*)h)->param_means(out); }
// state after the last step: x, theta (transformed, [4][N]), second-stage log-weights, k indices, ancestors, theta-bar, L
void orc_lw_state(void* h, double* x, double* theta, double* logw, uint32_t* kidx, uint32_t* anc, double* thetabar, double* L) {
    LWFilter* f = (LWFilter*)h;
    if (x) std::memcpy(x, f->x.data(), sizeof(double) * f->N);
    if (theta) for (int d = 0; d < DP; ++d) std::memcpy(theta + (size_t)d * f->N, f->th[d].data(), sizeof(double) * f->N);
    if (logw) std::memcpy(logw, f->cdfB.logw.data(), sizeof(double) * f->N);
    if (kidx) std::memcpy(kidx, f->kidx.data(), sizeof(uint32_t) * f->N);
    if (anc) std::memcpy(anc, f->anc.data(), sizeof(uint32_t) * f->N);
    if (thetabar) std::memcpy(thetabar, f->thetabar, sizeof(double) * DP);
    if (L) std::memcpy(L, f->L, sizeof(double) * DP * DP);
}
double orc_lw_ref_run(int N, const int* trans, const double* lo, const double* hi, double delta, const double* y, const double* z,
                      int T, uint32_t seed, double* per_step, double* means_out, int form, int rs) {
    return lw_ref_run(N, trans, lo, hi, delta, y, z, T, seed, per_step, means_out, form, rs);
}

// replicate aggregation, thread_pool.h:263-268
double orc_log_mean_exp(const double* v, int n) {
    double mx = *std::max_element(v, v + n), s = 0;
    for (int i = 0; i < n; ++i) s += std::exp(v[i] - mx);
    return mx + std::log(s) - std::log((double)n);
}

// param::transform inverse transforms + log-Jacobians, parameters.h:317-457
// kind: 0 null, 1 twice_fisher, 2 logit, 3 log  (enum order of parameters.h:27)
double orc_inv_transform(int kind, double tp) {
    switch (kind) {
        case 0: return tp;
        case 1: return (tp > 0) ? 2.0 / (1.0 + std::exp(-tp)) - 1.0 : 1.0 - 2.0 / (1.0 + std::exp(tp));
        case 2: return (tp > 0) ? 1.0 / (1.0 + std::exp(-tp)) : std::exp(tp) / (1.0 + std::exp(tp));
        default: return std::exp(tp);
    }
}
double orc_log_jacobian(int kind, double tp) {
    switch (kind) {
        case 0: return 0.0;
        case 1: return std::log(2.0) + tp - 2.0 * std::log(1.0 + std::exp(tp));
        case 2: return -tp - 2.0 * std::log(1.0 + std::exp(-tp));
        default: return tp;
    }
}

// exact log-likelihood of the linear-Gaussian model by the Kalman filter (independent anchor)
double orc_kalman_loglik(double phi, double sigma, double tau, const double* y, int T, double* per_step) {
    double mean = 0.0, var = sigma * sigma / (1.0 - phi * phi), ll = 0.0;
    for (int t = 0; t < T; ++t) {
        if (t > 0) { mean = phi * mean; var = phi * phi * var + sigma * sigma; }
        const double Sv = var + tau * tau;
        const double l = -0.5 * std::log(2.0 * M_PI * Sv) - 0.5 * (y[t] - mean) * (y[t] - mean) / Sv;
        if (per_step) per_step[t] = l;
        ll += l;
        const double K = var / Sv;
        mean = mean + K * (y[t] - mean);
        var = (1.0 - K) * var;
    }
    return ll;
}

}  // extern "C"
