// pf_kernels.h -- HIP kernels of the bootstrap-particle-filter step for gfx950 (wave64).
//
// One filter step = two kernels:
//   KA propagate_weight : [level-2 scan of tile sums] -> resampling targets -> two-level
//                         lower-bound search in the weight cdf -> gather ancestor state ->
//                         fSamp -> logGEv -> store x, logw, per-tile max.
//                         Replaces pf::BSFilter::filter's particle loop + the resampler's
//                         gather (call site example/estimate_univ_svol.h:124; in-tree twin
//                         include/ssme/liu_west_filter.h:1621-1640, :105-144).
//   KR normalize_scan   : global max from the per-tile maxima -> w = exp(logw - max) ->
//                         tile-local inclusive scan (cdf) + tile sums.
//                         Replaces the log-sum-exp passes (twin :1652-1659) and the weight
//                         normalisation of the resampler (:96-104).
// Particles are a structure of arrays in HBM: x[R][Npad], logw[R][Npad], cdf[R][Npad]
// (fp64), one row per filter/replicate; a tile is 2048 consecutive particles = 4 rows of
// 512 = 256 threads x 2 consecutive values, so every global access is one 16-byte
// double2 per lane, fully coalesced.
#pragma once
#include "ssme_math.h"

namespace ssme {

constexpr int kThreads = 256;
constexpr int kWave = 64;
constexpr int kRow = 512;
constexpr int kTile = 2048;
constexpr int kRowsPerTile = 4;
constexpr int kMaxTilesPerFilter = 2048;   // level-2 scan = up to 4 rows of 512 tile sums

enum { MODEL_SVOL = 0, MODEL_SVOL_LEVERAGE = 1, MODEL_LIN_GAUSS = 2 };
enum { RESAMP_MULTINOMIAL = 0, RESAMP_SYSTEMATIC = 1, RESAMP_STRATIFIED = 2, RESAMP_MULTINOMIAL_IID = 3 };

#define SSME_HALF_LOG_2PI 0.91893853320467274178

// Derived per-filter constants (host computes them with the same ssme_math functions).
struct ModelConst {
    double a0, a1, a2, a3, a4;
    int32_t bad;
    int32_t pad;
};

// Per-filter scalars living in device memory.
struct FilterScalars {
    double m;        // max log-weight of the last step
    double S;        // sum of exp(logw - m) of the last step (level-2 total)
    double prev;     // m_old + log S_old (log N after a resampling step)
    double loglik;   // running sum of log p(y_t | y_{1:t-1})
    double last_ll;  // last log conditional likelihood
    double pad[3];
};

struct StepArgs {
    // state
    const double* x_in;      // [R][Npad] particles of step t-1 (pre-resampling)
    double* x_out;           // [R][Npad]
    double* logw;            // [R][Npad]
    double* cdf;             // [R][Npad] tile-local inclusive sums of w
    uint32_t* anc;           // [R][Npad] or null
    double* tile_sum;        // [R][Bs]  A_b
    double* tile_esum;       // [R][Bs]  exponential-spacing tile sums (multinomial)
    double* tile_max;        // [R][Bs]
    FilterScalars* scal;     // [R]
    const ModelConst* mc;    // [R]
    const double* y;         // [T]
    const double* z;         // [T] or null
    double* per_step;        // [R][Tcap] or null
    int32_t N, Npad, B, Bs, nrows2, Bpow2;
    int32_t t, yi, Tcap;     // t: time index (RNG counter, schedule); yi: index into y/z
    int32_t model, resampler, resamp_sched;
    int32_t finalize_prev;   // KA: account log p(y_{t-1}|.) of the previous step
    uint32_t key0, key1, first_filter;
    double logN;
};

// ---------------------------------------------------------------------------------------
// Canonical row scan.  v[k][0..1] are this thread's two consecutive
// values of row k.  Afterwards value(k,c) = base[k] + (c ? s1[k] : s0[k]) is the inclusive
// sum and the exclusive sum is base[k] (+ s0[k] for c = 1); `total` is the sum of all rows.
// ---------------------------------------------------------------------------------------
template <int NR>
struct RowsScan {
    double base[NR], s0[NR], s1[NR], total;
};

template <int NR>
__device__ __forceinline__ void block_rows_scan(const double (&v)[NR][2], RowsScan<NR>& out, double* lds_w /* NR*4 */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double exc[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        out.s0[k] = v[k][0];
        out.s1[k] = out.s0[k] + v[k][1];
        double inc = out.s1[k];
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const double up = __shfl_up(inc, d, kWave);
            if (lane >= d) inc = inc + up;
        }
        const double e = __shfl_up(inc, 1, kWave);
        exc[k] = lane ? e : 0.0;
        if (lane == kWave - 1) lds_w[k * 4 + wave] = inc;
    }
    __syncthreads();
    double Q = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const double W0 = lds_w[k * 4 + 0], W1 = lds_w[k * 4 + 1], W2 = lds_w[k * 4 + 2], W3 = lds_w[k * 4 + 3];
        const double O0 = 0.0;
        const double O1 = O0 + W0;
        const double O2 = O1 + W1;
        const double O3 = O2 + W2;
        const double rowtot = O3 + W3;
        const double Ow = wave == 0 ? O0 : wave == 1 ? O1 : wave == 2 ? O2 : O3;
        out.base[k] = Q + (Ow + exc[k]);
        if (k + 1 < NR) Q = Q + rowtot; else out.total = Q + rowtot;
    }
    __syncthreads();   // lds_w may be reused by the caller
}

// Runtime row count (level-2 scans: 1..4 rows of tile sums).
struct Rows2 { double base[4], s0[4], s1[4], total; };

__device__ __forceinline__ void block_rows_scan_rt(const double (&v)[4][2], int nrows, Rows2& out, double* lds_w) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double exc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nrows) {
            out.s0[k] = v[k][0];
            out.s1[k] = out.s0[k] + v[k][1];
            double inc = out.s1[k];
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const double up = __shfl_up(inc, d, kWave);
                if (lane >= d) inc = inc + up;
            }
            const double e = __shfl_up(inc, 1, kWave);
            exc[k] = lane ? e : 0.0;
            if (lane == kWave - 1) lds_w[k * 4 + wave] = inc;
        }
    }
    __syncthreads();
    double Q = 0.0;
    out.total = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nrows) {
            const double W0 = lds_w[k * 4 + 0], W1 = lds_w[k * 4 + 1], W2 = lds_w[k * 4 + 2], W3 = lds_w[k * 4 + 3];
            const double O0 = 0.0;
            const double O1 = O0 + W0;
            const double O2 = O1 + W1;
            const double O3 = O2 + W2;
            const double rowtot = O3 + W3;
            const double Ow = wave == 0 ? O0 : wave == 1 ? O1 : wave == 2 ? O2 : O3;
            out.base[k] = Q + (Ow + exc[k]);
            if (k + 1 < nrows) Q = Q + rowtot; else out.total = Q + rowtot;
        }
    }
    __syncthreads();
}

// NaN-ignoring max fold ("if (v > m) m = v"), exact and order independent.
__device__ __forceinline__ double maxf(double m, double v) { return (v > m) ? v : m; }

__device__ __forceinline__ double block_max(double m, double* lds4) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = maxf(m, __shfl_xor(m, d, kWave));
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    double r = maxf(maxf(lds4[0], lds4[1]), maxf(lds4[2], lds4[3]));
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------
// Model callbacks, compiled in (the reference's virtual fSamp/q1Samp/logGEv).
// ---------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ double model_prop(const ModelConst& c, double x, double zn, double zcov) {
    if (MODEL == MODEL_SVOL_LEVERAGE) {   // test/test_pswarm.cpp:90-97
        const double e = dexp(-0.5 * x);
        const double mean = (c.a1 + c.a0 * (x - c.a1)) + (c.a4 * zcov) * e;
        return mean + zn * c.a3;
    }
    return c.a0 * x + zn * c.a1;          // univ_svol_bootstrap_filter.h:74-79
}

template <int MODEL>
__device__ __forceinline__ double model_logg(const ModelConst& c, double y, double x) {
    if (MODEL == MODEL_LIN_GAUSS) {
        const double d = (y - x) * c.a4;
        const double v = (-c.a3 - SSME_HALF_LOG_2PI) - 0.5 * (d * d);
        return c.bad ? -dinf() : v;
    }
    // univ_svol_bootstrap_filter.h:83-86 / test_pswarm.cpp:101-108 in kernel form
    const double logb = (MODEL == MODEL_SVOL) ? c.a3 : 0.0;
    const double ib2 = (MODEL == MODEL_SVOL) ? c.a4 : 1.0;
    const double hl = logb + 0.5 * x;
    const double e = dexp(-x);
    const double q = (y * y) * ib2;
    double v = (-hl - SSME_HALF_LOG_2PI) - 0.5 * (q * e);
    if (hl < -745.1332191019412) v = -dinf();
    if (MODEL == MODEL_SVOL && c.bad) v = -dinf();
    return v;
}

// two standard normals for the pair `pair` at time t (Box-Muller)
__device__ __forceinline__ void normal_pair(uint32_t pair, uint32_t t, uint32_t rep, uint32_t k0, uint32_t k1,
                                            double* z0, double* z1) {
    const u32x4 o = philox4x32_10(pair, t, rep, STREAM_PROP, k0, k1);
    const double u1 = u01_oc(o.v0, o.v1), u2 = u01_co(o.v2, o.v3);
    const double rad = dsqrt(-2.0 * dlog(u1));
    double sn, cs;
    dsincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

// fixed-probe lower bound over n = 2^k values; returns [0, n-1]
template <class F>
__device__ __forceinline__ int lower_bound_pow2(int n, double target, F get) {
    int pos = 0;
    for (int step = n >> 1; step >= 1; step >>= 1)
        if (get(pos + step - 1) < target) pos += step;
    return pos;
}

// ---------------------------------------------------------------------------------------
// KA: propagate + weight (with fused resampling search/gather of the previous step)
// grid = (B tiles, R filters), block = 256
// ---------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(kThreads) void ka_propagate_weight(const StepArgs a) {
    __shared__ double lds_w[16];
    __shared__ double lds_bc[4];
    __shared__ double lds_P[kMaxTilesPerFilter];
    __shared__ double lds_T[kMaxTilesPerFilter];

    const int tid = threadIdx.x;
    const int b = blockIdx.x, r = blockIdx.y;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const size_t rowoff = (size_t)r * a.Npad;
    const ModelConst mc = a.mc[r];
    const double y = a.y[a.yi];
    const double zcov = a.z ? a.z[a.yi] : 0.0;
    const bool resampled = (a.t > 0) && (a.t % a.resamp_sched == 0);

    // --- level-2 scan of the previous step's tile sums: prefixes P_b, ends T_b, total S ---
    double S = 0.0;
    if (a.t > 0 && (resampled || (b == 0 && a.finalize_prev))) {
        double v[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = k * kRow + 2 * tid;
            if (k < a.nrows2) {
                const double2 t2 = *reinterpret_cast<const double2*>(a.tile_sum + (size_t)r * a.Bs + j);
                v[k][0] = t2.x; v[k][1] = t2.y;
            } else { v[k][0] = 0.0; v[k][1] = 0.0; }
        }
        Rows2 l2;
        block_rows_scan_rt(v, a.nrows2, l2, lds_w);
        S = l2.total;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < a.nrows2) {
                const int j = k * kRow + 2 * tid;
                const double p0 = l2.base[k], p1 = l2.base[k] + l2.s0[k];
                lds_P[j] = p0; lds_P[j + 1] = p1;
                lds_T[j] = (j < a.B) ? p0 + v[k][0] : dinf();
                lds_T[j + 1] = (j + 1 < a.B) ? p1 + v[k][1] : dinf();
            }
        }
        // pad the level-1 search table to a power of two
        for (int j = a.nrows2 * kRow + tid; j < a.Bpow2; j += kThreads) lds_T[j] = dinf();
        __syncthreads();
        if (b == 0 && tid == 0 && a.finalize_prev) {
            FilterScalars* sc = a.scal + r;
            const double lse = sc->m + dlog(S);
            const double ll = lse - sc->prev;
            sc->S = S;
            sc->last_ll = ll;
            sc->loglik = sc->loglik + ll;
            sc->prev = resampled ? a.logN : lse;
            if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
        }
    }

    // --- standard normals for my 8 particles ---
    double zn[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t pair = (uint32_t)(b * (kTile / 2) + k * (kRow / 2) + tid);
        normal_pair(pair, (uint32_t)a.t, rep, a.key0, a.key1, &zn[k][0], &zn[k][1]);
    }

    double xin[4][2], lw_old[4][2];
    if (a.t == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xin[k][0] = 0.0; xin[k][1] = 0.0; lw_old[k][0] = 0.0; lw_old[k][1] = 0.0; }
    } else if (!resampled) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = rowoff + (size_t)b * kTile + k * kRow + 2 * tid;
            const double2 xv = *reinterpret_cast<const double2*>(a.x_in + idx);
            const double2 lv = *reinterpret_cast<const double2*>(a.logw + idx);
            xin[k][0] = xv.x; xin[k][1] = xv.y; lw_old[k][0] = lv.x; lw_old[k][1] = lv.y;
        }
    } else {
        // --- resampling targets against the cdf of step t-1 ---
        double tau[4][2];
        if (a.resampler == RESAMP_MULTINOMIAL) {
            // exponential spacings (liu_west_filter.h:105-139): U_(i) = sum_{j<=i} E_j / G
            double E[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i0 = b * kTile + k * kRow + 2 * tid;
                const u32x4 o = philox4x32_10((uint32_t)(i0 >> 1), (uint32_t)a.t, rep, STREAM_RESAMP, a.key0, a.key1);
                E[k][0] = (i0 < a.N) ? -dlog(u01_oc(o.v0, o.v1)) : 0.0;
                E[k][1] = (i0 + 1 < a.N) ? -dlog(u01_oc(o.v2, o.v3)) : 0.0;
            }
            RowsScan<4> es;
            block_rows_scan<4>(E, es, lds_w);
            // level-2 over the exponential tile sums written by KR(t-1)
            double v[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = k * kRow + 2 * tid;
                if (k < a.nrows2) {
                    const double2 t2 = *reinterpret_cast<const double2*>(a.tile_esum + (size_t)r * a.Bs + j);
                    v[k][0] = t2.x; v[k][1] = t2.y;
                } else { v[k][0] = 0.0; v[k][1] = 0.0; }
            }
            Rows2 l2e;
            block_rows_scan_rt(v, a.nrows2, l2e, lds_w);
            // broadcast my tile's exclusive prefix PE_b
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < a.nrows2) {
                    const int j = k * kRow + 2 * tid;
                    if (j == b) lds_bc[0] = l2e.base[k];
                    if (j + 1 == b) lds_bc[0] = l2e.base[k] + l2e.s0[k];
                }
            }
            __syncthreads();
            const double PEb = lds_bc[0];
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, a.key0, a.key1);
            const double G = l2e.total + (-dlog(u01_oc(ox.v0, ox.v1)));
            const double scale = S / G;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tau[k][0] = (PEb + (es.base[k] + es.s0[k])) * scale;
                tau[k][1] = (PEb + (es.base[k] + es.s1[k])) * scale;
            }
            __syncthreads();
        } else if (a.resampler == RESAMP_SYSTEMATIC) {
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, a.key0, a.key1);
            const double u0 = u01_co(ox.v0, ox.v1);
            const double scale = S / (double)a.N;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i0 = b * kTile + k * kRow + 2 * tid;
                tau[k][0] = ((double)i0 + u0) * scale;
                tau[k][1] = ((double)(i0 + 1) + u0) * scale;
            }
        } else {
            const double scale = S / (double)a.N;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i0 = b * kTile + k * kRow + 2 * tid;
                const u32x4 o = philox4x32_10((uint32_t)(i0 >> 1), (uint32_t)a.t, rep, STREAM_RESAMP, a.key0, a.key1);
                const double v0 = u01_co(o.v0, o.v1), v1 = u01_co(o.v2, o.v3);
                if (a.resampler == RESAMP_STRATIFIED) {
                    tau[k][0] = ((double)i0 + v0) * scale;
                    tau[k][1] = ((double)(i0 + 1) + v1) * scale;
                } else {
                    tau[k][0] = v0 * S;
                    tau[k][1] = v1 * S;
                }
            }
        }
        // --- two-level fixed-probe lower bound, then gather ---
        const double* cdf_r = a.cdf + rowoff;
        const double* xin_r = a.x_in + rowoff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double target = tau[k][c];
                int bb = lower_bound_pow2(a.Bpow2, target, [&](int j) { return lds_T[j]; });
                bb = bb < a.B - 1 ? bb : a.B - 1;
                const double Pb = lds_P[bb];
                const double* tile = cdf_r + (size_t)bb * kTile;
                const int j = lower_bound_pow2(kTile, target, [&](int q) { return Pb + tile[q]; });
                int anc = bb * kTile + j;
                anc = anc < a.N - 1 ? anc : a.N - 1;
                const int i = b * kTile + k * kRow + 2 * tid + c;
                if (a.anc && i < a.N) a.anc[rowoff + i] = (uint32_t)anc;
                xin[k][c] = xin_r[anc];
                lw_old[k][c] = 0.0;
            }
        }
    }

    // --- fSamp / q1Samp, logGEv, store, tile max ---
    double mx = -dinf();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = b * kTile + k * kRow + 2 * tid;
        double xo[2], lo[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double xn = (a.t == 0) ? zn[k][c] * mc.a2 : model_prop<MODEL>(mc, xin[k][c], zn[k][c], zcov);
            const double lg = lw_old[k][c] + model_logg<MODEL>(mc, y, xn);
            const bool valid = (i0 + c) < a.N;
            xo[c] = valid ? xn : 0.0;
            lo[c] = valid ? lg : -dinf();
            if (valid) mx = maxf(mx, lg);
        }
        const size_t idx = rowoff + (size_t)i0;
        *reinterpret_cast<double2*>(a.x_out + idx) = make_double2(xo[0], xo[1]);
        *reinterpret_cast<double2*>(a.logw + idx) = make_double2(lo[0], lo[1]);
    }
    mx = block_max(mx, lds_bc);
    if (tid == 0) a.tile_max[(size_t)r * a.Bs + b] = mx;
}

// ---------------------------------------------------------------------------------------
// KR: normalise + tile scan.  grid = (B, R), block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void kr_normalize_scan(const StepArgs a) {
    __shared__ double lds_w[16];
    __shared__ double lds_bc[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, r = blockIdx.y;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const size_t rowoff = (size_t)r * a.Npad;

    // global max over the per-tile maxima
    double m = -dinf();
    for (int j = tid; j < a.B; j += kThreads) m = maxf(m, a.tile_max[(size_t)r * a.Bs + j]);
    m = block_max(m, lds_bc);

    double w[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = b * kTile + k * kRow + 2 * tid;
        const double2 lv = *reinterpret_cast<const double2*>(a.logw + rowoff + i0);
        w[k][0] = (i0 < a.N) ? dexp(lv.x - m) : 0.0;
        w[k][1] = (i0 + 1 < a.N) ? dexp(lv.y - m) : 0.0;
    }
    RowsScan<4> sc;
    block_rows_scan<4>(w, sc, lds_w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i0 = b * kTile + k * kRow + 2 * tid;
        *reinterpret_cast<double2*>(a.cdf + rowoff + i0) = make_double2(sc.base[k] + sc.s0[k], sc.base[k] + sc.s1[k]);
    }
    if (tid == 0) {
        a.tile_sum[(size_t)r * a.Bs + b] = sc.total;
        if (b == 0) a.scal[r].m = m;
    }

    // exponential-spacing tile sums for the resampling consumed by step t+1
    const bool resample_now = ((a.t + 1) % a.resamp_sched == 0);
    if (a.resampler == RESAMP_MULTINOMIAL && resample_now) {
        double E[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = b * kTile + k * kRow + 2 * tid;
            const u32x4 o = philox4x32_10((uint32_t)(i0 >> 1), (uint32_t)(a.t + 1), rep, STREAM_RESAMP, a.key0, a.key1);
            E[k][0] = (i0 < a.N) ? -dlog(u01_oc(o.v0, o.v1)) : 0.0;
            E[k][1] = (i0 + 1 < a.N) ? -dlog(u01_oc(o.v2, o.v3)) : 0.0;
        }
        RowsScan<4> es;
        block_rows_scan<4>(E, es, lds_w);
        if (tid == 0) a.tile_esum[(size_t)r * a.Bs + b] = es.total;
    }
}

// ---------------------------------------------------------------------------------------
// KF: account the last step's log conditional likelihood.  grid = (R), block = 256
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void kf_finalize(const StepArgs a) {
    __shared__ double lds_w[16];
    const int tid = threadIdx.x;
    const int r = blockIdx.x;
    double v[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = k * kRow + 2 * tid;
        if (k < a.nrows2) {
            const double2 t2 = *reinterpret_cast<const double2*>(a.tile_sum + (size_t)r * a.Bs + j);
            v[k][0] = t2.x; v[k][1] = t2.y;
        } else { v[k][0] = 0.0; v[k][1] = 0.0; }
    }
    Rows2 l2;
    block_rows_scan_rt(v, a.nrows2, l2, lds_w);
    if (tid == 0) {
        FilterScalars* sc = a.scal + r;
        const bool resample_now = ((a.t + 1) % a.resamp_sched == 0);
        const double lse = sc->m + dlog(l2.total);
        const double ll = lse - sc->prev;
        sc->S = l2.total;
        sc->last_ll = ll;
        sc->loglik = sc->loglik + ll;
        sc->prev = resample_now ? a.logN : lse;
        if (a.per_step) a.per_step[(size_t)r * a.Tcap + a.t] = ll;
    }
}

// ---------------------------------------------------------------------------------------
// Weighted expectation of a built-in functional with the last step's weights
// (getExpectations(); twin liu_west_filter.h:1662-1683).  grid = (R), block = 256.
// Partial sums: per-thread strided, wave shuffle tree, 4 waves in order (deterministic).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_expectation(const double* x, const double* logw, const FilterScalars* scal,
                                                          int N, int Npad, int functional, double* out) {
    __shared__ double lds_n[4], lds_d[4];
    const int tid = threadIdx.x, r = blockIdx.x;
    const double m = scal[r].m;
    double num = 0.0, den = 0.0;
    for (int i = tid; i < N; i += kThreads) {
        const double w = dexp(logw[(size_t)r * Npad + i] - m);
        const double xv = x[(size_t)r * Npad + i];
        const double hv = functional == 0 ? xv : functional == 1 ? xv * xv : functional == 2 ? dexp(0.5 * xv) : 42.0;
        num = num + hv * w;
        den = den + w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { num = num + __shfl_xor(num, d, kWave); den = den + __shfl_xor(den, d, kWave); }
    if ((tid & 63) == 0) { lds_n[tid >> 6] = num; lds_d[tid >> 6] = den; }
    __syncthreads();
    if (tid == 0) {
        const double n4 = ((lds_n[0] + lds_n[1]) + lds_n[2]) + lds_n[3];
        const double d4 = ((lds_d[0] + lds_d[1]) + lds_d[2]) + lds_d[3];
        out[r] = n4 / d4;
    }
}

}  // namespace ssme
