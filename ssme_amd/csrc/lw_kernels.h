// lw_kernels.h -- Liu-West filter (auxiliary-particle form with covariates) for gfx950.
//
// Replaces LWFilterWithCovs::filter (include/ssme/liu_west_filter.h:971-1159) with the model callbacks of
// svol_lw_1_par (test/test_liu_west.cpp:82-157), update_parameter_proposal_components (:1184-1198), the
// multinomial resampler of states and parameters (:91-145) and the inverse parameter transforms
// (include/ssme/parameters.h:317-457, evaluated per dimension by enum instead of polymorphic objects).
// Both forms of the reference's filter family with covariates:
//   form 0  auxiliary-particle form   LWFilterWithCovs::filter    liu_west_filter.h:971-1159   (model svol_lw_1_par)
//   form 1  plain SISR form           LWFilter2WithCovs::filter   liu_west_filter.h:2191-2343  (model svol_lw_2_par,
//           test/test_liu_west.cpp:214-358: qSamp = the transition, so logFEv - logQEv cancels; no first-stage weights,
//           no k draw: stage 1 only resamples and takes the parameter moments, stage 2 adds logG to the carried weight)
// and the resampling schedule m_rs (:1139-1140, :2317-2318): resampling is lazy (the draw "of step t-1" runs at the
// start of step t); on steps without it the populations stay in place and the second-stage log-weights are carried.
// One time step t >= 1 is three launches:
//   k_lw_stage1 : resample (x, theta) by the previous second-stage weights (exact integer cdf B, search in LDS,
//                 gather) -> first-stage weight logG(y, propMu(x, z, theta)) -> cdf A; tile partial sums of the
//                 14 moments of the transformed parameters
//   k_lw_mid    : one block per filter: theta-bar, V, Cholesky factor of (1 - a^2) V; log-sum-exp of stage 1
//   k_lw_stage2 : k ~ Categorical(first-stage weights) (cdf A) -> shrink + jitter theta -> fSamp ->
//                 second-stage weight -> cdf B
// Layout: x[R][Npad] fp64; theta[R][Npad][4] (transformed space): ONE 32-byte record per particle, so that the resampling
// gather of a particle's parameters is one 32-byte access instead of four scattered 8-byte ones.
#pragma once
#include "pf_kernels.h"

namespace ssme {

constexpr int kDP = 4;             // phi, mu, sigma, rho
constexpr int kLwNT = 512;         // threads per 2048-particle tile (2 pairs per thread)
constexpr int kNMom = 14;          // 4 means + 10 second moments
enum { TR_NULL = 0, TR_TWICE_FISHER = 1, TR_LOGIT = 2, TR_LOG = 3 };      // enum order of parameters.h:27
// The transform set of the reference's Liu-West test models (test_liu_west.cpp:70: logit, null, log, twice_fisher) also exists as a
// compile-time constant (template parameter FT of the stage kernels): without the four scalar branches per transform the
// compiler interleaves the four exp chains of a particle -- N = 2^20: 67.9 -> 63.1 us per step.  Any other set: FT = false.
template <bool FT>
__device__ __forceinline__ int lw_trans_kind(const int32_t* trans, int d) {
    if (FT) return d == 0 ? TR_LOGIT : (d == 1 ? TR_NULL : (d == 2 ? TR_LOG : TR_TWICE_FISHER));
    return trans[d];
}
enum { STREAM_LW_PRIOR = 3 /* and 4 */, STREAM_LW_JIT = 5 /* and 6 */, STREAM_LW_K = 7, STREAM_LW_K_EXTRA = 8,
       STREAM_GAMMA_K = 80 };

struct LwScalars {
    double mB, SB;       // level-2 of the second-stage weights of the last step
    double lse1;         // log-sum-exp of the first-stage weights of the last step
    double loglik, last_ll;
    double prev;         // log-sum-exp of the log-weights the step started from: log N after a resampling, else the last lse
    double pad[2];
};

struct LwArgs {
    double* xB; double* thB;                    // [R][Npad], [R][4][Npad]: population after stage 2 (pre-resampling)
    double* xr; double* thr; double* lw1;       // resampled population and (form 0) logG(y | propMu) of it, (form 1) its carried log-weights
    double* lwB;                                // second-stage log-weights of the last step, kept only when resamp_sched > 1 (else null)
    double *cdfA, *tsumA, *tmaxA;               // exact integer cdf of the first-stage weights
    double *cdfB, *tsumB, *tmaxB;               // ... of the second-stage weights
    double* mom;                                // [R][B][16] tile partial sums of the 14 moments
    double* prop;                               // [R][16]: theta-bar[4], L (lower triangle, row-major)[10]
    uint32_t *anc, *kidx;                       // debug: resampling ancestors / k indices, or null
    uint32_t* ancbuf;                           // compose = 1: [R][Npad] resampling ancestors of this step (stage 1 -> stage 2)
    int32_t compose;                            // 1 (unsharded handles): stage 1 does not materialise the resampled population; stage 2
                                                // gathers (x, theta) of particle ancbuf[k_i] from the population the step started from
                                                // (xB / thB) and writes the new one into xr / thr -- the host then swaps the two pairs.
                                                // 40 of 208 bytes per particle-step less (the resampled (x, theta) written and re-read)
    LwScalars* scal;
    const double *y, *z;
    double y_now, z_now;                        // step API: this call's observation / covariate in the kernel arguments (by_value = 1)
    int32_t by_value;
    double* ll_host;                            // step API: host-mapped buffer for the R log conditional likelihoods, or null
    double* per_step;
    const double *gamB, *pgamB, *gtotB;         // Gamma tables of the resampling draw   (stream base 16)
    const double *gamA, *pgamA, *gtotA;         // Gamma tables of the k draw            (stream base 80)
    int32_t N, Npad, B, Bs, Bpow2, rshift, R;
    int32_t t, yi, gi, Tcap;
    int32_t finalize_prev;                      // stage 1 accounts log p(y_{t-1} | .) (series mode); 0 in step mode
    int32_t form;                               // 0: auxiliary-particle form (LWFilterWithCovs), 1: SISR form (LWFilter2WithCovs)
    int32_t resamp_sched;                       // m_rs: resample when (t + 1) % m_rs == 0
    // particle-sharded filter (all zero / Npad otherwise): this launch computes tiles tile0 .. tile0 + gridDim.x - 1 and
    // stores them at local offsets; the arrays a stage READS (stage 1: xB, thB, cdfB; stage 2: xr, thr, lw1, cdfA) are
    // windows starting at tile win_tile0 (the parameter records travel with their particles: [tiles][2048][4])
    int32_t tile0, win_tile0;
    int32_t win_tiles;                          // C++ shard driver, fixed-halo path: tiles held in the source windows (0: unchecked)
    int32_t* win_flag;                          // ... and where to record that an output tile's sources left them, or null
    int32_t fuse_mid;                           // 1: k_lw_stage2 takes theta-bar and the Cholesky factor itself (no k_lw_mid launch)
    double* momtot;                             // [R][16] moment totals from k_lw_mom_totals (filters of many tiles), or null: k_lw_mid adds them itself
    // split level-2 (k_level2_plan; filters of more than 2048 tiles or by policy): per draw T', A/A', source ranges, (m, S)
    const double *l2B_T, *l2B_R, *l2A_T, *l2A_R;      // [R][Bs]
    const int32_t *l2B_lo, *l2B_hi, *l2A_lo, *l2A_hi;
    const FilterScalars *l2B_s, *l2A_s;              // [R]: m and S of the draw's weights
    uint32_t key0, key1, first_filter;
    double logN, a_shrink;
    int32_t trans[kDP];
    double lo[kDP], hi[kDP];
};

// ---- parameter transforms (parameters.h inv_trans / trans), libm-free ------------------------------------
// `kind` is uniform (a kernel argument), so the switch is a scalar branch; exp(-|tp|) serves both signs.
__device__ __forceinline__ double tr_inv(int kind, double tp, const ExpTabEntry* etab) {
    if (kind == TR_NULL) return tp;
    if (kind == TR_LOG) return dexp_scaled_t(tp, 0, etab);
    const double t = dexp_scaled_t((tp >= 0.0) ? -tp : tp, 0, etab);
    const double den = 1.0 + t;
    if (kind == TR_LOGIT) return ((tp >= 0.0) ? 1.0 : t) / den;       // one division: the numerator is selected first
    const double q = 2.0 / den;
    return (tp >= 0.0) ? q - 1.0 : 1.0 - q;
}
__device__ __forceinline__ double tr_fwd(int kind, double p) {
    switch (kind) {
        case TR_NULL: return p;
        case TR_TWICE_FISHER: return dlog(1.0 + p) - dlog(1.0 - p);
        case TR_LOGIT: return dlog(p) - dlog(1.0 - p);
        default: return dlog(p);
    }
}
// theta records: [Npad][4] doubles per filter; a pair of particles is 64 contiguous bytes
__device__ __forceinline__ void th_load(const double* th, size_t idx, double (&t)[kDP]) {
    const double2* p = reinterpret_cast<const double2*>(th + idx * kDP);
    const double2 a = p[0], b = p[1];
    t[0] = a.x; t[1] = a.y; t[2] = b.x; t[3] = b.y;
}
__device__ __forceinline__ void th_store_pair(double* th, size_t idx0, const double (&t)[kDP][2]) {
    double2* p = reinterpret_cast<double2*>(th + idx0 * kDP);
    p[0] = make_double2(t[0][0], t[1][0]); p[1] = make_double2(t[2][0], t[3][0]);
    p[2] = make_double2(t[0][1], t[1][1]); p[3] = make_double2(t[2][1], t[3][1]);
}
// model callbacks of svol_lw_1_par
__device__ __forceinline__ double lw_logg(double y, double x, const ExpTabEntry* etab) {      // test_liu_west.cpp:132-136, kernel form
    const double hl = 0.5 * x;
    double v = (-hl - SSME_HALF_LOG_2PI) - 0.5 * ((y * y) * dexp_scaled_t(-x, 0, etab));
    if (hl < -745.1332191019412) v = -dinf();
    return v;
}
__device__ __forceinline__ double lw_propmu(double x, double z, const double (&tu)[kDP], const ExpTabEntry* etab) {    // :93-101
    double xt = tu[1] + tu[0] * (x - tu[1]);
    xt = xt + ((z * tu[3]) * tu[2]) * dexp_scaled_t(-0.5 * x, 0, etab);
    return xt;
}

// ---------------------------------------------------------------------------------------
// Shared selection: indices idx[k][c] = #{j : C_j < tau} for this tile's sorted-uniform targets against the exact
// integer cdf (tile sums tsum / maxima tmax / tile-local cdf).  Same arithmetic as k_filter_step.
// Returns the level-2 results m (global max log-weight) and S (integer weight sum).  Block = 512 threads.
// ---------------------------------------------------------------------------------------
struct LwLds {
    double* lds_T; double* lds_R; double* lds_stage;     // dynamic LDS
    double* seg_a; double* seg_l2; double* d1; int* cnt;
    const LogTabEntry* ltab; const ExpTabEntry* etab;    // dlog_u / dexp_scaled_t tables in LDS
};

// level-2 results of one draw as k_level2_plan left them (split level-2)
struct L2View { const double* T; const double* R; const int32_t* lo; const int32_t* hi; double m, S; };

template <bool BIG>
__device__ __forceinline__ void lw_select(const double* tsum, const double* tmax, const double* cdf, int B, int Bpow2, int rshift,
                                          int N, int b, double gam, double pgam, double pgam_next, double G,
                                          const u32x4 (&draw)[2], const LwLds& L, int (&idx)[2][2],
                                          double& m_out, double& S_out, int win_tile0, const L2View& v, int win_tiles = 0,
                                          int32_t* win_flag = nullptr) {
    constexpr int NT = kLwNT, NK = 2, NE = 4;
    const int tid = threadIdx.x;
    const int i_first = b * kTile;
    __builtin_amdgcn_s_setprio(3);                 // wave priority falls as the workgroup advances (see prio_at)
    double t_scale;
    int lo, hi;
    if constexpr (BIG) {
        m_out = v.m; S_out = v.S;
        t_scale = v.S / G;
        lo = v.lo[b]; hi = v.hi[b];
    } else {
        double A2[NE], M2[NE];
        level2_load<NT>(tsum, tmax, B, A2, M2);
        if (tid == 0) { L.cnt[0] = 0; L.cnt[1] = 0; }
        double Ap[NE], Tinc[NE], m, S;
        level2_scan<NT>(A2, M2, B, rshift, m, Ap, Tinc, S, L.d1, L.seg_l2, L.etab);
        m_out = m; S_out = S;
        t_scale = S / G;
        const double t_lo = __builtin_ceil(pgam * t_scale);
        const double t_hi = __builtin_ceil(pgam_next * t_scale) + (S * 0x1.0p-40 + 2.0);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e * NT < Bpow2) {
                const int j = e * NT + tid;
                if (j < Bpow2) {
                    L.lds_T[j] = (j < B) ? Tinc[e] : dinf();
                    L.lds_R[j] = (j < B) ? A2[e] / Ap[e] : 0.0;
                }
                const int w_lo = __popcll(__ballot(j < B && Tinc[e] < t_lo));
                const int w_hi = __popcll(__ballot(j < B && Tinc[e] < t_hi));
                if ((tid & 63) == 0) { if (w_lo) atomicAdd(&L.cnt[0], w_lo); if (w_hi) atomicAdd(&L.cnt[1], w_hi); }
            }
        }
        __syncthreads();
        lo = L.cnt[0]; hi = L.cnt[1];
    }
    lo = lo < B - 1 ? lo : B - 1;
    hi = hi < B - 1 ? hi : B - 1;
    int bb_min = __builtin_amdgcn_readfirstlane(lo);
    int span = __builtin_amdgcn_readfirstlane(hi) - bb_min + 1;
    if (win_flag) {
        // fixed-halo sharding (see k_filter_step): sources outside the exchanged window are reported; this launch stays in bounds
        const int first = win_tile0 > 0 ? win_tile0 : 0, last = win_tile0 + win_tiles - 1 < B - 1 ? win_tile0 + win_tiles - 1 : B - 1;
        if (bb_min < first || bb_min + span - 1 > last) {
            if (tid == 0) atomicOr(win_flag, 1);
            bb_min = bb_min < first ? first : (bb_min > last ? last : bb_min);
            span = 1;
        }
    }
    double2 stg0[NK], stg1[NK], stg2[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { stg0[k] = make_double2(0.0, 0.0); stg1[k] = stg0[k]; stg2[k] = stg0[k]; }
    if (span <= kStageTiles) {
        const double* src = cdf + (size_t)(bb_min - win_tile0) * kTile + tid * 2;
#pragma unroll
        for (int k = 0; k < NK; ++k) stg0[k] = *reinterpret_cast<const double2*>(src + k * NT * 2);
        if (span >= 2) {
#pragma unroll
            for (int k = 0; k < NK; ++k) stg1[k] = *reinterpret_cast<const double2*>(src + kTile + k * NT * 2);
        }
        if (span >= 3) {
#pragma unroll
            for (int k = 0; k < NK; ++k) stg2[k] = *reinterpret_cast<const double2*>(src + 2 * kTile + k * NT * 2);
        }
    }
    __builtin_amdgcn_s_setprio(2);
    // exponential spacings (liu_west_filter.h:105-139), exact tile scan; hides the tile loads
    double qe[NK][2], le[NK][2], se;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = i_first + (k * NT + tid) * 2;
        double e0, e1;
        pair_spacings(draw[k], L.ltab, &e0, &e1);          // words 2-3 of the pair's draw (32-bit uniforms, table log)
        qe[k][0] = __builtin_rint(e0 * 34359738368.0);
        qe[k][1] = __builtin_rint(e1 * 34359738368.0);
        if (i_first + kTile > N) {        // uniform: only the ragged last tile masks the particles beyond N
            asm volatile("");
            if (!(i0 < N)) qe[k][0] = 0.0;
            if (!(i0 + 1 < N)) qe[k][1] = 0.0;
        }
    }
    block_scan_f64<NT, NK>(qe, le, se, L.seg_a);
    const double ratio = gam / se;
    double tau[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double t1 = ratio * le[k][c];
            const double t2 = pgam + t1;
            tau[k][c] = __builtin_ceil(t2 * t_scale);
        }
    }
    if (span <= kStageTiles) {
        double* dst = L.lds_stage + tid * 2;
#pragma unroll
        for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + k * NT * 2) = stg0[k];
        if (span >= 2) {
#pragma unroll
            for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + kTile + k * NT * 2) = stg1[k];
        }
        if (span >= 3) {
#pragma unroll
            for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + 2 * kTile + k * NT * 2) = stg2[k];
        }
        const int b1 = bb_min + 1 < B ? bb_min + 1 : B - 1, b2 = bb_min + 2 < B ? bb_min + 2 : B - 1;
        const double T0 = BIG ? v.T[bb_min] : L.lds_T[bb_min];
        const double T1 = (bb_min + 1 < B) ? (BIG ? v.T[bb_min + 1] : L.lds_T[bb_min + 1]) : dinf();
        const double Pm = bb_min ? (BIG ? v.T[bb_min - 1] : L.lds_T[bb_min - 1]) : 0.0;
        const double R0 = BIG ? v.R[bb_min] : L.lds_R[bb_min], R1 = BIG ? v.R[b1] : L.lds_R[b1], R2 = BIG ? v.R[b2] : L.lds_R[b2];
        __syncthreads();
        // the four count-searches of a thread descend together (staged_search, pf_kernels.h)
        int soff[NK][2];
        staged_search<kTile, NK>(tau, span, Pm, T0, T1, R0, R1, R2, L.lds_stage, soff);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int a = bb_min * kTile + soff[k][c];
                idx[k][c] = a < N - 1 ? a : N - 1;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double target = tau[k][c];
                int bb = count_less_pow2(Bpow2, target, [&](int j) { return BIG ? (j < B ? v.T[j] : dinf()) : L.lds_T[j]; });
                bb = bb < B - 1 ? bb : B - 1;
                const double Pb = bb ? (BIG ? v.T[bb - 1] : L.lds_T[bb - 1]) : 0.0;
                const double tloc = __builtin_ceil((target - Pb) * (BIG ? v.R[bb] : L.lds_R[bb]));
                const double* tile = cdf + (size_t)(bb - win_tile0) * kTile;
                const int j = count_less_pow2(kTile, tloc, [&](int q) { return tile[q]; });
                int a = bb * kTile + j;
                idx[k][c] = a < N - 1 ? a : N - 1;
            }
        }
    }
    __builtin_amdgcn_s_setprio(1);
}

// Level-2 alone (steps without a resampling draw still account the log-sum-exp of the weights they start from)
template <bool BIG>
__device__ __forceinline__ void lw_level2_only(const double* tsum, const double* tmax, int B, int rshift, const LwLds& L,
                                               double& m_out, double& S_out, const L2View& v) {
    if constexpr (BIG) { m_out = v.m; S_out = v.S; }
    else {
        constexpr int NT = kLwNT, NE = 4;
        double A2[NE], M2[NE], Ap[NE], Tinc[NE];
        level2_load<NT>(tsum, tmax, B, A2, M2);
        level2_scan<NT>(A2, M2, B, rshift, m_out, Ap, Tinc, S_out, L.d1, L.seg_l2, L.etab);
    }
}

// log-weights lg[k][c] of this tile -> tile max, fixed-point weights, exact tile scan; stores cdf / tile sum / tile max
__device__ __forceinline__ void lw_store_cdf(const double (&lg)[2][2], int /*N*/, int i_first, double* cdf_row, double* tsum_row,
                                             double* tmax_row, int b, double* lds_d, double* lds_seg, const ExpTabEntry* etab, int tile0 = 0) {
    constexpr int NT = kLwNT, NK = 2;
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(0);
    double mx = -dinf();
    bool nan = false;
    // (callers give particles beyond N the log-weight -inf: no part in the maximum, and the clamped exp makes their q exactly 0)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) { const double l = lg[k][c]; nan = nan || (l != l); mx = (l > mx) ? l : mx; }
    }
    const double mb = block_max_nanprop<NT>(mx, nan, lds_d);
    double q[NK][2], inc[NK][2], total;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        q[k][0] = __builtin_rint(dexp_scaled_t(lg[k][0] - mb, kTileShift, etab));
        q[k][1] = __builtin_rint(dexp_scaled_t(lg[k][1] - mb, kTileShift, etab));
    }
    block_scan_f64<NT, NK>(q, inc, total, lds_seg);
#pragma unroll
    for (int k = 0; k < NK; ++k)
        *reinterpret_cast<double2*>(cdf_row + (i_first - tile0 * kTile) + (k * NT + tid) * 2) = make_double2(inc[k][0], inc[k][1]);
    if (tid == 0) { tsum_row[b - tile0] = total; tmax_row[b - tile0] = mb; }
}

#define LW_LDS_SETUP(a)                                                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                  \
    const int nT2 = (BIG || (a).Bpow2 < 2) ? 2 : (a).Bpow2;                                               \
    __shared__ double lds_seg_a[16];                                                                      \
    __shared__ double lds_seg_l2[64];                                                                     \
    __shared__ double lds_seg_c[16];                                                                      \
    __shared__ double lds_d1[16];                                                                         \
    __shared__ double lds_d2[16];                                                                         \
    __shared__ int lds_cnt[2];                                                                            \
    __shared__ __attribute__((aligned(16))) DrawTabs lds_dtab;                                                 \
    __shared__ ExpTabEntry lds_etab[SSME_EXP_TABLE_SIZE];                                                 \
    load_log_table<kLwNT>(&lds_dtab);                                                                      \
    load_exp_table<kLwNT>(lds_etab);                                                                      \
    LwLds L;                                                                                              \
    L.lds_stage = reinterpret_cast<double*>(smem); L.lds_T = L.lds_stage + kStageTiles * kTile; L.lds_R = L.lds_T + nT2;  \
    L.seg_a = lds_seg_a; L.seg_l2 = lds_seg_l2; L.d1 = lds_d1; L.cnt = lds_cnt;                           \
    L.ltab = lds_dtab.log; L.etab = lds_etab;                                                                 \
    __syncthreads();

// ---------------------------------------------------------------------------------------
// t = 0: prior draws, q1Samp, weights (liu_west_filter.h:1103-1122).  grid = (B, R), block = 512
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLwNT) void k_lw_init(const LwArgs a) {
    constexpr int NT = kLwNT, NK = 2;
    __shared__ double lds_seg_c[16];
    __shared__ double lds_d2[16];
    __shared__ ExpTabEntry lds_etab[SSME_EXP_TABLE_SIZE];
    load_exp_table<kLwNT>(lds_etab);
    __syncthreads();
    const int tid = threadIdx.x;
    const int gtile = xcd_tile_of_block((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int r = gtile / (int)gridDim.x, b = (gtile - r * (int)gridDim.x) + a.tile0;      // XCD-contiguous (filter, tile) map; global tile id
    const int out0 = a.tile0 * kTile;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const size_t rowoff = (size_t)r * a.Npad;
    const double y = a.by_value ? a.y_now : a.y[a.yi];
    const int i_first = b * kTile;
    double lg[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = i_first + (k * NT + tid) * 2;
        double zn[2];
        normal_pair((uint32_t)(i0 >> 1), 0u, rep, a.key0, a.key1, &zn[0], &zn[1]);
        double xo[2], tho[kDP][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = i0 + c;
            const u32x4 o1 = philox4x32_10((uint32_t)i, 0u, rep, STREAM_LW_PRIOR, a.key0, a.key1);
            const u32x4 o2 = philox4x32_10((uint32_t)i, 0u, rep, STREAM_LW_PRIOR + 1, a.key0, a.key1);
            const double u[kDP] = {u01_co(o1.v0, o1.v1), u01_co(o1.v2, o1.v3), u01_co(o2.v0, o2.v1), u01_co(o2.v2, o2.v3)};
            double tu[kDP];
#pragma unroll
            for (int d = 0; d < kDP; ++d) { tu[d] = a.lo[d] + u[d] * (a.hi[d] - a.lo[d]); tho[d][c] = tr_fwd(a.trans[d], tu[d]); }
            xo[c] = zn[c] * (tu[2] / dsqrt(1.0 - tu[0] * tu[0]));
            lg[k][c] = lw_logg(y, xo[c], lds_etab);
            if (i >= a.N) { xo[c] = 0.0; lg[k][c] = -dinf(); for (int d = 0; d < kDP; ++d) tho[d][c] = 0.0; }
        }
        *reinterpret_cast<double2*>(a.xB + rowoff + (i0 - out0)) = make_double2(xo[0], xo[1]);
        if (a.lwB) *reinterpret_cast<double2*>(a.lwB + rowoff + (i0 - out0)) = make_double2(lg[k][0], lg[k][1]);
        th_store_pair(a.thB, rowoff + (size_t)(i0 - out0), tho);
    }
    lw_store_cdf(lg, a.N, i_first, a.cdfB + rowoff, a.tsumB + (size_t)r * a.Bs, a.tmaxB + (size_t)r * a.Bs, b, lds_d2, lds_seg_c, lds_etab, a.tile0);
}

// ---------------------------------------------------------------------------------------
// Stage 1 (t >= 1).  grid = (B, R), block = 512, dynamic LDS as k_filter_step
// ---------------------------------------------------------------------------------------
// Row totals of 16 per-lane values (the last two are zeros here) in the balanced pairwise order over the 16 lanes of a DPP row.
// Level s exchanges with the lane DPP offers as a partner -- l^1, l^2, then the mirror images l^7 and l^15 -- and a lane keeps the
// values whose index bit s equals c_s(l): c_0 = l0^l2, c_1 = l1^l2, c_2 = l2^l3, c_3 = l3 (chosen so that both partners of every
// level agree on the bits already fixed and differ in the new one).  Lane l ends with value index c_0 + 2 c_1 + 4 c_2 + 8 c_3.
__device__ __forceinline__ int lw_row_tree_index(int l) {
    const int l0 = l & 1, l1 = (l >> 1) & 1, l2 = (l >> 2) & 1, l3 = (l >> 3) & 1;
    return (l0 ^ l2) | ((l1 ^ l2) << 1) | ((l2 ^ l3) << 2) | (l3 << 3);
}
template <int CTRL>
__device__ __forceinline__ double lw_tree_level(double lo, double hi, bool keep_hi) {
    const double send = keep_hi ? lo : hi, keep = keep_hi ? hi : lo;
    return keep + dpp_f64_perm<CTRL>(send);
}
__device__ __forceinline__ double lw_row_tree14(const double (&v)[16], int tid) {
    const int l = tid & 15;
    const bool c0 = ((l ^ (l >> 2)) & 1) != 0, c1 = (((l >> 1) ^ (l >> 2)) & 1) != 0, c2 = (((l >> 2) ^ (l >> 3)) & 1) != 0, c3 = ((l >> 3) & 1) != 0;
    double y[8], z[4], w[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) y[m] = lw_tree_level<0xB1>(v[2 * m], v[2 * m + 1], c0);         // quad_perm [1,0,3,2]: pairs
#pragma unroll
    for (int n = 0; n < 4; ++n) z[n] = lw_tree_level<0x4E>(y[2 * n], y[2 * n + 1], c1);         // quad_perm [2,3,0,1]: quads
#pragma unroll
    for (int p = 0; p < 2; ++p) w[p] = lw_tree_level<0x141>(z[2 * p], z[2 * p + 1], c2);        // row_half_mirror: half rows
    return lw_tree_level<0x140>(w[0], w[1], c3);                                                // row_mirror: the row
}

template <bool BIG, bool FT = false>
__global__ __launch_bounds__(kLwNT) void k_lw_stage1(const LwArgs a) {
    constexpr int NT = kLwNT, NK = 2;
    LW_LDS_SETUP(a)
    __shared__ double lds_mom[8][4][16];          // [wave][DPP row][moment]: row totals of the 14 moment sums
    const int tid = threadIdx.x;
    const int gtile = xcd_tile_of_block((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int r = gtile / (int)gridDim.x, bloc = gtile - r * (int)gridDim.x;   // XCD-contiguous (filter, tile) map
    const int b = bloc + a.tile0;                                              // global tile id
    const int out0 = a.tile0 * kTile, win0 = a.win_tile0 * kTile;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const size_t rowoff = (size_t)r * a.Npad;
    const double y = a.by_value ? a.y_now : a.y[a.yi], z = a.by_value ? a.z_now : (a.z ? a.z[a.yi] : 0.0);
    const int i_first = b * kTile;
    const size_t gidx = ((size_t)a.gi * a.R + r) * a.B + b;
    const double G = a.gtotB[(size_t)a.gi * a.R + r];
    const double pgam_next = (b + 1 < a.B) ? a.pgamB[gidx + 1] : G;

    // resample (x, theta) by the previous second-stage weights: liu_west_filter.h:91-145 via the exact cdf -- if the
    // schedule resampled at the end of step t-1 ((t - 1 + 1) % m_rs == 0, :1139-1140); otherwise everything stays in place
    const bool resampled = (a.t % a.resamp_sched) == 0;
    int anc[NK][2];
    double mB, SB;
    L2View vB{};
    if (BIG) {
        vB.T = a.l2B_T + (size_t)r * a.Bs; vB.R = a.l2B_R + (size_t)r * a.Bs; vB.lo = a.l2B_lo + (size_t)r * a.Bs;
        vB.hi = a.l2B_hi + (size_t)r * a.Bs; vB.m = a.l2B_s[r].m; vB.S = a.l2B_s[r].S;
    }
    if (resampled) {
        // one Philox call per particle pair: words 2-3 are the pair's exponential spacings (words 0-1 unused in this draw)
        u32x4 draw[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            draw[k] = philox4x32_10((uint32_t)(b * (kTile / 2) + k * NT + tid), (uint32_t)a.t, rep, STREAM_RESAMP, a.key0, a.key1);
        lw_select<BIG>(a.tsumB + (size_t)r * a.Bs, a.tmaxB + (size_t)r * a.Bs, a.cdfB + rowoff, a.B, a.Bpow2, a.rshift, a.N, b,
                       a.gamB[gidx], a.pgamB[gidx], pgam_next, G, draw, L, anc, mB, SB,
                       a.win_tile0, vB, a.win_tiles, a.win_flag);
    } else {
        lw_level2_only<BIG>(a.tsumB + (size_t)r * a.Bs, a.tmaxB + (size_t)r * a.Bs, a.B, a.rshift, L, mB, SB, vB);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { const int i = i_first + (k * NT + tid) * 2 + c; anc[k][c] = i < a.N - 1 ? i : a.N - 1; }
        }
    }
    if (bloc == 0 && tid == 0 && a.finalize_prev) {
        // log p(y_{t-1} | y_{1:t-2}).  Auxiliary form: :1047-1051 (two log-sum-exps minus twice that of the weights the step
        // started from; :1136 at t-1 = 0); SISR form: :2257-2264 (:2301 at t-1 = 0).
        LwScalars* sc = a.scal + r;
        const double Sd = (SB > 0.0) ? dldexp(SB, -a.rshift) : dnan();
        const double lseB = mB + dlog(Sd);
        const double ll = (a.form == 0 && a.t > 1) ? (lseB + sc->lse1) - 2.0 * sc->prev : lseB - sc->prev;
        sc->mB = mB; sc->SB = SB; sc->last_ll = ll; sc->loglik = sc->loglik + ll;
        sc->prev = resampled ? a.logN : lseB;
        if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
    }
    double lg[NK][2];
    const double* xB = a.xB + rowoff;
    double fold[kNMom][2];
    // the records of all 2 NK ancestors are requested before the first one is used (20 doubles in flight per thread: the
    // kernel has the registers, and a pair's gather latency is no longer paid once per pair)
    double xall[NK][2], ttall[NK][kDP][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = anc[k][c];
            xall[k][c] = xB[j - win0];
            double trec[kDP];
            th_load(a.thB, rowoff + (size_t)(j - win0), trec);
#pragma unroll
            for (int d = 0; d < kDP; ++d) ttall[k][d][c] = trec[d];
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = i_first + (k * NT + tid) * 2;
        double xo[2], tt[kDP][2], g1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            xo[c] = xall[k][c];
#pragma unroll
            for (int d = 0; d < kDP; ++d) tt[d][c] = ttall[k][d][c];
            if (a.anc && i0 + c < a.N) a.anc[rowoff + (i0 - out0) + c] = (uint32_t)anc[k][c];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double tu[kDP];
#pragma unroll
            for (int d = 0; d < kDP; ++d) tu[d] = tr_inv(lw_trans_kind<FT>(a.trans, d), tt[d][c], lds_etab);
            // the log-weight this particle starts the step with: 0 after a resampling, else the carried second-stage weight
            const bool valid = (i0 + c) < a.N;
            const double lw_old = (resampled || !valid) ? 0.0 : a.lwB[rowoff + (i0 - out0) + c];
            // form 0: first-stage weight :985-991 = carried weight + logG(y | propMu); g1 alone is what stage 2 subtracts (:1041-1043)
            // form 1: no first stage; the carried weight travels to stage 2 in the same buffer
            g1[c] = (a.form == 0) ? lw_logg(y, lw_propmu(xo[c], z, tu, lds_etab), lds_etab) : lw_old;
            lg[k][c] = (a.form == 0) ? lw_old + g1[c] : lw_old;
            if (!valid) { xo[c] = 0.0; lg[k][c] = -dinf(); g1[c] = -dinf(); }
            // moments of the transformed parameters (:1189-1193); canonical tree: fold the tile halves (k), then the pair (c)
            int q = 0;
#pragma unroll
            for (int d = 0; d < kDP; ++d) { const double v = valid ? tt[d][c] : 0.0; fold[q][c] = (k == 0) ? v : fold[q][c] + v; ++q; }
#pragma unroll
            for (int d = 0; d < kDP; ++d) {
#pragma unroll
                for (int e = 0; e <= d; ++e) { const double v = valid ? tt[d][c] * tt[e][c] : 0.0; fold[q][c] = (k == 0) ? v : fold[q][c] + v; ++q; }
            }
        }
        *reinterpret_cast<double2*>(a.lw1 + rowoff + (i0 - out0)) = make_double2(g1[0], g1[1]);
        if (a.compose) {
            *reinterpret_cast<uint2*>(a.ancbuf + rowoff + (i0 - out0)) = make_uint2((uint32_t)anc[k][0], (uint32_t)anc[k][1]);
        } else {
            *reinterpret_cast<double2*>(a.xr + rowoff + (i0 - out0)) = make_double2(xo[0], xo[1]);
            th_store_pair(a.thr, rowoff + (size_t)(i0 - out0), tt);
        }
    }
    // wave tree per 128-element segment of the folded half tile, then the 8 segments in order.  The tree is the balanced pairwise
    // one that the last lane of wave_incl_scan_f64 holds -- ((l0 + l1) + (l2 + l3)) + ... per 16-lane row, then (r0 + r1) + (r2 + r3)
    // -- but fourteen full wave scans were a tenth of the kernel's instructions.  lw_row_tree14 takes all fourteen sums through the
    // SAME tree at once: at each of its four levels a lane keeps half of its values and hands the other half to its partner, so
    // the work halves per level (about 100 instructions instead of 250) and lane j of a row ends with the row total of moment
    // lw_row_tree_index(j).  The four row totals of a wave meet in the final sum below.  Same additions, same bits.
    {
        double v[16];
#pragma unroll
        for (int q = 0; q < kNMom; ++q) v[q] = fold[q][0] + fold[q][1];
        v[14] = 0.0; v[15] = 0.0;
        const double rowtot = lw_row_tree14(v, tid);
        lds_mom[tid >> 6][(tid >> 4) & 3][lw_row_tree_index(tid & 15)] = rowtot;
    }
    if (a.form == 0) lw_store_cdf(lg, a.N, i_first, a.cdfA + rowoff, a.tsumA + (size_t)r * a.Bs, a.tmaxA + (size_t)r * a.Bs, b, lds_d2, lds_seg_c, lds_etab, a.tile0);
    else __syncthreads();
    // (a barrier after the lds_mom writes either way)
    if (tid < kNMom) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) s = s + ((lds_mom[w][3][tid] + lds_mom[w][2][tid]) + (lds_mom[w][1][tid] + lds_mom[w][0][tid]));
        a.mom[((size_t)r * gridDim.x + bloc) * 16 + tid] = s;       // gridDim.x = B unless the filter is sharded
    }
}

// theta-bar and the Cholesky factor of (1 - a^2) V from the 14 moment totals (:1184-1198); p[0..3] = theta-bar, p[4..13] = L
// (lower triangle, row-major).  One thread.
__device__ __forceinline__ void lw_proposal_components(const double* sums, int N, double a_shrink, double* p) {
    const double invN = 1.0 / (double)N;
    double tb[kDP], Sig[kDP][kDP], Lc[kDP][kDP];
    for (int d = 0; d < kDP; ++d) tb[d] = sums[d] * invN;
    const double h2 = 1.0 - a_shrink * a_shrink;
    int q = kDP;
    for (int d = 0; d < kDP; ++d) for (int e = 0; e <= d; ++e) { Sig[d][e] = h2 * (sums[q] * invN - tb[d] * tb[e]); ++q; }
    for (int d = 0; d < kDP; ++d) for (int e = 0; e < kDP; ++e) Lc[d][e] = 0.0;
    for (int j = 0; j < kDP; ++j) {
        double sdiag = Sig[j][j];
        for (int k = 0; k < j; ++k) sdiag = sdiag - Lc[j][k] * Lc[j][k];
        Lc[j][j] = (sdiag > 0.0) ? dsqrt(sdiag) : 0.0;
        for (int i = j + 1; i < kDP; ++i) {
            double v = Sig[i][j];
            for (int k = 0; k < j; ++k) v = v - Lc[i][k] * Lc[j][k];
            Lc[i][j] = (Lc[j][j] > 0.0) ? v / Lc[j][j] : 0.0;
        }
    }
    for (int d = 0; d < kDP; ++d) p[d] = tb[d];
    q = kDP;
    for (int d = 0; d < kDP; ++d) for (int e = 0; e <= d; ++e) p[q++] = Lc[d][e];
}

// ---------------------------------------------------------------------------------------
// Moment totals for filters of many tiles: one WAVE per (moment, filter) instead of one workgroup per filter for all 14
// (k_lw_mid is a single workgroup bound by memory round trips: 81 us at 8192 tiles).  Same order of additions as k_lw_mid:
// lane l adds its contiguous chunk of tiles in order, then the wave tree.  grid = (14, R), block = 64.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lw_mom_totals(const LwArgs a) {
    const int lane = threadIdx.x, q = blockIdx.x, r = blockIdx.y;
    const int c = (a.B + 63) / 64;
    const int j0 = lane * c, j1 = ((lane + 1) * c < a.B) ? (lane + 1) * c : a.B;
    double acc = 0.0;
    for (int j = j0; j < j1; j += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (j + u < j1) ? a.mom[((size_t)r * a.B + j + u) * 16 + q] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) if (j + u < j1) acc = acc + v[u];
    }
    const double sw = wave_incl_scan_f64(acc);
    if (lane == 63) a.momtot[(size_t)r * 16 + q] = sw;
}

// ---------------------------------------------------------------------------------------
// Mid: proposal components (:1184-1198) + log-sum-exp of the first-stage weights.  grid = (R), block = 256
// ---------------------------------------------------------------------------------------
template <bool BIG>
__global__ __launch_bounds__(kThreads) void k_lw_mid(const LwArgs a) {
    __shared__ double lds_seg[128];
    __shared__ double lds_d[16];
    __shared__ double sums[kNMom];
    const int tid = threadIdx.x, r = blockIdx.x;
    double A2[8], Ap[8], Tinc[8], M2[8], S, m;
    const bool aux = a.form == 0;                  // the SISR form has no first-stage weights
    if (!BIG && aux) level2_load<kThreads>(a.tsumA + (size_t)r * a.Bs, a.tmaxA + (size_t)r * a.Bs, a.B, A2, M2);
    // moment totals: wave w handles moments w, w+4, ...: 64 lanes add contiguous chunks of tiles in order, then the wave tree
    const int lane = tid & 63, wave = tid >> 6;
    const int c = (a.B + 63) / 64;
    // wave w sums moments w, w+4, w+8, w+12; for each, lane l adds its chunk of tiles in order and the wave tree adds the
    // 64 chunk sums.  The loads of all of a wave's moments are issued together (8 tiles x 4 moments per batch) so that
    // one memory latency is paid per batch, not per moment; the order of the additions is that of a plain loop.
    if (a.momtot) {
        if (tid < kNMom) sums[tid] = a.momtot[(size_t)r * 16 + tid];
    } else {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const int j0 = lane * c, j1 = ((lane + 1) * c < a.B) ? (lane + 1) * c : a.B;
        for (int j = j0; j < j1; j += 8) {
            double v[4][8];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = wave + 4 * qi;
#pragma unroll
                for (int u = 0; u < 8; ++u) v[qi][u] = (q < kNMom && j + u < j1) ? a.mom[((size_t)r * a.B + j + u) * 16 + q] : 0.0;
            }
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j + u < j1) acc[qi] = acc[qi] + v[qi][u];
            }
        }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int q = wave + 4 * qi;
            const double sw = wave_incl_scan_f64(acc[qi]);
            if (lane == 63 && q < kNMom) sums[q] = sw;
        }
    }
    m = 0.0; S = 0.0;
    if (aux) {
        if (BIG) { m = a.l2A_s[r].m; S = a.l2A_s[r].S; }
        else level2_scan<kThreads>(A2, M2, a.B, a.rshift, m, Ap, Tinc, S, lds_d, lds_seg, kExpTable);
    }
    __syncthreads();
    if (tid == 0) {
        LwScalars* sc = a.scal + r;
        const double Sd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
        if (aux) sc->lse1 = m + dlog(Sd);
        lw_proposal_components(sums, a.N, a.a_shrink, a.prop + (size_t)r * 16);
    }
}

// ---------------------------------------------------------------------------------------
// Stage 2 (t >= 1).  grid = (B, R), block = 512
// ---------------------------------------------------------------------------------------
template <bool BIG, bool FT = false>
__global__ __launch_bounds__(kLwNT) void k_lw_stage2(const LwArgs a) {
    constexpr int NT = kLwNT, NK = 2;
    LW_LDS_SETUP(a)
    const int tid = threadIdx.x;
    const int gtile = xcd_tile_of_block((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int r = gtile / (int)gridDim.x, b = (gtile - r * (int)gridDim.x) + a.tile0;      // XCD-contiguous (filter, tile) map; global tile id
    const int out0 = a.tile0 * kTile, win0 = a.win_tile0 * kTile;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const size_t rowoff = (size_t)r * a.Npad;
    const double y = a.by_value ? a.y_now : a.y[a.yi], z = a.by_value ? a.z_now : (a.z ? a.z[a.yi] : 0.0);
    const int i_first = b * kTile;
    const size_t gidx = ((size_t)a.gi * a.R + r) * a.B + b;
    const double G = a.gtotA[(size_t)a.gi * a.R + r];
    const double pgam_next = (b + 1 < a.B) ? a.pgamA[gidx + 1] : G;
    // theta-bar and the Cholesky factor: from k_lw_mid, or (fuse_mid: filters of at most 512 tiles, not sharded) taken here by
    // every workgroup from the tile partials, so that the step has no one-workgroup launch between its two stages
    // (9.3 of 83 us per step at N = 2^20).  The totals are added in k_lw_mid's order: 64 lanes add contiguous chunks of
    // tiles in order, then the wave tree; wave w takes moments w and w + 8.
    __shared__ double lds_sums[16];
    __shared__ double lds_prop[16];
    const bool fuse = !BIG && a.fuse_mid;
    double prop[16];
    if (!fuse) {
#pragma unroll
        for (int q = 0; q < 14; ++q) prop[q] = a.prop[(size_t)r * 16 + q];
    } else {
        // the tile partials ([B][16] doubles, 14 used) come in through LDS: one coalesced pass into the not yet used window
        // area (B <= 512 tiles x 14 doubles = at most the 56 KB of lds_T + lds_R + three staged tiles), then every lane
        // walks its chunk there.  (Read straight from global memory each lane touches its own cache line per tile:
        // 8192 line accesses per workgroup, +8 us per launch.)
        double* lds_momall = L.lds_stage;        // the start of the dynamic LDS: staged tiles | T' | A/A'
        const double* gm = a.mom + (size_t)r * a.B * 16;
        {
            // all loads of a thread are issued before its first LDS store (a rolled loop waits for every load in turn);
            // B * 14 doubles fit the window area (host check), i.e. B * 7 <= 8 * NT pairs
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int it = tid + u * NT, j = it / 7, p2 = it - j * 7;
                v[u] = make_double2(0.0, 0.0);
                if (it < a.B * 7) v[u] = *reinterpret_cast<const double2*>(gm + (size_t)j * 16 + 2 * p2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int it = tid + u * NT, j = it / 7, p2 = it - j * 7;
                if (it < a.B * 7) *reinterpret_cast<double2*>(lds_momall + j * kNMom + 2 * p2) = v[u];
            }
        }
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6;
        const int cch = (a.B + 63) / 64;
        const int j0 = lane * cch, j1 = ((lane + 1) * cch < a.B) ? (lane + 1) * cch : a.B;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = wave + 8 * qi;
            double acc = 0.0;
            if (q < kNMom) for (int j = j0; j < j1; ++j) acc = acc + lds_momall[j * kNMom + q];
            const double sw = wave_incl_scan_f64(acc);
            if (lane == 63 && q < kNMom) lds_sums[q] = sw;
        }
        __syncthreads();                                   // the window area is free again (lw_select fills it next)
    }

    // form 0: k ~ Categorical(first-stage weights): k_gen.sample, :1006 (every step, whatever the resampling schedule);
    // form 1: every particle continues itself (:2206-2235)
    int kk[NK][2];
    double mA = 0.0, SA = 0.0;
    L2View vA{};
    if (BIG) {
        vA.T = a.l2A_T + (size_t)r * a.Bs; vA.R = a.l2A_R + (size_t)r * a.Bs; vA.lo = a.l2A_lo + (size_t)r * a.Bs;
        vA.hi = a.l2A_hi + (size_t)r * a.Bs; vA.m = a.l2A_s[r].m; vA.S = a.l2A_s[r].S;
    }
    // one Philox call per particle pair (counter (pair, t, filter, STREAM_LW_K)): words 0-1 -> the pair's two state normals of
    // fSamp (Box-Muller as in the bootstrap kernel: 40-bit radius uniform, 24-bit angle), words 2-3 -> the k draw's spacings
    u32x4 draw[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k)
        draw[k] = philox4x32_10((uint32_t)(b * (kTile / 2) + k * NT + tid), (uint32_t)a.t, rep, STREAM_LW_K, a.key0, a.key1);
    if (a.form == 0) {
        lw_select<BIG>(a.tsumA + (size_t)r * a.Bs, a.tmaxA + (size_t)r * a.Bs, a.cdfA + rowoff, a.B, a.Bpow2, a.rshift, a.N, b,
                       a.gamA[gidx], a.pgamA[gidx], pgam_next, G, draw, L, kk, mA, SA,
                       a.win_tile0, vA, a.win_tiles, a.win_flag);
    } else {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { const int i = i_first + (k * NT + tid) * 2 + c; kk[k][c] = i < a.N - 1 ? i : a.N - 1; }
        }
    }
    if (fuse) {
        if (tid == 0) {
            lw_proposal_components(lds_sums, a.N, a.a_shrink, lds_prop);
            if (b == a.tile0) {                            // the filter's first workgroup keeps k_lw_mid's records
                double* p = a.prop + (size_t)r * 16;
                for (int q = 0; q < 14; ++q) p[q] = lds_prop[q];
                if (a.form == 0) {
                    const double Sd = (SA > 0.0) ? dldexp(SA, -a.rshift) : dnan();
                    a.scal[r].lse1 = mA + dlog(Sd);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 14; ++q) prop[q] = readfirstlane_f64(lds_prop[q]);
    }
    double lg[NK][2];
    // The sources of the thread's 2 NK particles are two dependent gathers each (k -> the ancestor it composes with -> state,
    // parameters).  All ancestor indices are requested at once; the records are then software-pipelined: particle p + 1's three
    // loads are in flight while particle p's ~800 instructions run (all of them up front would need 12 more registers per
    // particle and the kernel would lose a workgroup per CU).
    const double* xsrc = a.compose ? a.xB : a.xr;
    const double* thsrc = a.compose ? a.thB : a.thr;
    int jsrc[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) jsrc[k][c] = a.compose ? (int)a.ancbuf[rowoff + kk[k][c]] : kk[k][c];
    }
    double nx_x, nx_lw1, nx_th[kDP];
    auto request = [&](int k, int c) {
        nx_x = xsrc[rowoff + (jsrc[k][c] - win0)];
        nx_lw1 = a.lw1[rowoff + (kk[k][c] - win0)];
        th_load(thsrc, rowoff + (size_t)(jsrc[k][c] - win0), nx_th);
    };
    request(0, 0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = i_first + (k * NT + tid) * 2;
        double zs[2];
        pair_normals(draw[k].v0, draw[k].v1, &lds_dtab, &zs[0], &zs[1]);
        double xo[2], tho[kDP][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = i0 + c;
            const int j = kk[k][c];
            const double xk = nx_x, lw1k = nx_lw1;
            double thrk[kDP];
#pragma unroll
            for (int d = 0; d < kDP; ++d) thrk[d] = nx_th[d];
            if (c == 0) request(k, 1);
            else if (k + 1 < NK) request(k + 1, 0);
            double e[kDP];
            {
                // the four jitter normals of a particle from ONE call: two Box-Muller pairs, (words 0-1) and (words 2-3)
                const u32x4 o1 = philox4x32_10((uint32_t)i, (uint32_t)a.t, rep, STREAM_LW_JIT, a.key0, a.key1);
                pair_normals(o1.v0, o1.v1, &lds_dtab, &e[0], &e[1]);
                pair_normals(o1.v2, o1.v3, &lds_dtab, &e[2], &e[3]);
            }
            double tu[kDP];
            int q = kDP;
#pragma unroll
            for (int d = 0; d < kDP; ++d) {
                const double thk = thrk[d];
                const double mm = a.a_shrink * thk + (1.0 - a.a_shrink) * prop[d];      // :1024
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w <= d; ++w) { acc = acc + prop[q] * e[w]; ++q; }
                tho[d][c] = mm + acc;                                                    // MVN draw, :1026-1027
                tu[d] = tr_inv(lw_trans_kind<FT>(a.trans, d), tho[d][c], lds_etab);
            }
            const double mean = (tu[1] + tu[0] * (xk - tu[1])) + ((z * tu[3]) * tu[2]) * dexp_scaled_t(-0.5 * xk, 0, lds_etab);       // fSamp :114-121
            xo[c] = mean + zs[c] * (tu[2] * dsqrt(1.0 - tu[3] * tu[3]));
            // form 0: logG(y | x', theta') - logG(y | propMu_k, m_k)  (:1041-1043);  form 1: carried weight + logG  (:2223-2225,
            // where logFEv - logQEv vanishes: svol_lw_2_par proposes from the transition)
            lg[k][c] = (a.form == 0) ? lw_logg(y, xo[c], lds_etab) - lw1k : lw1k + lw_logg(y, xo[c], lds_etab);
            if (a.kidx && i < a.N) a.kidx[rowoff + (i - out0)] = (uint32_t)j;
            if (i >= a.N) { xo[c] = 0.0; lg[k][c] = -dinf(); }
        }
        double* xdst = a.compose ? a.xr : a.xB;          // compose: the other population pair (the host swaps them after the launch)
        double* thdst = a.compose ? a.thr : a.thB;
        *reinterpret_cast<double2*>(xdst + rowoff + (i0 - out0)) = make_double2(xo[0], xo[1]);
        if (a.lwB) *reinterpret_cast<double2*>(a.lwB + rowoff + (i0 - out0)) = make_double2(lg[k][0], lg[k][1]);
        th_store_pair(thdst, rowoff + (size_t)(i0 - out0), tho);
    }
    lw_store_cdf(lg, a.N, i_first, a.cdfB + rowoff, a.tsumB + (size_t)r * a.Bs, a.tmaxB + (size_t)r * a.Bs, b, lds_d2, lds_seg_c, lds_etab, a.tile0);
}

// ---------------------------------------------------------------------------------------
// Accounts the last step's log conditional likelihood.  grid = (R), block = 256
// ---------------------------------------------------------------------------------------
template <bool BIG>
__global__ __launch_bounds__(kThreads) void k_lw_finalize(const LwArgs a) {
    __shared__ double lds_seg[128];
    __shared__ double lds_d[16];
    const int tid = threadIdx.x, r = blockIdx.x;
    double A2[8], Ap[8], Tinc[8], M2[8], S, m;
    if (BIG) { m = a.l2B_s[r].m; S = a.l2B_s[r].S; }
    else {
        level2_load<kThreads>(a.tsumB + (size_t)r * a.Bs, a.tmaxB + (size_t)r * a.Bs, a.B, A2, M2);
        level2_scan<kThreads>(A2, M2, a.B, a.rshift, m, Ap, Tinc, S, lds_d, lds_seg, kExpTable);
    }
    if (tid == 0) {
        LwScalars* sc = a.scal + r;
        const double Sd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
        const double lseB = m + dlog(Sd);
        const double ll = (a.form == 0 && a.t > 0) ? (lseB + sc->lse1) - 2.0 * sc->prev : lseB - sc->prev;
        sc->mB = m; sc->SB = S; sc->last_ll = ll; sc->loglik = sc->loglik + ll;
        sc->prev = ((a.t + 1) % a.resamp_sched == 0) ? a.logN : lseB;
        if (a.per_step) a.per_step[(size_t)r * a.Tcap + a.t] = ll;
        if (a.ll_host) a.ll_host[r] = ll;
    }
}

// Weighted expectations under the last second-stage (pre-resampling) weights, two launches (getExpectations(),
// liu_west_filter.h:1054-1075 / :2267-2290, for built-in functionals):
// k_lw_param_partials, grid = (B tiles, R): per-tile sums of w, w theta_d (untransformed), w x, w x^2, w exp(x/2) with
// w = q_j exp(m_tile - m), into the moment scratch mom[r][b][0..7]; k_lw_param_means, grid = (R): adds the tile partials
// in tile order and divides: out[r][0..3] = E[theta_d], [4] = E[x], [5] = E[x^2], [6] = E[exp(x/2)], [7] = E[42].
constexpr int kLwNExp = 8;
__global__ __launch_bounds__(kThreads) void k_lw_param_partials(const LwArgs a) {
    __shared__ double lds_n[4][kLwNExp];
    __shared__ double lds_m[16];
    const int tid = threadIdx.x, b = blockIdx.x, r = blockIdx.y;
    double mx = -dinf();
    bool nan = false;
    for (int j = tid; j < a.B; j += kThreads) { const double v = a.tmaxB[(size_t)r * a.Bs + j]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    const double m = block_max_nanprop<kThreads>(mx, nan, lds_m);
    const double scale = dexp(a.tmaxB[(size_t)r * a.Bs + b] - m);
    double acc[kLwNExp] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int i_end = ((b + 1) * kTile < a.N) ? (b + 1) * kTile : a.N;
    for (int i = b * kTile + tid; i < i_end; i += kThreads) {
        const double c1 = a.cdfB[(size_t)r * a.Npad + i];
        const double c0 = (i & (kTile - 1)) ? a.cdfB[(size_t)r * a.Npad + i - 1] : 0.0;
        const double w = (c1 - c0) * scale;
        const double xv = a.xB[(size_t)r * a.Npad + i];
        acc[kDP] += w;
        double trec[kDP];
        th_load(a.thB, (size_t)r * a.Npad + i, trec);
        for (int d = 0; d < kDP; ++d) acc[d] += w * tr_inv(a.trans[d], trec[d], kExpTable);
        acc[5] += w * xv;
        acc[6] += w * (xv * xv);
        acc[7] += w * dexp(0.5 * xv);
    }
    for (int q = 0; q < kLwNExp; ++q) {
        double v = acc[q];
        for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, kWave);
        if ((tid & 63) == 0) lds_n[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid < kLwNExp) a.mom[((size_t)r * a.B + b) * 16 + tid] = ((lds_n[0][tid] + lds_n[1][tid]) + lds_n[2][tid]) + lds_n[3][tid];
}

__global__ __launch_bounds__(kWave) void k_lw_param_means(const LwArgs a, double* out /*[R][8]*/) {
    const int r = blockIdx.x, q = threadIdx.x;
    double t = 0.0;
    if (q < kLwNExp) {
        // tile order, as a plain loop would add them; the loads of 16 tiles are issued together (one memory latency per
        // batch instead of one per tile: 115 us at 512 tiles otherwise)
        for (int b0 = 0; b0 < a.B; b0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (b0 + u < a.B) ? a.mom[((size_t)r * a.B + b0 + u) * 16 + q] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) if (b0 + u < a.B) t = t + v[u];
        }
    }
    const double den = __shfl(t, kDP, kWave);
    if (q < kDP) out[(size_t)r * kLwNExp + q] = t / den;
    else if (q == kDP) out[(size_t)r * kLwNExp + 7] = 42.0 * (den / den);          // NaN weights stay NaN
    else if (q < kLwNExp) out[(size_t)r * kLwNExp + q - 1] = t / den;               // x, x^2, exp(x/2) -> slots 4, 5, 6
}

// weights w_j = q_j exp(m_tile - m) 2^-41 and UNTRANSFORMED parameters of one filter, for host-side functionals.
// grid = (B), block = 256.  out: w[Npad], theta[4][Npad]
__global__ __launch_bounds__(kThreads) void k_lw_weights(const LwArgs a, int r, double* out) {
    __shared__ double lds_m[16];
    const int tid = threadIdx.x, b = blockIdx.x;
    double mx = -dinf();
    bool nan = false;
    for (int j = tid; j < a.B; j += kThreads) { const double v = a.tmaxB[(size_t)r * a.Bs + j]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    const double m = block_max_nanprop<kThreads>(mx, nan, lds_m);
    const double sc = (m != m) ? dnan() : dexp_scaled(a.tmaxB[(size_t)r * a.Bs + b] - m, -kTileShift);
    for (int j = tid; j < kTile; j += kThreads) {
        const int i = b * kTile + j;
        if (i >= a.N) break;
        const double c1 = a.cdfB[(size_t)r * a.Npad + i];
        const double c0 = j ? a.cdfB[(size_t)r * a.Npad + i - 1] : 0.0;
        out[i] = (c1 - c0) * sc;
        double trec[kDP];
        th_load(a.thB, (size_t)r * a.Npad + i, trec);
        for (int d = 0; d < kDP; ++d) out[(size_t)(1 + d) * a.Npad + i] = tr_inv(a.trans[d], trec[d], kExpTable);
    }
}

}  // namespace ssme
