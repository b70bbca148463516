// pf_kernels.h -- HIP kernels of the bootstrap-particle-filter step for gfx950 (wave64).
//
// One filter step = ONE kernel, k_filter_step.  Per 2048-particle tile:
//   level-2: rescale + exact scan of the previous step's tile sums (per-tile weight scales, so no
//            global max pass and no separate normalisation kernel)       twin liu_west_filter.h:1652-1659
//   resampling targets (sorted uniforms by exponential spacings)          twin :105-139
//   search of the previous step's integer weight cdf, tiles staged in LDS  twin :112-139
//   gather ancestor state, fSamp, logGEv                                  example/univ_svol_bootstrap_filter.h:74-86
//   tile max, q = rne(exp(logw - max_tile) 2^41), exact integer tile scan  twin :96-104
// It replaces pf::BSFilter::filter (call site example/estimate_univ_svol.h:124).
//
// Particles are a structure of arrays in HBM: x[R][Npad] (fp64, ping-pong), cdf[R][Npad] (fixed point:
// integers < 2^53 held exactly in fp64, ping-pong), per-tile sums/maxima; one row per filter.  A tile is 2048 consecutive
// particles = NT threads x (1024/NT) pairs, so every streaming access is one 16-byte load/store
// per lane, fully coalesced.  Log-weights never leave registers (unless resampling is not every
// step, or in debug mode).
//
// The weight cdf is EXACT integer arithmetic: every value is an integer below
// 2^53 carried in an fp64 register, so v_add_f64 adds exactly; sums are associative, the cdf is
// monotone by construction and the ancestor of a target is an integer count -- independent of scan
// tree, search strategy and block shape.
#pragma once
#include "ssme_math.h"
#include "model_api.h"

namespace ssme {

typedef unsigned long long u64;

constexpr int kThreads = 256;            // helper kernels; KA/KR are templated on their block size
constexpr int kWave = 64;
constexpr int kRow = 512;
constexpr int kTile = 2048;                // large tile: N <= 2048 (one-tile series kernel), N > 2^18, Liu-West, sharded filters
constexpr int kTileSmall = 512;
constexpr int kTileMid = 1024;            // small tile: 2048 < N <= 2^18, so that a mid-size filter spreads over the chip (N = 2^16: 128 workgroups)
constexpr int kMaxTilesPerFilter = 2048;   // in-kernel level-2 (one entry per thread at NT = 512 .. four at 512 threads)
constexpr int kSplitLevel2Above = 1024;    // measured (profiles/r02_level2_split.txt): in-kernel 24.7 vs split 25.4 us at 768 tiles, equal at 1024, 67.9 vs 43.3 at 1536
constexpr int kL2Scratch = 64;             // doubles of l2_work per filter after the block-local scans: [0, 16) block totals, [16] m,
constexpr int kL2Offsets = 32;             //   [kL2Offsets, + 17) exclusive offsets of the blocks and S' (l2_inkernel)
constexpr int kMaxTilesSplit = 16384;      // split level-2 (k_level2_plan + k_filter_step<.., true>): N <= 2^25
constexpr int kStageTiles = 3;             // cdf tiles staged in LDS per output tile
constexpr int kEShift = 35;                // exponential spacings: qE = rne(E * 2^35)
constexpr int kTileShift = 41;             // tile-local fixed point: q = rne(exp(logw - m_tile) * 2^41), tile sums <= 2^52

enum { RESAMP_MULTINOMIAL = 0, RESAMP_SYSTEMATIC = 1, RESAMP_STRATIFIED = 2, RESAMP_MULTINOMIAL_IID = 3 };

#define SSME_HALF_LOG_2PI 0.91893853320467274178

#ifdef SSME_ABLATE
#define ABL(a, bit) (((a).ablate >> (bit)) & 1)
// diagnostic phase stamps (100 MHz constant clock), one row of 16 per block; never in the product build
#define STAMP(a, i) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ABL(a, bit) 0
#define STAMP(a, i) do {} while (0)
#endif

// Per-filter scalars living in device memory.
struct FilterScalars {
    double m;        // max log-weight of the last step (NaN if any log-weight is NaN)
    double S;        // exact integer sum of the rescaled tile sums of the last step
    double prev;     // m_old + log S_old (log N after a resampling step)
    double loglik;   // running sum of log p(y_t | y_{1:t-1})
    double last_ll;  // last log conditional likelihood
    double pad[3];
};

struct StepArgs {
    const double* x_in;        // [R][Npad] particles of step t-1 (pre-resampling)
    double* x_out;             // [R][Npad]
    double* logw;              // [R][Npad] or null: only kept when resampling is not every step, or in debug mode
    const double* cdf_in;      // [R][Npad] tile-local inclusive integer sums of q, step t-1
    double* cdf_out;           // [R][Npad] step t
    const double* tsum_in;     // [R][Bs] tile sums A_b (tile scale), step t-1
    double* tsum_out;
    const double* tmax_in;     // [R][Bs] tile maxima m_b, step t-1
    double* tmax_out;
    uint32_t* anc;             // [R][Npad] or null
    FilterScalars* scal;       // [R]
    const ModelConst* mc;      // [R]
    const double* y;           // [T]
    const double* z;           // [T] or null
    double y_now, z_now;       // step API: the observation and covariate of THIS call travel in the kernel arguments (by_value = 1)
    double y_now_v[3];         // ... components 1 .. 3 of a vector observation (user models with dim_y > 1, model_api.h)
    size_t xplane;             // user models with dim_x > 1: component d of the particles lives at x_in / x_out + d * xplane
    int32_t by_value;          // 1: use y_now / z_now instead of y[yi] / z[yi] (no upload, no memory read)
    double* ll_host;           // step API: host-mapped buffer that receives the R log conditional likelihoods from the accounting kernel, or null
    double* per_step;          // [R][Tcap] or null
    double* small_ms;          // [R][Tcap][2] scratch of k_filter_series_lane: (m_t, S_t), then lse_t
    const double* gam;         // [nT][R][B] Gamma(n_b) draws          (multinomial)
    const double* pgam;        // [nT][R][B] exclusive prefixes of gam
    const double* gtot;        // [nT][R]    sum(gam) + E_{N+1}
    int32_t N, Npad, B, Bs, Bpow2, rshift, R;
    int32_t tile;              // particles per tile (2048, 1024 or 512): part of the arithmetic specification
    int32_t t, yi, gi, Tcap;   // t: time index (RNG counter, schedule); yi / gi: rows of y / gamma tables
    int32_t resampler, resamp_sched;
    int32_t finalize_prev;     // account log p(y_{t-1}|.) of the previous step
    int32_t tile0;             // particle-sharded filter: global id of this launch's first OUTPUT tile (0 otherwise); outputs
                               // are stored at local offsets (tile - tile0)
    int32_t win_tile0;         // global id of the first SOURCE tile held in x_in / cdf_in (0 otherwise)
    int32_t win_tiles;         // C++ shard driver, fixed-halo path: tiles held in the source window (0: unchecked)
    int32_t* win_flag;         // ... and where to record that some output tile's sources left it ([0] flag), or null
    int32_t* ticket;           // step API: [R] arrival counters; the last workgroup of a filter accounts the step in the same launch (null: kf_finalize does)
    // split level-2 (filters of more than 2048 tiles, or forced for tests): written by k_level2_plan, read by k_filter_step<..,true>
    double* l2_T;              // [R][Bs] inclusive prefixes T'_b of the rescaled tile sums
    double* l2_R;              // [R][Bs] A_b / A'_b
    int32_t* l2_lo;            // [R][Bs] first / last source tile of every output tile's targets
    int32_t* l2_hi;
    double* l2_work;           // [R][Bs] block-local scans + [R][64] per filter: [0,16) block totals, [16] m, [17] S', [32,49) exclusive
                               // offsets of the blocks: scratch of the multi-workgroup level-2 (k_l2_scan_blocks / k_l2_ranges,
                               // filters of more than 1024 tiles), or null
    int32_t l2_inkernel;       // 1 (round 3, unsharded bootstrap filters of more than 1024 tiles): the level-2 is ONE launch --
                               // k_l2_scan_blocks, whose last-arriving workgroup takes the block offsets, S' and the accounting --
                               // and k_filter_step<.., true> finds its own source-tile range in T'_j = offset[j / 1024] + local
                               // scan[j] (a 64-entry window around its own tile id, binary search beyond it): no k_l2_ranges launch
    int32_t* l2_ticket;        // [R] arrival counters of that launch (zero between launches)
    const double* l2_tsrc;     // k_filter_step<.., true> reads T'_j = l2_offs[r * l2_offs_stride + j / 1024] + l2_tsrc[r * Bs + j]:
    const double* l2_offs;     //   l2_T and zeros (stride 0) after the table kernels, l2_work's scans and offsets (stride 64) with l2_inkernel
    int32_t l2_offs_stride;
    int32_t prio_mode;         // wave-priority schedule of k_filter_step (prio_at): 0 none, 1 single residency wave, 2 several
    int32_t stream_stores;     // 1: particles and cdf are stored non-temporally (grids that are resident all at once)
    const uint32_t* keyp;      // [2] Philox key (the seed), device resident so that a captured graph survives ssme_pf_set_seed
    uint32_t first_filter;
    double logN;
    int32_t ablate;            // measurement builds only (-DSSME_ABLATE): skip sections, results invalid
    unsigned long long* stamps;  // measurement builds only: phase time stamps
};

// ---------------------------------------------------------------------------------------
// DPP wave primitives.  update_dpp(old = fill, ...) with bound_ctrl = 0: lanes whose source is
// out of range, or whose row is masked off, receive `fill`.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64_neginf(double v) {    // fill = -inf
    const u64 b = d2bits(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)0xfff00000, (int)(uint32_t)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return bits2d(((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64_zero(double v) {      // fill = +0.0
    const u64 b = d2bits(v);
    int lo, hi;
    if (ROWMASK == 0xF) {
        // all rows enabled: a lane either has a source lane or is out of range, and bound_ctrl supplies the 0 --
        // no `old` operand, so the compiler need not zero the destination before every DPP move
        lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROWMASK, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROWMASK, 0xF, false);
    }
    return bits2d(((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo);
}
// inclusive wave scan of integer-valued doubles (exact while every partial sum < 2^53)
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
    v = v + dpp_f64_zero<0x111, 0xF>(v);
    v = v + dpp_f64_zero<0x112, 0xF>(v);
    v = v + dpp_f64_zero<0x114, 0xF>(v);
    v = v + dpp_f64_zero<0x118, 0xF>(v);
    v = v + dpp_f64_zero<0x142, 0xA>(v);
    v = v + dpp_f64_zero<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ double wave_shr1_f64(double v) { return dpp_f64_zero<0x138, 0xF>(v); }
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const u64 b = d2bits(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return bits2d(((u64)hi << 32) | lo);
}
__device__ __forceinline__ double readfirstlane_f64(double v) {       // a value every lane holds -> scalar registers
    const u64 b = d2bits(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return bits2d(((u64)hi << 32) | lo);
}

// row_shr:n = 0x110+n ; row_bcast:15 = 0x142 ; row_bcast:31 = 0x143 ; wave_shr:1 = 0x138
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// max over the wave of non-NaN doubles (lane 63 holds it after the scan; broadcast by readlane)
// Inside a row of 16 lanes the reduction is a BUTTERFLY (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): every lane has
// a source lane, so the DPP moves need no fill value -- 3 instructions per step instead of 5 with the -inf fill of a shifted
// scan; only the two cross-row steps (row_bcast:15 / :31) keep it.  A maximum does not depend on the order: same bits.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_perm(double v) {       // full permutations inside a row: no fill needed
    const u64 b = d2bits(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return bits2d(((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = dmaxnum(v, dpp_f64_perm<0xB1>(v));            // quad_perm [1,0,3,2]
    v = dmaxnum(v, dpp_f64_perm<0x4E>(v));            // quad_perm [2,3,0,1]
    v = dmaxnum(v, dpp_f64_perm<0x141>(v));           // row_half_mirror
    v = dmaxnum(v, dpp_f64_perm<0x140>(v));           // row_mirror: every lane of a row holds the row's maximum
    v = dmaxnum(v, dpp_f64_neginf<0x142, 0xA>(v));
    v = dmaxnum(v, dpp_f64_neginf<0x143, 0xC>(v));
    return bits2d(readlane_u64(d2bits(v), 63));
}

// Block max with NaN propagation: returns NaN if any thread passes nan = true.
// lds: NT/64 doubles, not reused by the caller before its next barrier.  One barrier.
template <int NT>
__device__ __forceinline__ double block_max_nanprop(double m, bool nan, double* lds) {
    m = wave_max_f64(m);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    const int any_nan = __syncthreads_or(nan ? 1 : 0);
    double r = lds[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = dmaxnum(r, lds[w]);
    return any_nan ? dnan() : r;
}

// ---------------------------------------------------------------------------------------
// Exact inclusive scan of P = 2*NT*NK integer-valued doubles by a block of NT threads; every partial sum < 2^53, so each
// v_add_f64 is exact and the sums are associative: any wave/segment decomposition gives the same.
// Thread tid holds NK pairs: q[k][c] = value[(k*NT + tid)*2 + c].
// incl[k][c] = sum of all values up to and including that position; total = sum of all.
// lds_seg: 16 doubles private to this call (no trailing barrier).  One __syncthreads().
// ---------------------------------------------------------------------------------------
template <int NT, int NK>
__device__ __forceinline__ void block_scan_f64(const double (&q)[NK][2], double (&incl)[NK][2], double& total, double* lds_seg) {
    constexpr int WPR = NT / 64, NSEG = NK * WPR;
    static_assert(NSEG <= 16, "segment totals are scanned inside one 16-lane row");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s0[NK], s1[NK], exc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        s0[k] = q[k][0];
        s1[k] = s0[k] + q[k][1];
        const double inc = wave_incl_scan_f64(s1[k]);
        exc[k] = wave_shr1_f64(inc);
        if (lane == 63) lds_seg[k * WPR + wave] = inc;
    }
    __syncthreads();
    double sv = (lane & 15) < NSEG ? lds_seg[lane & 15] : 0.0;
    sv = sv + dpp_f64_zero<0x111, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x112, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x114, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x118, 0xF>(sv);
    total = readlane_f64(sv, 15);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int seg = k * WPR + wave;
        const double pre = seg ? readlane_f64(sv, seg - 1) : 0.0;
        const double base = pre + exc[k];
        incl[k][0] = base + s0[k];
        incl[k][1] = base + s1[k];
    }
}

// ---------------------------------------------------------------------------------------
// Model callbacks, compiled in (the reference's virtual fSamp/q1Samp/logGEv).
// ---------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ double model_prop(const ModelConst& c, double x, double zn, double zcov, const ExpTabEntry* etab) {
#if SSME_HAS_USER_MODEL
    if constexpr (MODEL == MODEL_USER0) return user_calls<ssme_user_model0>::prop(c, x, zn, zcov, etab);      // model_api.h
#endif
    if (MODEL == MODEL_SVOL_LEVERAGE) {   // test/test_pswarm.cpp:90-97
        const double e = dexp_scaled_t(-0.5 * x, 0, etab);
        const double mean = (c.a1 + c.a0 * (x - c.a1)) + (c.a4 * zcov) * e;
        return mean + zn * c.a3;
    }
    return c.a0 * x + zn * c.a1;          // univ_svol_bootstrap_filter.h:74-79
}

template <int MODEL>
__device__ __forceinline__ double model_logg(const ModelConst& c, double y, double x, const ExpTabEntry* etab) {
#if SSME_HAS_USER_MODEL
    if constexpr (MODEL == MODEL_USER0) return c.bad ? -dinf() : user_calls<ssme_user_model0>::logg(c, y, x, etab);
#endif
    if (MODEL == MODEL_LIN_GAUSS) {
        const double d = (y - x) * c.a4;
        const double v = (-c.a3 - SSME_HALF_LOG_2PI) - 0.5 * (d * d);
        return c.bad ? -dinf() : v;
    }
    // univ_svol_bootstrap_filter.h:83-86 / test_pswarm.cpp:101-108 in kernel form
    const double logb = (MODEL == MODEL_SVOL) ? c.a3 : 0.0;
    const double ib2 = (MODEL == MODEL_SVOL) ? c.a4 : 1.0;
    const double hl = logb + 0.5 * x;
    const double e = dexp_scaled_t(-x, 0, etab);
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
    const double rad = dsqrt(-2.0 * dlog_pn(u1));
    double sn, cs;
    dsincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

// the same in two halves, so that memory latency can be hidden between them
__device__ __forceinline__ void normal_pair_radius(uint32_t pair, uint32_t t, uint32_t rep, uint32_t k0, uint32_t k1,
                                                   double* rad, double* u2) {
    const u32x4 o = philox4x32_10(pair, t, rep, STREAM_PROP, k0, k1);
    *rad = dsqrt(-2.0 * dlog_pn(u01_oc(o.v0, o.v1)));
    *u2 = u01_co(o.v2, o.v3);
}
__device__ __forceinline__ void normal_pair_angle(double rad, double u2, double* z0, double* z1) {
    double sn, cs;
    dsincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

// ---- bootstrap filter: ONE Philox call per particle pair and time step -------------------------------------------------
// The thread that propagates pair p at time t also draws that pair's two exponential spacings of the multinomial resampler,
// so one counter (pair, t, filter, STREAM_PROP) feeds both: words 0-1 -> Box-Muller (radius uniform: 40 bits, strictly inside
// (0,1), i.e. |z| < 7.45; angle: the low 24 bits of word 1), words 2-3 -> E_0, E_1 = -log(u), u on the 2^-32 midpoint grid
// (the spacings are quantised to 2^-35 right away).  Logs by dlog_u (table in LDS, no division).  DESIGN.md section 4.1.
__device__ __forceinline__ u32x4 pair_words(uint32_t pair, uint32_t t, uint32_t rep, uint32_t k0, uint32_t k1) {
    return philox4x32_10(pair, t, rep, STREAM_PROP, k0, k1);
}
__device__ __forceinline__ void pair_spacings(const u32x4& o, const LogTabEntry* tab, double* e0, double* e1) {
    *e0 = -dlog_u32(u01_mid32(o.v2), tab);
    *e1 = -dlog_u32(u01_mid32(o.v3), tab);
}
// the draw tables of a hot kernel in LDS: dlog_u / dlog_u32 and dsincos_k24 (2 KiB)
struct DrawTabs {
    LogTabEntry log[SSME_LOG_TABLE_SIZE];
    SinCosEntry sc[SSME_SINCOS_TABLE_SIZE];
};
__device__ __forceinline__ void pair_normals(uint32_t w0, uint32_t w1, const DrawTabs* tab, double* z0, double* z1) {
    const double rad = dsqrt_pn(-2.0 * dlog_u(u01_mid40(w0, w1), tab->log));
    double sn, cs;
    dsincos_k24(w1, tab->sc, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}
// the tables of dlog_u, dsincos_k24 and dexp_scaled_t, in device memory; every workgroup of the hot kernels copies them
// into LDS (3 x 64 x 16 bytes) before its first use
static __device__ const LogTabEntry kLogTable[SSME_LOG_TABLE_SIZE] = {SSME_LOG_TABLE_ROWS};
static __device__ const SinCosEntry kSinCosTable[SSME_SINCOS_TABLE_SIZE] = {SSME_SINCOS_TABLE_ROWS};
static __device__ const ExpTabEntry kExpTable[SSME_EXP_TABLE_SIZE] = {SSME_EXP_TABLE_ROWS};
template <int NT>
__device__ __forceinline__ void load_log_table(DrawTabs* lds_tab) {
    static_assert(SSME_LOG_TABLE_SIZE == SSME_SINCOS_TABLE_SIZE, "one pass fills both");
    for (int i = threadIdx.x; i < SSME_LOG_TABLE_SIZE; i += NT) { lds_tab->log[i] = kLogTable[i]; lds_tab->sc[i] = kSinCosTable[i]; }
}
template <int NT>
__device__ __forceinline__ void load_exp_table(ExpTabEntry* lds_tab) {
    for (int i = threadIdx.x; i < SSME_EXP_TABLE_SIZE; i += NT) lds_tab[i] = kExpTable[i];
}
// Gamma(shape) draw, Marsaglia & Tsang (2000), attempts driven by the Philox counter
__device__ __forceinline__ double gamma_draw(uint32_t b, uint32_t t, uint32_t rep, uint32_t k0, uint32_t k1, double shape,
                                             uint32_t stream_base = STREAM_GAMMA) {
    const double d = shape - 0.3333333333333333;
    const double c = 1.0 / dsqrt(9.0 * d);
    for (int a = 0; a < 32; ++a) {
        const u32x4 o1 = philox4x32_10(b, t, rep, stream_base + 2 * a, k0, k1);
        const u32x4 o2 = philox4x32_10(b, t, rep, stream_base + 2 * a + 1, k0, k1);
        const double rad = dsqrt(-2.0 * dlog_pn(u01_oc(o1.v0, o1.v1)));
        double sn, cs;
        dsincos2pi(u01_co(o1.v2, o1.v3), &sn, &cs);
        const double xn = rad * cs;
        const double v = 1.0 + c * xn;
        if (v > 0.0) {
            const double v3 = (v * v) * v;
            const double lhs = dlog_pn(u01_oc(o2.v0, o2.v1));
            const double rhs = ((0.5 * (xn * xn) + d) - d * v3) + d * dlog(v3);
            if (lhs < rhs) return d * v3;
        }
    }
    return d;
}

// #{ j < n : get(j) < target } for monotone get, n = 2^k (a NaN target counts nothing)
template <class F>
__device__ __forceinline__ int count_less_pow2(int n, double target, F get) {
    int pos = 0;
    for (int step = n >> 1; step >= 1; step >>= 1)
        if (get(pos + step - 1) < target) pos += step;
    // pos in [0, n-1]; the last element is not probed: callers clamp the count to n-1 anyway
    return pos;
}

// Level-2 of one filter: global max m over the tile maxima (NaN propagating), rescaled integer
// tile sums A'_b = rint(A_b exp(m_b - m) 2^(rg-41)), their exact inclusive scan.  Thread tid holds
// entries j = e*NT + tid, e < NE = 2048/NT: at B <= NT every wave owns live entries, so the serial
// exp / divide per entry is spread over the whole block.  Rows without live entries are skipped.
// 1 + nrows barriers.  lds_d: NT/64 doubles, lds_seg: 16 doubles per row (4 rows).
template <int NT>
__device__ __forceinline__ void level2_load(const double* ts, const double* tm, int B, double (&A)[2048 / NT],
                                            double (&mb)[2048 / NT]) {
    constexpr int NE = 2048 / NT;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = e * NT + threadIdx.x;
        if (j < B) { A[e] = ts[j]; mb[e] = tm[j]; } else { A[e] = 0.0; mb[e] = 0.0; }
    }
}

template <int NT>
__device__ __forceinline__ void level2_scan(const double (&A)[2048 / NT], const double (&mb)[2048 / NT], int B, int rshift,
                                            double& m, double (&Ap)[2048 / NT], double (&Tinc)[2048 / NT], double& S,
                                            double* lds_d, double* lds_seg, const ExpTabEntry* etab) {
    constexpr int NE = 2048 / NT, NW = NT / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double mx = -dinf();
    bool nan = false;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (e * NT < B) {
            const int j = e * NT + threadIdx.x;
            if (j < B) { const double v = mb[e]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
        }
    }
    m = block_max_nanprop<NT>(mx, nan, lds_d);
    double inc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        Ap[e] = 0.0; inc[e] = 0.0;
        if (e * NT < B) {
            const int j = e * NT + threadIdx.x;
            // NaN (m or m_b NaN) is squashed to 0 by dexp_scaled's clamp: A' = rint(A * 0) = 0
            if (j < B) Ap[e] = __builtin_rint(A[e] * dexp_scaled_t(mb[e] - m, rshift - kTileShift, etab));
            inc[e] = wave_incl_scan_f64(Ap[e]);
            if (lane == 63) lds_seg[e * 16 + wave] = inc[e];
        }
    }
    __syncthreads();
    double carry = 0.0;
    S = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        Tinc[e] = 0.0;
        if (e * NT < B) {
            double sv = (lane & 15) < NW ? lds_seg[e * 16 + (lane & 15)] : 0.0;
            sv = sv + dpp_f64_zero<0x111, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x112, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x114, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x118, 0xF>(sv);
            const double pre = wave ? readlane_f64(sv, wave - 1) : 0.0;
            Tinc[e] = (carry + pre) + inc[e];
            carry = carry + readlane_f64(sv, 15);
        }
    }
    S = carry;
}

// Workgroup -> tile map.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup id mod 8), and each XCD has
// its own L2.  Consecutive tiles share source cdf tiles (a tile's sorted targets fall into 1-3 neighbouring tiles), so
// XCD k is given a CONTIGUOUS range of tiles: the re-reads of a neighbour's tile then hit that XCD's L2 instead of
// being fetched once per XCD.  Any bijection is correct (results do not depend on the map).
__device__ __forceinline__ int xcd_tile_of_block(int bid, int nblocks) {
    constexpr int kXcd = 8;
    const int q = nblocks / kXcd, r = nblocks % kXcd;
    const int k = bid % kXcd, idx = bid / kXcd;
    return k * q + (k < r ? k : r) + idx;
}

// Bounds [t_lo, t_hi] of the integer resampling targets of output tile b (first particle i_first, nvalid valid
// outputs), known without the random spacings.  Shared by k_filter_step and the shard planner (k_shard_plan),
// which must agree to the bit on which source tiles a tile touches.
__device__ __forceinline__ void tile_target_bounds(int resampler, double S, int N, int i_first, int nvalid, double pgam,
                                                   double pgam_next, double G, double u0, double& t_scale, double& t_lo,
                                                   double& t_hi) {
    if (resampler == RESAMP_MULTINOMIAL) {
        t_scale = S / G;                          // targets: (pgam + Gamma_b E_cum/E_tile) * S'/G  
        t_lo = __builtin_ceil(pgam * t_scale);
        t_hi = __builtin_ceil(pgam_next * t_scale) + (S * 0x1.0p-40 + 2.0);   // slack covers the rounding of ratio*E_tile vs Gamma_b
    } else if (resampler == RESAMP_SYSTEMATIC) {
        t_scale = S / (double)N;
        t_lo = __builtin_ceil(((double)i_first + u0) * t_scale);
        t_hi = __builtin_ceil(((double)(i_first + nvalid - 1) + u0) * t_scale);
    } else {
        t_scale = S / (double)N;
        t_lo = __builtin_ceil((double)i_first * t_scale);
        t_hi = __builtin_ceil((double)(i_first + nvalid) * t_scale);
    }
}


// Wave priority by phase.  Two workgroups share a CU and the hardware issues the OLDER waves first, so without this the
// first-dispatched workgroup runs at full speed, the second one lags by ~5 us, and for that tail the CU holds half its
// waves.  Lowering the priority as a workgroup advances lets the one that is behind catch up:
// both then finish together and the CU stays full.  mode 1 (the whole grid is resident at once: 20.3 -> 19.2 us at
// N = 2^20): 3 | after the tile loads are issued 2 | after the search 1 | after logG 0.  mode 2 (several residency waves of
// workgroups, e.g. 4096 filters' tiles; +2 %): 3 | 2 | after the LDS fill 1 | after the search 0.  Scheduling only --
// results do not depend on it.
__device__ __forceinline__ void prio_at(int mode, int idx) {
    if (mode == 1) {
        if (idx == 0) __builtin_amdgcn_s_setprio(3); else if (idx == 3) __builtin_amdgcn_s_setprio(2);
        else if (idx == 6) __builtin_amdgcn_s_setprio(1); else if (idx == 8) __builtin_amdgcn_s_setprio(0);
    } else if (mode == 2) {
        if (idx == 0) __builtin_amdgcn_s_setprio(3); else if (idx == 3) __builtin_amdgcn_s_setprio(2);
        else if (idx == 5) __builtin_amdgcn_s_setprio(1); else if (idx == 6) __builtin_amdgcn_s_setprio(0);
    } else if (mode == 3) {          // experimental schedules (SSME_PRIO_MODE): later drops
        if (idx == 0) __builtin_amdgcn_s_setprio(3); else if (idx == 4) __builtin_amdgcn_s_setprio(2);
        else if (idx == 7) __builtin_amdgcn_s_setprio(1); else if (idx == 9) __builtin_amdgcn_s_setprio(0);
    } else if (mode == 4) {          // earlier drops
        if (idx == 0) __builtin_amdgcn_s_setprio(3); else if (idx == 2) __builtin_amdgcn_s_setprio(2);
        else if (idx == 4) __builtin_amdgcn_s_setprio(1); else if (idx == 6) __builtin_amdgcn_s_setprio(0);
    } else if (mode == 5) {          // two levels only
        if (idx == 0) __builtin_amdgcn_s_setprio(1); else if (idx == 6) __builtin_amdgcn_s_setprio(0);
    }
}
#define PRIO_AT(i) prio_at(a.prio_mode, i)

// ---------------------------------------------------------------------------------------
// k_filter_step: one bootstrap-filter step for every tile of every filter.
// grid = (B tiles, R filters), block = NT threads, NK = TILE/(2 NT) particle pairs per thread (TILE = 2048: NT = 256/512/1024;
// TILE = 512: NT = 256), dynamic LDS = (2*max(Bpow2,2) + 3*TILE) * 8 bytes.
// RS >= 0 compiles the reference's configuration in (resampler RS, resampling every step, t > 0, no debug outputs): the
// uniform branches of the general kernel disappear and the compiler schedules across what they separated; RS = -1 is
// the general kernel (any resampler / schedule / t = 0 / ancestor and log-weight recording).  Same arithmetic, same bits.
// Phase order is chosen so that arithmetic hides memory latency: the cdf tiles are requested,
// then the Box-Muller radii are computed while they arrive; the ancestor states are requested,
// then the Box-Muller angles are computed while they arrive.
// ---------------------------------------------------------------------------------------
// The count-searches of a thread in the staged tiles, all NQ of them descending TOGETHER, one level per round: a round is NQ
// independent ds_read_b64 (position = absolute LDS byte address, the level's offset in the immediate field), ONE wait, then per
// search compare + select + add -- 3 VALU instructions per probe and log2(TILE) dependent LDS round trips per thread.
// Written as inline assembly because the compiler, given the same loop in C++, either spends a fourth instruction per probe
// on the address or serialises the NQ chains (44 dependent round trips; both seen in the ISA, round 3).  The wait names the
// loaded values as in/out operands, so nothing that consumes them can be scheduled above it.
template <int STEP, int NQ>
struct lds_count_search {
    static __device__ __forceinline__ void run(uint32_t (&pa)[NQ], const double (&t)[NQ]) {
        static_assert(NQ == 2 || NQ == 4 || NQ == 8, "one, two or four particle pairs per thread");
        double v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(v[q]) : "v"(pa[q]), "n"((STEP - 1) * 8) : "memory");
        if constexpr (NQ == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        else if constexpr (NQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]));
#pragma unroll
        for (int q = 0; q < NQ; ++q) pa[q] = (v[q] < t[q]) ? pa[q] + (uint32_t)(STEP * 8) : pa[q];
        lds_count_search<STEP / 2, NQ>::run(pa, t);
    }
};
template <int NQ>
struct lds_count_search<0, NQ> {
    static __device__ __forceinline__ void run(uint32_t (&)[NQ], const double (&)[NQ]) {}
};

// The staged search of one thread: targets tau (integers in the filter's global fixed point) -> positions among the `span`
// (<= 3) cdf tiles staged in LDS at lds_stage, returned as ELEMENT offsets from the first staged tile (0 .. span TILE - 1).
// Pm / T0 / T1: inclusive prefixes T' of the tile before the first staged tile and of the first two staged tiles;
// R0..R2: A / A' of the staged tiles.  Shared by k_filter_step and the Liu-West selection (lw_select).
// A position is an absolute LDS byte address with the tile select folded in, so a probe is ds_read_b64 (constant offset
// field) + compare + add + select.  The target's tile is a uniform case split on `span`: two staged tiles (the common case)
// need one compare and three selects per particle instead of two and nine.
template <int TILE, int NK>
__device__ __forceinline__ void staged_search(const double (&tau)[NK][2], int span, double Pm, double T0, double T1, double R0, double R1,
                                              double R2, const double* lds_stage, int (&off)[NK][2]) {
    double tloc[NK][2];
    uint32_t pb[NK][2];
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const uint32_t stage_a = (uint32_t)(__UINTPTR_TYPE__)(lds_cdouble*)lds_stage;
    if (span == 1) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { tloc[k][c] = __builtin_ceil((tau[k][c] - Pm) * R0); pb[k][c] = stage_a; }
        }
    } else if (span == 2) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool up = T0 < tau[k][c];
                tloc[k][c] = __builtin_ceil((tau[k][c] - (up ? T0 : Pm)) * (up ? R1 : R0));
                pb[k][c] = up ? stage_a + (uint32_t)(TILE * 8) : stage_a;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double target = tau[k][c];
                int sel = (T0 < target ? 1 : 0) + (T1 < target ? 1 : 0);
                sel = sel < span - 1 ? sel : span - 1;
                const double Pb = sel == 0 ? Pm : (sel == 1 ? T0 : T1);
                const double Rb = sel == 0 ? R0 : (sel == 1 ? R1 : R2);
                tloc[k][c] = __builtin_ceil((target - Pb) * Rb);
                pb[k][c] = stage_a + (uint32_t)sel * (uint32_t)(TILE * 8);
            }
        }
    }
    uint32_t pa[2 * NK];
    double tq[2 * NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { pa[2 * k] = pb[k][0]; pa[2 * k + 1] = pb[k][1]; tq[2 * k] = tloc[k][0]; tq[2 * k + 1] = tloc[k][1]; }
    lds_count_search<TILE / 2, 2 * NK>::run(pa, tq);
#pragma unroll
    for (int k = 0; k < NK; ++k) { off[k][0] = (int)((pa[2 * k] - stage_a) >> 3); off[k][1] = (int)((pa[2 * k + 1] - stage_a) >> 3); }
}

// 16-byte store of a particle pair.  stream = 1: non-temporal, the lines leave the XCD's L2 as they are written.  A launch
// whose workgroups are all resident at once ends with every L2 full of dirty lines (2 MB per XCD at N = 2^20), and the
// write-back at the end of the kernel is then serial time: 15.9 -> 14.3 us per step at N = 2^20.  Grids of several
// residency waves overlap that write-back with the next workgroups' arithmetic and lose 3 % with streaming stores.
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_pair(double* p, double v0, double v1, int stream) {
    if (stream) {
        f64x2_t v = {v0, v1};
        __builtin_nontemporal_store(v, reinterpret_cast<f64x2_t*>(p));
    } else *reinterpret_cast<double2*>(p) = make_double2(v0, v1);
}
template <int MODEL, int NT, bool BIG = false, int TILE = kTile, int RS = -1, bool WL2 = false>
__global__ __launch_bounds__(NT) void k_filter_step(const StepArgs a) {
    constexpr int NK = TILE / 2 / NT;
    constexpr bool HOT = RS >= 0;
    static_assert(NK >= 1 && NK * NT * 2 == TILE, "tile = 2 NT NK particles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nT2 = (BIG || a.Bpow2 < 2) ? 2 : a.Bpow2;      // BIG: level-2 comes from k_level2_plan, no tables in LDS
    // the staged tiles come FIRST: their LDS address is then a compile-time constant and every search probe is a
    // ds_read_b64 with the level's offset in its immediate field (the level-2 tables behind them are indexed at run time anyway)
    double* lds_stage = reinterpret_cast<double*>(smem);         // [3][TILE] staged cdf tiles, 16-byte aligned
    double* lds_T = lds_stage + kStageTiles * TILE;              // [Bpow2] inclusive prefixes of A'
    double* lds_R = lds_T + nT2;                                 // [Bpow2] A_b / A'_b
    __shared__ double lds_seg_a[16];
    __shared__ double lds_seg_l2[64];
    __shared__ double lds_seg_c[16];
    __shared__ int lds_cnt[2];
    __shared__ double lds_R3[4];         // A_b / A'_b of the staged tiles
    __shared__ double lds_d1[16];
    __shared__ double lds_d2[16];
    __shared__ __attribute__((aligned(16))) DrawTabs lds_dtab;
    __shared__ __attribute__((aligned(16))) ExpTabEntry lds_etab[SSME_EXP_TABLE_SIZE];

    const int tid = threadIdx.x;
    load_log_table<NT>(&lds_dtab);        // visible after the first barrier below (every path has one before its first draw)
    load_exp_table<NT>(lds_etab);        // first used behind level2_scan's first barrier (block max)
    // (filter, tile) of this workgroup: the launch's tiles in filter-major order, a contiguous range per XCD
    const int gtile = xcd_tile_of_block((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int r = gtile / (int)gridDim.x, bloc = gtile - r * (int)gridDim.x;
    const int b = bloc + a.tile0;                       // global tile id (tile0 = 0 unless the filter is sharded over GPUs)
    const int out0 = a.tile0 * TILE, win0 = a.win_tile0 * TILE;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
    const size_t rowoff = (size_t)r * a.Npad;
    const ModelConst mc = a.mc[r];
    constexpr int DX = model_dx<MODEL>(), DY = model_dy<MODEL>();           // 1 unless a user model says otherwise (model_api.h)
    constexpr bool VEC = DX > 1 || DY > 1;
    const double y = a.by_value ? a.y_now : a.y[(size_t)a.yi * DY];
    const double zcov = a.by_value ? a.z_now : (a.z ? a.z[a.yi] : 0.0);
    const int rsm = HOT ? RS : a.resampler;
    const bool first_step = !HOT && a.t == 0;
    const bool resampled = HOT ? true : ((a.t > 0) && (a.t % a.resamp_sched == 0));
    const bool need_l2 = HOT ? true : ((a.t > 0) && (resampled || (!BIG && bloc == 0 && a.finalize_prev)));
    const bool sorted = rsm != RESAMP_MULTINOMIAL_IID;
    uint32_t* const anc_p = HOT ? nullptr : a.anc;
    double* const logw_p = HOT ? nullptr : a.logw;
    const int i_first = b * TILE;
    const int nvalid = (a.N - i_first) < TILE ? (a.N - i_first) : TILE;    // valid outputs in this tile (>= 1)
    const bool ragged = nvalid < TILE;                                     // uniform: this tile holds particles beyond N

    STAMP(a, 0);
    PRIO_AT(0);
    // --- issue the level-2 loads first: previous step's tile sums and maxima ---
    constexpr int NE = 2048 / NT;
    double A2[NE], M2[NE], ApL2[NE];
    // Filters of at most 128 tiles (512 filters x 2^14: eight tiles each; one filter of 2^16: 128): every WAVE holds all tile
    // sums, two per lane (entries lane and lane + 64), and takes the level-2 by itself -- wave max, wave scans, no LDS
    // hand-over and one barrier less.  Afterwards a thread keeps the entry the general layout gives it (j = tid: the low
    // half in wave 0, the high half in wave 1), so everything after the level-2 reads the same registers; the other waves'
    // copies fail every `j < B` test below.  Integer sums and a max: the same bits as the block-wide form.
    // (a template parameter, WL2 = the host's "B <= 128": as a run-time branch its mere presence cost the N = 2^20 kernel 2 %)
    const bool wave_l2 = WL2 && !BIG && need_l2;
    double Ahi = 0.0, Mhi = 0.0;
    if (wave_l2) {
        const int ln = tid & 63;
#pragma unroll
        for (int e = 0; e < NE; ++e) { A2[e] = 0.0; M2[e] = 0.0; }
        if (ln < a.B) { A2[0] = a.tsum_in[(size_t)r * a.Bs + ln]; M2[0] = a.tmax_in[(size_t)r * a.Bs + ln]; }
        if (a.B > 64 && ln + 64 < a.B) { Ahi = a.tsum_in[(size_t)r * a.Bs + ln + 64]; Mhi = a.tmax_in[(size_t)r * a.Bs + ln + 64]; }
    } else if (need_l2 && !BIG) level2_load<NT>(a.tsum_in + (size_t)r * a.Bs, a.tmax_in + (size_t)r * a.Bs, a.B, A2, M2);
    const double* l2T = a.l2_T + (size_t)r * a.Bs;
    const double* l2R = a.l2_R + (size_t)r * a.Bs;
    // T'_j of the split level-2 = l2_offs[j / 1024] + l2_tsrc[j]: a table plus zeros (k_level2_plan / k_l2_ranges), or block
    // offset + block-local scan (l2_inkernel) -- one expression, no branch at the reads
    const bool l2ink = BIG && a.l2_inkernel;
    const double* l2ts = a.l2_tsrc + (size_t)r * a.Bs;
    const double* l2of = a.l2_offs + (size_t)r * a.l2_offs_stride;
    auto l2Tg = [&](int j) -> double { return l2of[j >> 10] + l2ts[j]; };
    // l2_inkernel: wave 0 looks for this tile's source range itself; the 64 entries around its own tile id, requested now
    // (two loads; their sum is taken where it is needed, after the other loads of the head have been issued)
    const int l2w0 = (b - 32 < 0 || a.B <= 64) ? 0 : (b - 32 > a.B - 64 ? a.B - 64 : b - 32);
    double l2win_o = 0.0, l2win_t = dinf();
    if (l2ink && need_l2 && sorted && tid < 64 && l2w0 + tid < a.B) { l2win_o = l2of[(l2w0 + tid) >> 10]; l2win_t = l2ts[l2w0 + tid]; }
    if (tid == 0) { lds_cnt[0] = 0; lds_cnt[1] = 0; }
    double gam = 0.0, pgam = 0.0, pgam_next = 0.0, G = 1.0;
    const bool multinomial = resampled && rsm == RESAMP_MULTINOMIAL;
    if (multinomial) {
        const size_t gidx = ((size_t)a.gi * a.R + r) * a.B + b;
        gam = a.gam[gidx]; pgam = a.pgam[gidx]; G = a.gtot[(size_t)a.gi * a.R + r];
        pgam_next = (b + 1 < a.B) ? a.pgam[gidx + 1] : G;
    }

    STAMP(a, 1);
    PRIO_AT(1);
    // --- level-2: global max, rescaled tile sums A', inclusive prefixes T', total S'; tile range of my targets ---
    double S = 0.0;
    double t_scale = 0.0, u0 = 0.0;
    if (need_l2 && BIG) {
        // split level-2: S', T', A/A' and this tile's source range were computed once per filter by k_level2_plan
        S = a.scal[r].S;
        double t_lo, t_hi;
        if (rsm == RESAMP_SYSTEMATIC) {
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, key0, key1);
            u0 = u01_co(ox.v0, ox.v1);
        }
        tile_target_bounds(rsm, S, a.N, i_first, nvalid, pgam, pgam_next, G, u0, t_scale, t_lo, t_hi);
        if (l2ink && sorted && tid < 64) {
            // lo = #{T'_j < t_lo}, hi = #{T'_j < t_hi}.  T' is nondecreasing, so the window [w0, w0 + 64) gives a count
            // exactly when it brackets the target: something below it inside (or w0 = 0) and something not below it inside (or
            // the window reaches B).  Weights that are not wildly uneven put the targets of tile b a few tiles from b; beyond
            // that a 64-ary descent (strides 256, 4, 1: three rounds for 16384 tiles) finds the same counts.
            const bool at_end = l2w0 + 64 >= a.B;
            const double l2win = l2win_o + l2win_t;
            int lo = __popcll(__ballot(l2win < t_lo)), hi = __popcll(__ballot(l2win < t_hi));
            const bool ok = (lo > 0 || l2w0 == 0) && (lo < 64 || at_end) && (hi > 0 || l2w0 == 0) && (hi < 64 || at_end);
            lo += l2w0; hi += l2w0;
            if (!ok) {
                lo = 0; hi = 0;
                auto round = [&](int stride) {
                    const int jl = lo + (tid + 1) * stride - 1, jh = hi + (tid + 1) * stride - 1;
                    const double vl = jl < a.B ? l2Tg(jl) : dinf(), vh = jh < a.B ? l2Tg(jh) : dinf();
                    lo += __popcll(__ballot(vl < t_lo)) * stride;
                    hi += __popcll(__ballot(vh < t_hi)) * stride;
                };
                round(256); round(4); round(1);
            }
            if (tid == 0) { lds_cnt[0] = lo; lds_cnt[1] = hi; }
        }
    }
    if (need_l2 && !BIG) {
        double Tinc[NE];
        double m;
#ifdef SSME_ABLATE
        if (a.stamps) { asm volatile("" :: "v"(A2[0]), "v"(M2[0])); }   // force the loads to have landed
        STAMP(a, 13);
#endif
        if (wave_l2) {
            __syncthreads();                               // the LDS tables and lds_cnt = 0 (level2_scan's first barrier otherwise)
            const int ln = tid & 63;
            const bool two = a.B > 64;                     // uniform: a second entry per lane
            const bool v0 = ln < a.B, v1 = two && (ln + 64 < a.B);
            const double mv = M2[0];
            bool nanv = v0 && (mv != mv);
            double mx = (v0 && !nanv) ? mv : -dinf();
            if (two) {
                const bool n1 = v1 && (Mhi != Mhi);
                if (v1 && !n1) mx = (Mhi > mx) ? Mhi : mx;
                nanv = nanv || n1;
            }
            const double mw = wave_max_f64(mx);
            m = __ballot(nanv) ? dnan() : mw;
#pragma unroll
            for (int e = 0; e < NE; ++e) { ApL2[e] = 0.0; Tinc[e] = 0.0; }
            if (v0) ApL2[0] = __builtin_rint(A2[0] * dexp_scaled_t(mv - m, a.rshift - kTileShift, lds_etab));
            Tinc[0] = wave_incl_scan_f64(ApL2[0]);
            S = readlane_f64(Tinc[0], 63);
            if (two) {
                double Ap1 = 0.0;
                if (v1) Ap1 = __builtin_rint(Ahi * dexp_scaled_t(Mhi - m, a.rshift - kTileShift, lds_etab));
                const double T1 = S + wave_incl_scan_f64(Ap1);
                S = readlane_f64(T1, 63);
                if ((tid >> 6) == 1) { A2[0] = Ahi; ApL2[0] = Ap1; Tinc[0] = T1; }       // the general layout: thread j = tid holds entry j
            }
        } else
        level2_scan<NT>(A2, M2, a.B, a.rshift, m, ApL2, Tinc, S, lds_d1, lds_seg_l2, lds_etab);
        STAMP(a, 14);
        // bounds [t_lo, t_hi] of this tile's targets, known to every thread without the spacings
        double t_lo = 0.0, t_hi = dinf();
        if (rsm == RESAMP_SYSTEMATIC) {
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, key0, key1);
            u0 = u01_co(ox.v0, ox.v1);
        }
        tile_target_bounds(rsm, S, a.N, i_first, nvalid, pgam, pgam_next, G, u0, t_scale, t_lo, t_hi);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e * NT < a.Bpow2) {
                const int j = e * NT + tid;
                if (j < a.Bpow2) lds_T[j] = (j < a.B) ? Tinc[e] : dinf();
                if (resampled && sorted) {
                    // #{T'_j < t_lo}, #{T'_j < t_hi}: wave popcounts, one LDS atomic per wave
                    const int w_lo = __popcll(__ballot(j < a.B && Tinc[e] < t_lo));
                    const int w_hi = __popcll(__ballot(j < a.B && Tinc[e] < t_hi));
                    if ((tid & 63) == 0) { if (w_lo) atomicAdd(&lds_cnt[0], w_lo); if (w_hi) atomicAdd(&lds_cnt[1], w_hi); }
                }
            }
        }
        STAMP(a, 15);
        if (bloc == 0 && tid == 0 && a.finalize_prev) {
            FilterScalars* sc = a.scal + r;
            const double Sdd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
            const double lse = m + dlog(Sdd);
            const double ll = lse - sc->prev;
            sc->m = m;
            sc->S = S;
            sc->last_ll = ll;
            sc->loglik = sc->loglik + ll;
            sc->prev = resampled ? a.logN : lse;
            if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
        }
    }
    STAMP(a, 2);
    PRIO_AT(2);

    // --- request the cdf tiles my targets fall into (coalesced 16-byte loads into registers) ---
    int bb_min = 0, span = kStageTiles + 1;
    // three explicitly named register tiles (a runtime-indexed array would be placed in scratch memory)
    double2 stg0[NK], stg1[NK], stg2[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { stg0[k] = make_double2(0.0, 0.0); stg1[k] = stg0[k]; stg2[k] = stg0[k]; }
    const double* cdf_r = a.cdf_in + rowoff;
    const double* xin_r = a.x_in + rowoff;
    if (resampled) {
        __syncthreads();          // lds_T, lds_cnt visible
        if (sorted) {
            int lo, hi;
            if (BIG && !l2ink) { lo = a.l2_lo[(size_t)r * a.Bs + b]; hi = a.l2_hi[(size_t)r * a.Bs + b]; }
            else { lo = lds_cnt[0]; hi = lds_cnt[1]; }
            lo = lo < a.B - 1 ? lo : a.B - 1;
            hi = hi < a.B - 1 ? hi : a.B - 1;
            bb_min = __builtin_amdgcn_readfirstlane(lo);
            span = __builtin_amdgcn_readfirstlane(hi) - bb_min + 1;
        }
        if (a.win_flag) {
            // fixed-halo sharding: the window this rank holds was exchanged BEFORE anyone knew the plan.  If my sources
            // leave it, say so (the host reruns the series on the exact path) and stay inside the buffer for this launch.
            const int first = a.win_tile0 > 0 ? a.win_tile0 : 0, last = a.win_tile0 + a.win_tiles - 1 < a.B - 1 ? a.win_tile0 + a.win_tiles - 1 : a.B - 1;
            if (bb_min < first || bb_min + span - 1 > last) {
                if (tid == 0) atomicOr(a.win_flag, 1);
                bb_min = bb_min < first ? first : (bb_min > last ? last : bb_min);
                span = 1;
            }
        }
        if (!BIG) {
            // A_b / A'_b (maps a global target into its tile's fixed point).  Only the staged tiles' ratios are needed, so the
            // IEEE division runs in the one or two waves that hold those entries, off the path to the first barrier; an
            // output tile spanning more than three source tiles (iid resampling, degenerate weights) publishes all of them.
            if (span <= kStageTiles) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (e * NT < a.Bpow2) {
                        const int j = e * NT + tid;
#pragma unroll
                        for (int sl = 0; sl < kStageTiles; ++sl) {
                            const int bs = bb_min + sl < a.B ? bb_min + sl : a.B - 1;
                            if (j == bs) lds_R3[sl] = A2[e] / ApL2[e];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (e * NT < a.Bpow2) {
                        const int j = e * NT + tid;
                        if (j < a.Bpow2) lds_R[j] = (j < a.B) ? A2[e] / ApL2[e] : 0.0;
                    }
                }
                __syncthreads();
            }
        }
        if (span <= kStageTiles) {
            // a uniform base (scalar registers) plus a 32-bit lane offset: no 64-bit address arithmetic per lane
            const unsigned char* src = reinterpret_cast<const unsigned char*>(cdf_r + (size_t)(bb_min - a.win_tile0) * TILE);
            const uint32_t lane_off = (uint32_t)tid * 16u;
#pragma unroll
            for (int k = 0; k < NK; ++k) stg0[k] = *reinterpret_cast<const double2*>(src + (lane_off + (uint32_t)(k * NT * 16)));
            if (span >= 2) {
#pragma unroll
                for (int k = 0; k < NK; ++k) stg1[k] = *reinterpret_cast<const double2*>(src + (lane_off + (uint32_t)(TILE * 8 + k * NT * 16)));
            }
            if (span >= 3) {
#pragma unroll
                for (int k = 0; k < NK; ++k) stg2[k] = *reinterpret_cast<const double2*>(src + (lane_off + (uint32_t)(2 * TILE * 8 + k * NT * 16)));
            }
        }
    }
    STAMP(a, 3);
    PRIO_AT(3);

    // --- exponential spacings of the multinomial resampler (liu_west_filter.h:105-139), exact tile scan;
    //     this arithmetic hides the latency of the tile loads ---
    double le[NK][2], se = 1.0;
    uint32_t nw0[NK], nw1[NK];            // words 0-1 of each pair's Philox output: the propagation normals (below)
    if (!resampled) __syncthreads();      // the log table is in LDS (the resampling path has passed a barrier already)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const u32x4 o = pair_words((uint32_t)(b * (TILE / 2) + k * NT + tid), (uint32_t)a.t, rep, key0, key1);
        nw0[k] = o.v0; nw1[k] = o.v1;
        if (multinomial) {
            const int i0 = i_first + (k * NT + tid) * 2;
            double e0, e1;
            if (ABL(a, 1)) { e0 = 1.0 + 1e-6 * (double)(i0 & 1023); e1 = 1.0; }
            else pair_spacings(o, lds_dtab.log, &e0, &e1);
            le[k][0] = __builtin_rint(e0 * 34359738368.0 /* 2^35 */);
            le[k][1] = __builtin_rint(e1 * 34359738368.0);
            if (ragged) {                 // only the last tile of a filter whose N is not a multiple of the tile: a uniform branch
                asm volatile("");         // (kept a branch: as selects these masks cost every tile 12 instructions per particle pair)
                if (!(i0 < a.N)) le[k][0] = 0.0;
                if (!(i0 + 1 < a.N)) le[k][1] = 0.0;
            }
        }
    }
    if (multinomial) {
        double qe[NK][2];
#pragma unroll
        for (int k = 0; k < NK; ++k) { qe[k][0] = le[k][0]; qe[k][1] = le[k][1]; }
        block_scan_f64<NT, NK>(qe, le, se, lds_seg_a);
    }
    STAMP(a, 4);
    PRIO_AT(4);

    double xin[NK][2], lw_old[NK][2];
    int srcv[NK][2];                      // vector states: where the other components of the source particle are
#pragma unroll
    for (int k = 0; k < NK; ++k) { srcv[k][0] = 0; srcv[k][1] = 0; }
    if (first_step) {
#pragma unroll
        for (int k = 0; k < NK; ++k) { xin[k][0] = 0.0; xin[k][1] = 0.0; lw_old[k][0] = 0.0; lw_old[k][1] = 0.0; }
    } else if (!resampled) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const size_t idx = rowoff + (size_t)(i_first - out0) + (k * NT + tid) * 2;
            const double2 xv = *reinterpret_cast<const double2*>(a.x_in + idx);
            const double2 lv = *reinterpret_cast<const double2*>(logw_p + idx);
            xin[k][0] = xv.x; xin[k][1] = xv.y; lw_old[k][0] = lv.x; lw_old[k][1] = lv.y;
            if constexpr (DX > 1) { srcv[k][0] = i_first + (k * NT + tid) * 2; srcv[k][1] = srcv[k][0] + 1; }
        }
    } else {
        // --- integer resampling targets in [0, S'] ---
        double tau[NK][2];
        const double ratio = gam / se;                  // per tile: Gamma_b * E_j / sum_tile(E)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i0 = i_first + (k * NT + tid) * 2;
            if (rsm == RESAMP_MULTINOMIAL) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double t1 = ratio * le[k][c];
                    const double t2 = pgam + t1;
                    tau[k][c] = __builtin_ceil(t2 * t_scale);
                }
            } else if (rsm == RESAMP_SYSTEMATIC) {
                tau[k][0] = __builtin_ceil(((double)i0 + u0) * t_scale);
                tau[k][1] = __builtin_ceil(((double)(i0 + 1) + u0) * t_scale);
            } else {
                const u32x4 o = philox4x32_10((uint32_t)(i0 >> 1), (uint32_t)a.t, rep, STREAM_RESAMP, key0, key1);
                const double v0 = u01_co(o.v0, o.v1), v1 = u01_co(o.v2, o.v3);
                if (rsm == RESAMP_STRATIFIED) {
                    tau[k][0] = __builtin_ceil(((double)i0 + v0) * t_scale);
                    tau[k][1] = __builtin_ceil(((double)(i0 + 1) + v1) * t_scale);
                } else {
                    tau[k][0] = __builtin_ceil(v0 * S);
                    tau[k][1] = __builtin_ceil(v1 * S);
                }
            }
        }

        if (span <= kStageTiles) {
            // --- staged tiles -> LDS, then count-search in LDS ---
            {
                double* dst = lds_stage + tid * 2;
#pragma unroll
                for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + k * NT * 2) = stg0[k];
                if (span >= 2) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + TILE + k * NT * 2) = stg1[k];
                }
                if (span >= 3) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) *reinterpret_cast<double2*>(dst + 2 * TILE + k * NT * 2) = stg2[k];
                }
            }
            const int b1 = bb_min + 1 < a.B ? bb_min + 1 : a.B - 1, b2 = bb_min + 2 < a.B ? bb_min + 2 : a.B - 1;
            const double T0 = BIG ? l2Tg(bb_min) : lds_T[bb_min];
            const double T1 = (bb_min + 1 < a.B) ? (BIG ? l2Tg(bb_min + 1) : lds_T[bb_min + 1]) : dinf();
            const double Pm = bb_min ? (BIG ? l2Tg(bb_min - 1) : lds_T[bb_min - 1]) : 0.0;
            __syncthreads();
            const double R0 = BIG ? l2R[bb_min] : lds_R3[0], R1 = BIG ? l2R[b1] : lds_R3[1], R2 = BIG ? l2R[b2] : lds_R3[2];
            STAMP(a, 5);
            PRIO_AT(5);
            // all 2 NK count-searches of the thread descend together (staged_search)
            int soff[NK][2];
            if (ABL(a, 2)) {
#pragma unroll
                for (int k = 0; k < NK; ++k) { soff[k][0] = (int)(d2bits(tau[k][0]) >> 20) & 2047; soff[k][1] = (int)(d2bits(tau[k][1]) >> 20) & 2047; }
            } else staged_search<TILE, NK>(tau, span, Pm, T0, T1, R0, R1, R2, lds_stage, soff);
            // gather: the ancestor's state at a 32-bit byte offset from a uniform base (no 64-bit address arithmetic per lane)
            const unsigned char* xbase = reinterpret_cast<const unsigned char*>(xin_r - win0);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    int anc = bb_min * TILE + soff[k][c];
                    anc = anc < a.N - 1 ? anc : a.N - 1;
                    const int i = i_first + (k * NT + tid) * 2 + c;
                    if (anc_p && i < a.N) anc_p[rowoff + i - out0] = (uint32_t)anc;
                    xin[k][c] = *reinterpret_cast<const double*>(xbase + ((uint32_t)anc << 3));
                    if constexpr (DX > 1) srcv[k][c] = anc;
                    lw_old[k][c] = 0.0;
                }
            }
        } else {
            // general path (iid multinomial, or an output tile spanning many cdf tiles): probes in L2
#pragma unroll
            for (int k = 0; k < NK; ++k) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double target = tau[k][c];
                    int bb = count_less_pow2(a.Bpow2, target, [&](int j) { return BIG ? (j < a.B ? l2Tg(j) : dinf()) : lds_T[j]; });
                    bb = bb < a.B - 1 ? bb : a.B - 1;
                    const double Pb = bb ? (BIG ? l2Tg(bb - 1) : lds_T[bb - 1]) : 0.0;
                    const double tloc = __builtin_ceil((target - Pb) * (BIG ? l2R[bb] : lds_R[bb]));
                    const double* tile = cdf_r + (size_t)(bb - a.win_tile0) * TILE;
                    const int j = count_less_pow2(TILE, tloc, [&](int q) { return tile[q]; });
                    int anc = bb * TILE + j;
                    anc = anc < a.N - 1 ? anc : a.N - 1;
                    const int i = i_first + (k * NT + tid) * 2 + c;
                    if (anc_p && i < a.N) anc_p[rowoff + i - out0] = (uint32_t)anc;
                    xin[k][c] = xin_r[anc - win0];
                    if constexpr (DX > 1) srcv[k][c] = anc;
                    lw_old[k][c] = 0.0;
                }
            }
        }
    }
    STAMP(a, 6);
    PRIO_AT(6);

    // --- standard normals (Box-Muller); this arithmetic hides the latency of the ancestor gather ---
    double zn[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (ABL(a, 0)) { zn[k][0] = 0.25 + 1e-9 * (double)nw0[k]; zn[k][1] = -0.25; }
        else pair_normals(nw0[k], nw1[k], &lds_dtab, &zn[k][0], &zn[k][1]);
    }
    STAMP(a, 7);
    PRIO_AT(7);

    // --- fSamp / q1Samp, logGEv, tile max ---
    double lg[NK][2];
    double mx = -dinf();
    bool nan = false;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = i_first + (k * NT + tid) * 2;
        double xo[2];
        double xov[VEC ? kMaxDim : 1][2];  // components 1 .. of a vector state
        if constexpr (VEC) {
#if SSME_HAS_USER_MODEL
            // vector state / observation (user model with dim_x or dim_y > 1): component 0 travels the scalar path above (gather, the
            // pair's Box-Muller draw); components d >= 1 are gathered from their planes at the same source index and take their
            // normals from one more Philox call per pair each (counter stream STREAM_XDIM + d)
            double yv[kMaxDim];
            yv[0] = y;
#pragma unroll
            for (int d = 1; d < DY; ++d) yv[d] = a.by_value ? a.y_now_v[d - 1] : a.y[(size_t)a.yi * DY + d];
            double znd[kMaxDim][2];
            znd[0][0] = zn[k][0]; znd[0][1] = zn[k][1];
#pragma unroll
            for (int d = 1; d < DX; ++d) {
                const u32x4 o = philox4x32_10((uint32_t)(b * (TILE / 2) + k * NT + tid), (uint32_t)a.t, rep, (uint32_t)(STREAM_XDIM + d), key0, key1);
                pair_normals(o.v0, o.v1, &lds_dtab, &znd[d][0], &znd[d][1]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double xv[kMaxDim], zv[kMaxDim], xn[kMaxDim];
                xv[0] = xin[k][c];
#pragma unroll
                for (int d = 1; d < DX; ++d) xv[d] = first_step ? 0.0 : a.x_in[(size_t)d * a.xplane + rowoff + (size_t)srcv[k][c]];
#pragma unroll
                for (int d = 0; d < DX; ++d) zv[d] = znd[d][c];
                if (first_step) user_calls<ssme_user_model0>::init_vec(mc, zv, xn);
                else user_calls<ssme_user_model0>::prop_vec(mc, xv, zv, zcov, xn, lds_etab);
                const double gv = mc.bad ? -dinf() : user_calls<ssme_user_model0>::logg_vec(mc, yv, xn, lds_etab);
                xo[c] = xn[0];
#pragma unroll
                for (int d = 1; d < DX; ++d) xov[d][c] = xn[d];
                lg[k][c] = lw_old[k][c] + gv;
            }
#endif
        } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double xn = first_step ? zn[k][c] * mc.a2 : model_prop<MODEL>(mc, xin[k][c], zn[k][c], zcov, lds_etab);
            const double l = lw_old[k][c] + (ABL(a, 3) ? -0.5 * xn * xn : model_logg<MODEL>(mc, y, xn, lds_etab));
            xo[c] = xn;
            lg[k][c] = l;
        }
        }
        if (ragged) {                     // particles beyond N: state 0, log-weight -inf (weight 0, no part in the maximum)
            asm volatile("");
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (!((i0 + c) < a.N)) {
                    xo[c] = 0.0; lg[k][c] = -dinf();
                    if constexpr (DX > 1) { for (int d = 1; d < DX; ++d) xov[d][c] = 0.0; }
                }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { const double l = lg[k][c]; nan = nan || (l != l); mx = (l > mx) ? l : mx; }
        const size_t idx = rowoff + (size_t)(i0 - out0);
        store_pair(reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(a.x_out + rowoff + (size_t)(i_first - out0)) + (uint32_t)(k * NT + tid) * 16u),
                   xo[0], xo[1], a.stream_stores);
        if (logw_p) *reinterpret_cast<double2*>(logw_p + idx) = make_double2(lg[k][0], lg[k][1]);
        if constexpr (DX > 1) {
#pragma unroll
            for (int d = 1; d < DX; ++d) *reinterpret_cast<double2*>(a.x_out + (size_t)d * a.xplane + idx) = make_double2(xov[d][0], xov[d][1]);
        }
    }
    STAMP(a, 8);
    PRIO_AT(8);
    const double mb = block_max_nanprop<NT>(mx, nan, lds_d2);
    STAMP(a, 9);
    PRIO_AT(9);

    // --- tile-local fixed-point weights and their exact inclusive scan (the next step's cdf) ---
    double q[NK][2], inc[NK][2], total;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (ABL(a, 4)) { q[k][0] = (double)(d2bits(lg[k][0] - mb) >> 24); q[k][1] = (double)(d2bits(lg[k][1] - mb) >> 24); }
        else {
            // (a particle beyond N carries log-weight -inf: the clamped exp makes its q exactly 0, no mask needed)
            q[k][0] = __builtin_rint(dexp_scaled_t(lg[k][0] - mb, kTileShift, lds_etab));
            q[k][1] = __builtin_rint(dexp_scaled_t(lg[k][1] - mb, kTileShift, lds_etab));
        }
    }
    if (ABL(a, 5)) {
#pragma unroll
        for (int k = 0; k < NK; ++k) { inc[k][0] = q[k][0]; inc[k][1] = q[k][0] + q[k][1]; }
        total = inc[0][1] + 1048576.0;
    } else block_scan_f64<NT, NK>(q, inc, total, lds_seg_c);
    STAMP(a, 12);
    PRIO_AT(12);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        store_pair(reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(a.cdf_out + rowoff + (size_t)(i_first - out0)) + (uint32_t)(k * NT + tid) * 16u),
                   inc[k][0], inc[k][1], a.stream_stores);
    }
    STAMP(a, 10);
    PRIO_AT(10);
    bool fused_accounting = false;
    if constexpr (!HOT && !BIG) fused_accounting = a.ticket != nullptr;
    if (tid == 0) {
        double* ts = a.tsum_out + (size_t)r * a.Bs + (b - a.tile0);
        double* tm = a.tmax_out + (size_t)r * a.Bs + (b - a.tile0);
        if (fused_accounting) {      // device-coherent stores: another workgroup (another XCD, another L2) reads them in this launch
            __hip_atomic_store(ts, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tm, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { *ts = total; *tm = mb; }
    }
    if constexpr (!HOT && !BIG) {
        if (fused_accounting) {
            // Step API (mod.filter(y); mod.getLogCondLike()): the workgroup that arrives last at its filter's counter does
            // what kf_finalize does in a second launch -- level-2 of the weights just written, log p(y_t | y_{1:t-1}) --
            // so that a filter() call is ONE launch.  Only the tile sums / maxima cross workgroups: they are written and read
            // with device-scope accesses and ordered by the counter, so no cache write-back of the particle arrays is needed.
            // Ordering.  The formally ordered form -- an ACQ_REL read-modify-write of the counter at agent scope -- was measured
            // (round 3, tools/step_latency.cpp): the release half is a write-back of the XCD's whole L2 (buffer_wbl2 sc1) and the
            // acquire half an invalidate (buffer_inv sc1), i.e. the cache write-back of the particle arrays this design exists
            // to avoid: filter() 27.8 -> 40.0 us at N = 2^20, 41 -> 55 us for 64 filters x 2^14 (profiles/r03_step_api_handover.txt).
            // Kept instead: agent-scope RELAXED atomic stores of the two values (global_store_dwordx2 ... sc1: written through
            // to the memory side, past the non-coherent L2), an explicit s_waitcnt vmcnt(0) (both stores acknowledged before
            // the next instruction issues -- the compiler's workgroup-scope release fence emits NO wait outside tgsplit mode,
            // so round 2's form left the two stores and the counter's atomic free to reach their different L2 channels in any
            // order; seen in the ISA this round, profiles/r03_step_api_handover.txt), then the agent-scope relaxed atomic add;
            // the last arriver reads with agent-scope atomic loads (global_load_dwordx2 ... sc1: not served from its own L2),
            // issued only after the atomic's return value has arrived.  The C++ memory model gives relaxed atomics on different
            // addresses no inter-thread order; this form leans on the completion wait instead of a scope-wide fence.  Cost: one
            // thread per workgroup waits for two store acknowledgements at the end of the kernel (measured: none,
            // profiles/r03_step_api_handover.txt).
            __shared__ int lds_last;
            if (tid == 0) {
                int last = 1;
                if (gridDim.x > 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the two stores above have been acknowledged
                    last = __hip_atomic_fetch_add(a.ticket + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
                }
                lds_last = last;
            }
            __syncthreads();
            if (lds_last) {
                constexpr int NE2 = 2048 / NT;
                double A2[NE2], M2[NE2], Ap2[NE2], Tinc2[NE2], S2, m2;
#pragma unroll
                for (int e = 0; e < NE2; ++e) {
                    const int j = e * NT + tid;
                    A2[e] = 0.0; M2[e] = 0.0;
                    if (j < a.B) {
                        A2[e] = __hip_atomic_load(a.tsum_out + (size_t)r * a.Bs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        M2[e] = __hip_atomic_load(a.tmax_out + (size_t)r * a.Bs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                level2_scan<NT>(A2, M2, a.B, a.rshift, m2, Ap2, Tinc2, S2, lds_d1, lds_seg_l2, lds_etab);
                if (tid == 0) {
                    FilterScalars* sc = a.scal + r;
                    const bool resample_now = ((a.t + 1) % a.resamp_sched == 0);
                    const double Sd = (S2 > 0.0) ? dldexp(S2, -a.rshift) : dnan();
                    const double lse = m2 + dlog(Sd);
                    const double ll = lse - sc->prev;
                    sc->m = m2;
                    sc->S = S2;
                    sc->last_ll = ll;
                    sc->loglik = sc->loglik + ll;
                    sc->prev = resample_now ? a.logN : lse;
                    if (a.per_step) a.per_step[(size_t)r * a.Tcap + a.t] = ll;
                    if (gridDim.x > 1) __hip_atomic_store(a.ticket + r, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch (kernel boundary orders it)
                    if (a.ll_host) a.ll_host[r] = ll;
                }
            }
        }
    }
#ifdef SSME_ABLATE
    __syncthreads();
    STAMP(a, 11);
    PRIO_AT(11);
#endif
}

// ---------------------------------------------------------------------------------------
// KF: account the last step's log conditional likelihood.  grid = (R), block = 256.
// Reads the tile sums / maxima the last k_filter_step wrote (passed as tsum_in / tmax_in).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void kf_finalize(const StepArgs a) {
    __shared__ double lds_seg[128];
    __shared__ double lds_d[16];
    const int tid = threadIdx.x;
    const int r = blockIdx.x;
    double A2[8], Ap[8], Tinc[8], M2[8], S, m;
    level2_load<kThreads>(a.tsum_in + (size_t)r * a.Bs, a.tmax_in + (size_t)r * a.Bs, a.B, A2, M2);
    level2_scan<kThreads>(A2, M2, a.B, a.rshift, m, Ap, Tinc, S, lds_d, lds_seg, kExpTable);
    if (tid == 0) {
        FilterScalars* sc = a.scal + r;
        const bool resample_now = ((a.t + 1) % a.resamp_sched == 0);
        const double Sd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
        const double lse = m + dlog(Sd);
        const double ll = lse - sc->prev;
        sc->m = m;
        sc->S = S;
        sc->last_ll = ll;
        sc->loglik = sc->loglik + ll;
        sc->prev = resample_now ? a.logN : lse;
        if (a.per_step) a.per_step[(size_t)r * a.Tcap + a.t] = ll;
        if (a.ll_host) a.ll_host[r] = ll;
    }
}

// ---------------------------------------------------------------------------------------
// Split level-2 for filters of more than 2048 tiles (N > 2^22; also selectable for tests): ONE workgroup per filter
// does what every workgroup of k_filter_step otherwise repeats -- global max, rescaled integer tile sums A', their exact
// inclusive scan T', the total S', A/A' -- and additionally the source-tile range [lo, hi] of every output tile
// (same tile_target_bounds, same counts), and accounts log p(y_{t-1} | .).  Same arithmetic as level2_scan: integer
// sums are exact and associative, so the results are those of the in-kernel level-2 to the bit.
// grid = (R), block = 1024, dynamic LDS = Bpow2 doubles.  a.t = the step about to run (its targets); plan_ranges = 0
// for the call that only accounts the last step's log-likelihood (the kf_finalize role).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_level2_plan(const StepArgs a, int plan_ranges) {
    constexpr int NT = 1024, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l2[];
    double* lds_T = reinterpret_cast<double*>(smem_l2);          // [Bpow2]
    __shared__ double lds_d[16];
    __shared__ double lds_seg[16];
    const int tid = threadIdx.x, r = blockIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* ts = a.tsum_in + (size_t)r * a.Bs;
    const double* tm = a.tmax_in + (size_t)r * a.Bs;
    double* Tp = a.l2_T + (size_t)r * a.Bs;
    double* Rp = a.l2_R + (size_t)r * a.Bs;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const bool resampled = (a.t % a.resamp_sched == 0);
    // This kernel is ONE workgroup per filter and bound by memory round trips, not by arithmetic: every tile sum and maximum
    // is therefore requested up front (tile j = k * 1024 + tid, k < 16: coalesced, all loads in flight at once) and the
    // max, the rescale and the scan run from registers.
    constexpr int KMAX = kMaxTilesSplit / NT;                    // 16
    double Areg[KMAX], Mreg[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int j = k * NT + tid;
        Areg[k] = 0.0; Mreg[k] = -dinf();
        if (j < a.B) { Areg[k] = ts[j]; Mreg[k] = tm[j]; }
    }
    // global max of the tile maxima, NaN propagating
    double mx = -dinf();
    bool nan = false;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k * NT + tid < a.B) { const double v = Mreg[k]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    }
    const double m = block_max_nanprop<NT>(mx, nan, lds_d);
    // rescaled tile sums and their exact inclusive scan, 1024 tiles per round
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k * NT < a.Bpow2) {
            const int j = k * NT + tid;
            double A = 0.0, Ap = 0.0;
            if (j < a.B) {
                A = Areg[k];
                const double ex = dexp_scaled_t(Mreg[k] - m, a.rshift - kTileShift, kExpTable);
                Ap = __builtin_rint(A * ex);
            }
            const double inc = wave_incl_scan_f64(Ap);
            __syncthreads();                                     // lds_seg of the previous round has been read
            if (lane == 63) lds_seg[wave] = inc;
            __syncthreads();
            double sv = (lane & 15) < NW ? lds_seg[lane & 15] : 0.0;
            sv = sv + dpp_f64_zero<0x111, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x112, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x114, 0xF>(sv);
            sv = sv + dpp_f64_zero<0x118, 0xF>(sv);
            const double pre = wave ? readlane_f64(sv, wave - 1) : 0.0;
            const double T = (carry + pre) + inc;
            carry = carry + readlane_f64(sv, 15);
            if (j < a.Bpow2) lds_T[j] = (j < a.B) ? T : dinf();
            if (j < a.B) { Tp[j] = T; Rp[j] = A / Ap; }
        }
    }
    const double S = carry;
    __syncthreads();
    if (plan_ranges && resampled && a.resampler != RESAMP_MULTINOMIAL_IID) {
        const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
        double G = 1.0, u0 = 0.0;
        const size_t g0 = ((size_t)a.gi * a.R + r) * a.B;
        if (a.resampler == RESAMP_MULTINOMIAL) G = a.gtot[(size_t)a.gi * a.R + r];
        else if (a.resampler == RESAMP_SYSTEMATIC) {
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, key0, key1);
            u0 = u01_co(ox.v0, ox.v1);
        }
        // The source-tile range [lo, hi] of every output tile.  The bounds grow with the tile index (t_lo(b) <= t_hi(b),
        // t_lo(b) <= t_lo(b + 1)), so a thread takes CONSECUTIVE tiles and every count after its first starts where the
        // last one ended: a short gallop instead of a log2(B)-level descent (this phase is bound by the LDS traffic of one
        // CU: 2 x 14 probes per tile at B = 16384 before, 3-4 now).  The Gamma prefixes of a thread's tiles are all
        // requested before the first use (Areg / Mreg are dead by now).
        const int per = (a.B + NT - 1) / NT;                     // <= KMAX
        const int bb0 = tid * per;
        double pgreg[KMAX + 1];
#pragma unroll
        for (int k = 0; k <= KMAX; ++k) {
            const int b = bb0 + k;
            pgreg[k] = 0.0;
            if (a.resampler == RESAMP_MULTINOMIAL && k <= per && b <= a.B) pgreg[k] = (b < a.B) ? a.pgam[g0 + b] : G;
        }
        // #{ j : T'_j < t } given that every entry below `from` is < t (entries past B are +inf)
        auto count_from = [&](int from, double t) {
            int p = from, sz = 1;
            while (p + sz - 1 < a.Bpow2 && lds_T[p + sz - 1] < t) { p += sz; sz <<= 1; }
            for (sz >>= 1; sz >= 1; sz >>= 1)
                if (p + sz - 1 < a.Bpow2 && lds_T[p + sz - 1] < t) p += sz;
            return p;
        };
        int prev = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int b = bb0 + k;
            if (k < per && b < a.B) {
                const int i_first = b * a.tile;
                const int nvalid = (a.N - i_first) < a.tile ? (a.N - i_first) : a.tile;
                double ts_, t_lo, t_hi;
                tile_target_bounds(a.resampler, S, a.N, i_first, nvalid, pgreg[k], pgreg[k + 1], G, u0, ts_, t_lo, t_hi);
                const int lo = (k == 0) ? count_less_pow2(a.Bpow2, t_lo, [&](int j) { return lds_T[j]; }) : count_from(prev, t_lo);
                const int hi = count_from(lo, t_hi);
                prev = lo;
                a.l2_lo[(size_t)r * a.Bs + b] = lo < a.B - 1 ? lo : a.B - 1;
                a.l2_hi[(size_t)r * a.Bs + b] = hi < a.B - 1 ? hi : a.B - 1;
            }
        }
    }
    if (tid == 0) {
        FilterScalars* sc = a.scal + r;
        sc->m = m;
        sc->S = S;
        if (a.finalize_prev) {
            const double Sdd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
            const double lse = m + dlog(Sdd);
            const double ll = lse - sc->prev;
            sc->last_ll = ll;
            sc->loglik = sc->loglik + ll;
            sc->prev = resampled ? a.logN : lse;
            if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
            if (a.ll_host) a.ll_host[r] = ll;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same split level-2 over SEVERAL workgroups, for filters of more than 1024 tiles.  k_level2_plan is one workgroup per
// filter and its time grows with the number of tiles (B = 4096: 15 us, 16384: 51 us per launch -- a sixth of the step at
// N >= 2^23, and every rank of a sharded filter plans ALL tiles); here workgroup g takes tiles [1024 g, 1024 g + 1024):
//   k_l2_scan_blocks  global max (every workgroup reads all maxima itself), rescaled sums A', A / A', the inclusive scan
//                     of its own 1024 tiles and their total
//   k_l2_ranges       offsets from the (at most 16) totals, T' = local scan + offset for all tiles into LDS and for its
//                     own tiles into l2_T, source ranges of its own tiles, and (workgroup 0) the accounting.
// Integer sums are exact, so T', S', the ranges and the log-likelihood are k_level2_plan's to the bit.
// grid = (ceil(B / 1024), R), block = 1024; k_l2_ranges: dynamic LDS = Bpow2 doubles.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_l2_scan_blocks(const StepArgs a) {
    constexpr int NT = 1024, NW = NT / 64, KMAX = kMaxTilesSplit / NT;
    __shared__ double lds_d[16];
    __shared__ double lds_seg[16];
    const int tid = threadIdx.x, blk = blockIdx.x, r = blockIdx.y;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* ts = a.tsum_in + (size_t)r * a.Bs;
    const double* tm = a.tmax_in + (size_t)r * a.Bs;
    double Mreg[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int j = k * NT + tid;
        Mreg[k] = (j < a.B) ? tm[j] : -dinf();
    }
    const int jo = blk * NT + tid;
    const double A = (jo < a.B) ? ts[jo] : 0.0;
    const double Mo = (jo < a.B) ? tm[jo] : 0.0;
    double mx = -dinf();
    bool nan = false;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k * NT + tid < a.B) { const double v = Mreg[k]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    }
    const double m = block_max_nanprop<NT>(mx, nan, lds_d);
    double Ap = 0.0;
    if (jo < a.B) Ap = __builtin_rint(A * dexp_scaled_t(Mo - m, a.rshift - kTileShift, kExpTable));
    const double inc = wave_incl_scan_f64(Ap);
    if (lane == 63) lds_seg[wave] = inc;
    __syncthreads();
    double sv = (lane & 15) < NW ? lds_seg[lane & 15] : 0.0;
    sv = sv + dpp_f64_zero<0x111, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x112, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x114, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x118, 0xF>(sv);
    const double pre = wave ? readlane_f64(sv, wave - 1) : 0.0;
    double* Tloc = a.l2_work + (size_t)r * a.Bs;
    double* blkv = a.l2_work + (size_t)a.R * a.Bs + (size_t)r * kL2Scratch;
    if (jo < a.B) {
        Tloc[jo] = pre + inc;
        a.l2_R[(size_t)r * a.Bs + jo] = A / Ap;
    }
    if (tid == 0 && !a.l2_inkernel) {
        blkv[blk] = readlane_f64(sv, 15);
        if (blk == 0) blkv[KMAX] = m;
    }
    if (wave == 0 && a.l2_inkernel) {
        // One launch: the workgroup that arrives last adds up the block totals.  The total goes out as an agent-scope store and
        // has left the CU (s_waitcnt) before the arrival is counted; the last arriver reads the totals with agent-scope loads
        // issued after its own count came back -- the hand-over of the step API's ticket (k_filter_step), at 16 workgroups.
        // Lane k of its first wave takes block k: one load, a 16-lane scan (integer sums: exact in any order).
        const int nblk = (int)gridDim.x;
        const double tot = readlane_f64(sv, 15);
        int arrived = 0;
        if (lane == 0) {
            __hip_atomic_store(&blkv[blk], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            arrived = __hip_atomic_fetch_add(a.l2_ticket + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == nblk - 1) {
            const double v = (lane < nblk) ? __hip_atomic_load(&blkv[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            double tinc = v;
            tinc = tinc + dpp_f64_zero<0x111, 0xF>(tinc);
            tinc = tinc + dpp_f64_zero<0x112, 0xF>(tinc);
            tinc = tinc + dpp_f64_zero<0x114, 0xF>(tinc);
            tinc = tinc + dpp_f64_zero<0x118, 0xF>(tinc);
            const double S = readlane_f64(tinc, 15);
            double* off = blkv + kL2Offsets;
            if (lane <= KMAX) off[lane] = (lane < KMAX) ? tinc - v : S;       // exclusive offset of block k (k >= nblk: S')
            if (lane == 0) {
                __hip_atomic_store(a.l2_ticket + r, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool resampled = (a.t % a.resamp_sched == 0);
                FilterScalars* sc = a.scal + r;
                sc->m = m;
                sc->S = S;
                if (a.finalize_prev) {
                    const double Sdd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
                    const double lse = m + dlog(Sdd);
                    const double ll = lse - sc->prev;
                    sc->last_ll = ll;
                    sc->loglik = sc->loglik + ll;
                    sc->prev = resampled ? a.logN : lse;
                    if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
                    if (a.ll_host) a.ll_host[r] = ll;
                }
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_l2_ranges(const StepArgs a, int plan_ranges) {
    constexpr int NT = 1024, KMAX = kMaxTilesSplit / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l2r[];
    double* lds_T = reinterpret_cast<double*>(smem_l2r);         // [Bpow2]
    const int tid = threadIdx.x, blk = blockIdx.x, r = blockIdx.y;
    const double* Tloc = a.l2_work + (size_t)r * a.Bs;
    const double* blkv = a.l2_work + (size_t)a.R * a.Bs + (size_t)r * kL2Scratch;
    double* Tp = a.l2_T + (size_t)r * a.Bs;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const bool resampled = (a.t % a.resamp_sched == 0);
    const int nblk = (a.B + NT - 1) / NT;
    // local scans of all tiles, requested before anything else
    double Treg[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int j = k * NT + tid;
        Treg[k] = (j < a.B) ? Tloc[j] : 0.0;
    }
    double off[KMAX + 1];
    off[0] = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) off[k + 1] = off[k] + ((k < nblk) ? blkv[k] : 0.0);
    const double S = off[KMAX];
    const double m = blkv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int j = k * NT + tid;
        if (j < a.Bpow2) {
            const double T = off[k] + Treg[k];
            lds_T[j] = (j < a.B) ? T : dinf();
            if (k == blk && j < a.B) Tp[j] = T;
        }
    }
    __syncthreads();
    const int b = blk * NT + tid;
    if (plan_ranges && resampled && a.resampler != RESAMP_MULTINOMIAL_IID && b < a.B) {
        const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
        double G = 1.0, u0 = 0.0, pg = 0.0, pgn = 0.0;
        const size_t g0 = ((size_t)a.gi * a.R + r) * a.B;
        if (a.resampler == RESAMP_MULTINOMIAL) {
            G = a.gtot[(size_t)a.gi * a.R + r];
            pg = a.pgam[g0 + b]; pgn = (b + 1 < a.B) ? a.pgam[g0 + b + 1] : G;
        } else if (a.resampler == RESAMP_SYSTEMATIC) {
            const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, rep, STREAM_RESAMP_EXTRA, key0, key1);
            u0 = u01_co(ox.v0, ox.v1);
        }
        const int i_first = b * a.tile;
        const int nvalid = (a.N - i_first) < a.tile ? (a.N - i_first) : a.tile;
        double ts_, t_lo, t_hi;
        tile_target_bounds(a.resampler, S, a.N, i_first, nvalid, pg, pgn, G, u0, ts_, t_lo, t_hi);
        int lo = 0, hi = 0;
        for (int step = a.Bpow2 >> 1; step >= 1; step >>= 1) {       // the two counts descend together
            if (lds_T[lo + step - 1] < t_lo) lo += step;
            if (lds_T[hi + step - 1] < t_hi) hi += step;
        }
        a.l2_lo[(size_t)r * a.Bs + b] = lo < a.B - 1 ? lo : a.B - 1;
        a.l2_hi[(size_t)r * a.Bs + b] = hi < a.B - 1 ? hi : a.B - 1;
    }
    if (blk == 0 && tid == 0) {
        FilterScalars* sc = a.scal + r;
        sc->m = m;
        sc->S = S;
        if (a.finalize_prev) {
            const double Sdd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
            const double lse = m + dlog(Sdd);
            const double ll = lse - sc->prev;
            sc->last_ll = ll;
            sc->loglik = sc->loglik + ll;
            sc->prev = resampled ? a.logN : lse;
            if (a.per_step) a.per_step[(size_t)r * a.Tcap + (a.t - 1)] = ll;
            if (a.ll_host) a.ll_host[r] = ll;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Particle-sharded filter (one filter over `world` GPUs, B/world tiles each): the range of SOURCE tiles
// [lo, hi] that the output tiles of every rank touch at step a.t, from the gathered tile sums / maxima.
// Runs the same level-2 and the same target bounds as k_filter_step, so the ranges are exact.
// grid = 1, block = 512, dynamic LDS = max(Bpow2, 2) doubles.  lo_hi: [world][2] ints.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_shard_plan(const StepArgs a, int world, int Bl, int32_t* lo_hi, int margin = 0, int32_t* flag = nullptr) {
    constexpr int NT = 512, NE = 2048 / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_plan[];
    double* lds_T = reinterpret_cast<double*>(smem_plan);
    __shared__ double lds_seg[64];
    __shared__ double lds_d[16];
    const int tid = threadIdx.x;
    double A2[NE], M2[NE], Ap[NE], Tinc[NE], S, m;
    level2_load<NT>(a.tsum_in, a.tmax_in, a.B, A2, M2);
    level2_scan<NT>(A2, M2, a.B, a.rshift, m, Ap, Tinc, S, lds_d, lds_seg, kExpTable);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = e * NT + tid;
        if (j < a.Bpow2) lds_T[j] = (j < a.B) ? Tinc[e] : dinf();
    }
    __syncthreads();
    if (tid < world) {
        const int bF = tid * Bl, bL = bF + Bl - 1 < a.B - 1 ? bF + Bl - 1 : a.B - 1;      // Bl = ceil(B / world): the last rank may own fewer
        int lo = 0, hi = a.B - 1;
        if (a.resampler != RESAMP_MULTINOMIAL_IID) {
            const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
            double G = 1.0, pgF = 0.0, pgFn = 0.0, pgL = 0.0, pgLn = 0.0, u0 = 0.0;
            if (a.resampler == RESAMP_MULTINOMIAL) {
                const size_t g0 = (size_t)a.gi * a.B;                       // R = 1
                G = a.gtot[a.gi];
                pgF = a.pgam[g0 + bF]; pgFn = (bF + 1 < a.B) ? a.pgam[g0 + bF + 1] : G;
                pgL = a.pgam[g0 + bL]; pgLn = (bL + 1 < a.B) ? a.pgam[g0 + bL + 1] : G;
            } else if (a.resampler == RESAMP_SYSTEMATIC) {
                const u32x4 ox = philox4x32_10(0u, (uint32_t)a.t, a.first_filter, STREAM_RESAMP_EXTRA, key0, key1);
                u0 = u01_co(ox.v0, ox.v1);
            }
            const int iF = bF * a.tile, iL = bL * a.tile;
            const int nvF = (a.N - iF) < a.tile ? (a.N - iF) : a.tile, nvL = (a.N - iL) < a.tile ? (a.N - iL) : a.tile;
            double ts, t_lo, t_hi, unused;
            tile_target_bounds(a.resampler, S, a.N, iF, nvF, pgF, pgFn, G, u0, ts, t_lo, unused);
            tile_target_bounds(a.resampler, S, a.N, iL, nvL, pgL, pgLn, G, u0, ts, unused, t_hi);
            lo = count_less_pow2(a.Bpow2, t_lo, [&](int j) { return lds_T[j]; });
            hi = count_less_pow2(a.Bpow2, t_hi, [&](int j) { return lds_T[j]; });
            lo = lo < a.B - 1 ? lo : a.B - 1;
            hi = hi < a.B - 1 ? hi : a.B - 1;
        }
        lo_hi[2 * tid] = lo;
        lo_hi[2 * tid + 1] = hi;
        if (flag) {
            // C++ shard driver: does this rank's window stay inside its fixed halo?  flag[0]: some window left it;
            // flag[1] / flag[2]: widest reach left / right of the own tiles seen so far
            const int left = bF - lo, right = hi - bL;
            if (left > margin || right > margin) atomicOr(&flag[0], 1);
            if (left > 0) atomicMax(&flag[1], left);
            if (right > 0) atomicMax(&flag[2], right);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Gamma tables for the multinomial resampler (data independent: seed, t, filter, tile only).
// k_gamma_draw: grid = (ceil(B/256), nT, R).  k_gamma_prefix: one workgroup per (ti, r).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_gamma_draw(double* gam, int N, int B, int R, int t0, const uint32_t* keyp,
                                                         uint32_t first_filter, uint32_t gamma_stream, int tile) {
    const uint32_t key0 = keyp[0], key1 = keyp[1];
    const int b = blockIdx.x * kThreads + threadIdx.x;
    const int ti = blockIdx.y, r = blockIdx.z;
    if (b >= B) return;
    const int nb = (N - b * tile) < tile ? (N - b * tile) : tile;
    gam[((size_t)ti * R + r) * B + b] = gamma_draw((uint32_t)b, (uint32_t)(t0 + ti), first_filter + (uint32_t)r, key0, key1, (double)nb,
                                                   gamma_stream);
}

// One workgroup per (ti, r) row.  The exclusive prefix is the SEQUENTIAL sum run = run + g[b] (its rounding is part of
// the specification), so one lane adds; but the row is brought into LDS by all 256 threads first and the prefixes leave
// through LDS as well, in chunks of 2048 tiles: the adding lane then sees LDS latency instead of one dependent global
// load per tile (a single filter step of the step API spent ~100 us here at 512 tiles).
__global__ __launch_bounds__(kThreads) void k_gamma_prefix(const double* gam, double* pgam, double* gtot, int B, int R,
                                                           int nT, int t0, const uint32_t* keyp, uint32_t first_filter,
                                                           uint32_t extra_stream) {
    __shared__ double buf[2048];
    const int id = blockIdx.x, tid = threadIdx.x;
    const int ti = id / R, r = id % R;
    const double* g = gam + (size_t)id * B;
    double* p = pgam + (size_t)id * B;
    double run = 0.0;                                  // meaningful in thread 0 only
    for (int c0 = 0; c0 < B; c0 += 2048) {
        const int n = (B - c0) < 2048 ? (B - c0) : 2048;
        for (int j = tid; j < n; j += kThreads) buf[j] = g[c0 + j];
        __syncthreads();
        if (tid == 0) {
            for (int j = 0; j < n; ++j) { const double v = buf[j]; buf[j] = run; run = run + v; }
        }
        __syncthreads();
        for (int j = tid; j < n; j += kThreads) p[c0 + j] = buf[j];
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t key0 = keyp[0], key1 = keyp[1];
        const u32x4 ox = philox4x32_10(0u, (uint32_t)(t0 + ti), first_filter + (uint32_t)r, extra_stream, key0, key1);
        gtot[id] = run + (-dlog_pn(u01_oc(ox.v0, ox.v1)));
    }
}

// the same for short rows (few tiles per filter, many filters): one thread per (ti, r)
__global__ __launch_bounds__(kThreads) void k_gamma_prefix_rows(const double* gam, double* pgam, double* gtot, int B, int R,
                                                                int nT, int t0, const uint32_t* keyp, uint32_t first_filter,
                                                                uint32_t extra_stream) {
    const uint32_t key0 = keyp[0], key1 = keyp[1];
    const int id = blockIdx.x * kThreads + threadIdx.x;
    if (id >= nT * R) return;
    const int ti = id / R, r = id % R;
    const double* g = gam + (size_t)id * B;
    double* p = pgam + (size_t)id * B;
    double run = 0.0;
    for (int b = 0; b < B; ++b) { p[b] = run; run = run + g[b]; }
    const u32x4 ox = philox4x32_10(0u, (uint32_t)(t0 + ti), first_filter + (uint32_t)r, extra_stream, key0, key1);
    gtot[id] = run + (-dlog_pn(u01_oc(ox.v0, ox.v1)));
}

// ---------------------------------------------------------------------------------------
// Weighted expectations of built-in functionals with the last step's (pre-resampling) weights
// (getExpectations(); twin liu_west_filter.h:1662-1683; callers pswarm_filter.h:87-89,383-385).
// The weights are the fixed-point weights the resampler uses: w_j = q_j exp(m_tile - m), q_j = cdf_j - cdf_{j-1}.
//   k_expect_partials  grid (B tiles, R): per tile sum_j h_f(x_j) q_j for ALL requested functionals at once (the tile's
//                      denominator sum_j q_j is its exact integer tile sum A_b, already in memory); every CU takes part
//   k_expect_final     grid (R): E_f = sum_b num_{b,f} e^{m_b - m} / sum_b A_b e^{m_b - m}
//   k_expect_mean      grid (1): the swarm aggregate, the plain mean over the R members (pswarm_filter.h:103,136)
// Summation trees are fixed (per-thread strided, wave xor tree, waves in order), so results are reproducible.
// ---------------------------------------------------------------------------------------
constexpr int kMaxFunctionals = 4;
struct FunctionalIds { int32_t n; int32_t id[kMaxFunctionals]; };

__device__ __forceinline__ double builtin_h(int functional, double xv) {
    return functional == 0 ? xv : functional == 1 ? xv * xv : functional == 2 ? dexp(0.5 * xv) : 42.0;
}
__device__ __forceinline__ double wave_sum_xor(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, kWave);
    return v;
}

__global__ __launch_bounds__(kThreads) void k_expect_partials(const double* x, const double* cdf, int N, int Npad, int Bs, int tile,
                                                              FunctionalIds fs, double* part /*[R][Bs][kMaxFunctionals]*/) {
    __shared__ double lds[kMaxFunctionals][4];
    const int tid = threadIdx.x, b = blockIdx.x, r = blockIdx.y;
    const double* xr = x + (size_t)r * Npad + (size_t)b * tile;
    const double* cr = cdf + (size_t)r * Npad + (size_t)b * tile;
    const int nvalid = (N - b * tile) < tile ? (N - b * tile) : tile;
    double num[kMaxFunctionals] = {0.0, 0.0, 0.0, 0.0};
    for (int j = tid; j < nvalid; j += kThreads) {
        {
            const double q = cr[j] - (j ? cr[j - 1] : 0.0);
            const double xv = xr[j];
#pragma unroll
            for (int f = 0; f < kMaxFunctionals; ++f)
                if (f < fs.n) num[f] = num[f] + builtin_h(fs.id[f], xv) * q;
        }
    }
#pragma unroll
    for (int f = 0; f < kMaxFunctionals; ++f) {
        num[f] = wave_sum_xor(num[f]);
        if ((tid & 63) == 0) lds[f][tid >> 6] = num[f];
    }
    __syncthreads();
    if (tid < kMaxFunctionals)
        part[((size_t)r * Bs + b) * kMaxFunctionals + tid] = ((lds[tid][0] + lds[tid][1]) + lds[tid][2]) + lds[tid][3];
}

__global__ __launch_bounds__(kThreads) void k_expect_final(const double* part, const double* tsum, const double* tmax, int B, int Bs,
                                                           int R, FunctionalIds fs, double* out /*[n][R]*/) {
    __shared__ double lds[kMaxFunctionals + 1][4];
    __shared__ double lds_m[16];
    const int tid = threadIdx.x, r = blockIdx.x;
    double mx = -dinf();
    bool nan = false;
    for (int j = tid; j < B; j += kThreads) { const double v = tmax[(size_t)r * Bs + j]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    const double m = block_max_nanprop<kThreads>(mx, nan, lds_m);
    double acc[kMaxFunctionals + 1] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int j = tid; j < B; j += kThreads) {
        const double sc = dexp(tmax[(size_t)r * Bs + j] - m);
#pragma unroll
        for (int f = 0; f < kMaxFunctionals; ++f)
            if (f < fs.n) acc[f] = acc[f] + part[((size_t)r * Bs + j) * kMaxFunctionals + f] * sc;
        acc[kMaxFunctionals] = acc[kMaxFunctionals] + tsum[(size_t)r * Bs + j] * sc;
    }
#pragma unroll
    for (int f = 0; f <= kMaxFunctionals; ++f) {
        acc[f] = wave_sum_xor(acc[f]);
        if ((tid & 63) == 0) lds[f][tid >> 6] = acc[f];
    }
    __syncthreads();
    if (tid < fs.n) {
        const double n4 = ((lds[tid][0] + lds[tid][1]) + lds[tid][2]) + lds[tid][3];
        const double d4 = ((lds[kMaxFunctionals][0] + lds[kMaxFunctionals][1]) + lds[kMaxFunctionals][2]) + lds[kMaxFunctionals][3];
        out[(size_t)tid * R + r] = (m != m) ? dnan() : n4 / d4;
    }
}

// Swarm aggregation in one launch: block f < n: mean over the R members of expectation row f ([n][R] in exp_rows);
// block n: mean of the members' last log conditional likelihoods.  out: n + 1 doubles at
// out[0..n) and out[slot_ll] (device-mapped host memory: the host polls it).  grid (n + 1), block 256.
// num_threads > 0: the reference's mean of per-thread means -- split_data_thread_pool deals member i to thread i % num_threads
// (thread_pool.h:443-447), every thread averages its own members (intra_agg_func, pswarm_filter.h:130-160) and the thread
// averages are averaged (inter_agg_func, :96-128).  That is a weighted mean with weight 1 / (num_threads * members of the
// thread), equal to the plain mean only when num_threads divides R.  num_threads <= 0: the plain mean.
__global__ __launch_bounds__(kThreads) void k_swarm_means(const double* exp_rows, const FilterScalars* scal, int R, int n, int slot_ll,
                                                          double* out, int num_threads) {
    __shared__ double lds[4];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int T = (num_threads > 0 && num_threads < R) ? num_threads : (num_threads >= R ? R : 0);
    double s = 0.0;
    for (int r = tid; r < R; r += kThreads) {
        const double v = (f < n) ? exp_rows[(size_t)f * R + r] : scal[r].last_ll;
        if (T > 0) {
            const int members = R / T + ((r % T) < (R % T) ? 1 : 0);
            s = s + v / (double)members;
        } else s = s + v;
    }
    s = wave_sum_xor(s);
    if ((tid & 63) == 0) lds[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[f < n ? f : slot_ll] = (((lds[0] + lds[1]) + lds[2]) + lds[3]) / (double)(T > 0 ? T : R);
}

// Normalisable weights of one filter for host-side functionals (arbitrary std::function h: pswarm_filter.h:44,87-89):
// w_j = q_j exp(m_tile - m) 2^-41  (= exp(logw_j - max logw) to 2^-41).  grid (B), block 256.
__global__ __launch_bounds__(kThreads) void k_weights(const double* cdf, const double* tmax, int N, int B, int tile, double* w) {
    __shared__ double lds_m[16];
    const int tid = threadIdx.x, b = blockIdx.x;
    double mx = -dinf();
    bool nan = false;
    for (int j = tid; j < B; j += kThreads) { const double v = tmax[j]; nan = nan || (v != v); mx = (v > mx) ? v : mx; }
    const double m = block_max_nanprop<kThreads>(mx, nan, lds_m);
    const double sc = (m != m) ? dnan() : dexp_scaled(tmax[b] - m, -kTileShift);
    const double* cr = cdf + (size_t)b * tile;
    for (int j = tid; j < tile; j += kThreads)
        if (b * tile + j < N) w[(size_t)b * tile + j] = (cr[j] - (j ? cr[j - 1] : 0.0)) * sc;
}

}  // namespace ssme
