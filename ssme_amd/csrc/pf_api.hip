// pf_api.hip -- extern "C" boundary (include/ssme_pf.h) over the gfx950 kernels.
// Host side of the drop-in: owns device buffers, the HIP stream, the hipGraph of a series.
// No torch types; no CPU fallback: without a HIP device every call fails with SSME_ERR_HIP.
#include "../../include/ssme_pf.h"
#include "pf_kernels.h"
#include "lw_kernels.h"
#include "pf_small.h"
#include "shard_driver.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace ssme;

struct ssme_pf_s {
    ssme_pf_config cfg;
    int N, R, Npad, B, Bs, Bpow2, rshift;
    int tile;                // particles per tile: 2048, 1024 or 512 (cfg.tile_particles, or by N: default_tile)
    size_t lds_bytes;
    int t;                   // next time index
    bool params_set;
    int debug_anc, keep_logw;
    int graph_mode;
    int shard_rank, shard_world;   // particle-sharded filter: this handle computes tiles [rank*Bl, rank*Bl + sh_Bown); world = 0: unsharded
    int sh_Bl, sh_Bown;            // Bl = ceil(B / world) tiles per rank in every layout (gathers, halos); the last rank owns B - (world-1) Bl >= 1 of them
    hipStream_t own_stream;  // the stream created with the handle (stream may be replaced by ssme_pf_set_stream)
    int32_t* plan_dev;       // [world][2] source-tile ranges (k_shard_plan)
    int split_l2;            // 1: level-2 by k_level2_plan (filters of more than 2048 tiles, or forced by set_debug bit 2)
    double *l2_T, *l2_R;     // [R][Bs] split level-2 outputs
    double* l2_work;         // [R][Bs] + [R][32]: scratch of the multi-workgroup level-2
    int32_t *l2_lo, *l2_hi;
    int32_t* l2_ticket;      // [R] arrival counters of the one-launch level-2 (StepArgs::l2_inkernel)
    int l2_tables;           // set_debug bit 4: the split level-2 writes its tables and ranges (k_level2_plan / k_l2_ranges), as sharded filters do
    size_t lds_bytes_big, lds_bytes_plan;
    double* small_ms;        // [R][tcap][2] scratch of the one-tile whole-series kernel
    double* yz_step;         // device [2]: y and z of the step API, uploaded by ONE copy
    int32_t* ticket;         // device [R]: arrival counters of the step API's in-kernel accounting (zero between launches)
    double* pin;             // pinned, device-mapped host staging: [2 .. 2+R) log conditional likelihoods of the step API (written by the accounting kernel)
    double* pin_dev;         // the same memory as the device sees it
    int gamma_t0, gamma_rows;   // step API: the Gamma tables hold time indices gamma_t0 .. gamma_t0 + gamma_rows - 1
    int num_cus;             // compute units of the device (priority schedule of the step kernel)
    int small_series;        // 1: one-tile filters run the whole series in one launch (k_filter_series_small)
    double y_step_v[3];      // step API: components 1 .. of this call's vector observation
    int dx, dy;              // state / observation dimension: 1, or the user model's dim_x / dim_y (model_api.h)
    int nt;                  // threads per 2048-particle tile of k_filter_step (256, 512, 1024)
    // C++ shard driver (ssme_pf_shard_run_series): halo buffers [margin | own tiles | margin] x 2048 doubles, ping-pong;
    // this rank's tile sums / maxima, their gathered and repacked forms; exact-path window buffers; flag + statistics
    double *sh_x[2], *sh_c[2], *sh_loc, *sh_raw, *sh_tsum, *sh_tmax, *sh_winx, *sh_winc;
    int32_t* sh_flag;        // [0] a window left the halo ON THIS RANK, [1] / [2] widest reach left / right of the own tiles (in tiles),
                             // [3] max of [0] over all ranks (ncclAllReduce after the time loop): what the fallback decision reads
    int32_t sh_stats[4];     // host copy of sh_flag after the last native series
    int sh_margin, sh_rows, sh_path;   // sh_path: path of the last native series (1 fixed halo, 2 exact)
    int sh_check;            // 1 while the driver's fixed-halo path launches a step: the kernel verifies its source tiles
    long sh_exchanged;       // tiles received from other ranks during the last native series
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    float last_ms;
    // device state; [2] = ping-pong (step t reads cur, writes cur ^ 1)
    double* x[2];
    double* cdf[2];          // integer-valued doubles (< 2^53)
    double* tsum[2];
    double* tmax[2];
    double *logw, *ybuf, *zbuf, *per_step, *scratchR;
    double *exp_part, *exp_out;  // expectations: [R][Bs][4] per-tile numerators; [5][R] per-filter values + [5] means over filters
    double* wscratch;            // [Npad] weights of one filter (host-side functionals), allocated on first use
    double *gam, *pgam, *gtot;   // Gamma tables of the multinomial resampler, gcap time rows
    uint32_t* anc;
    uint32_t* keybuf;        // [2] Philox key = seed (lo, hi)
    FilterScalars* scal;
    ModelConst* mc;
    int cur;                 // buffer index holding the latest step's output
    int ycap, tcap, gcap;
    // graph cache
    hipGraphExec_t gexec;
    int g_T, g_has_z, g_debug, g_logw, g_nt;
    std::vector<ModelConst> h_mc;
    std::string err;
};

static int fail(ssme_pf_handle h, int code, const char* what, hipError_t e) {
    if (h) { h->err = std::string(what) + ": " + hipGetErrorString(e); }
    return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, SSME_ERR_HIP, #call, e_); } while (0)

// Wait for a stream with the latency of a poll: hipStreamSynchronize spins only briefly and then sleeps on an interrupt,
// which adds ~200 us to a filter() call whose kernels take longer than that window (measured: 512 filters x 2^14
// particles, 146 vs 420 us per call).  Poll for up to ~5 ms, then fall back to the blocking wait.
constexpr uint64_t kStepPending = 0x7ff8a5a5c3c3e1e1ull;     // "result not written yet" in the step API's mapped result buffer

static hipError_t wait_stream_low_latency(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(s);
    }
}

// Step APIs: the R results are awaited where they land.  The host marks the mapped result slots kStepPending (a NaN
// payload no computation produces) before the launch and polls them instead of the stream's completion signal, which
// trails the kernel's last store by its end-of-kernel cache write-back.  The stream stays ordered: the next call
// queues behind this one.  A launch that takes longer than 5 ms falls back to the stream (and reports its errors).
static void mark_results_pending(double* slots, int R) {
    volatile uint64_t* res = reinterpret_cast<volatile uint64_t*>(slots);
    for (int r = 0; r < R; ++r) res[r] = kStepPending;
}
static hipError_t wait_results(hipStream_t s, const double* slots, int R) {
    const volatile uint64_t* res = reinterpret_cast<const volatile uint64_t*>(slots);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (int r = 0; r < R;) {
        if (res[r] != kStepPending) { ++r; continue; }
        if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(s);
    }
    return hipSuccess;
}

// SSME_F32 handles (the reference instantiated with float_t = float, example/main.cpp:13): float at the boundary --
// observations, covariates and parameters are rounded to float on entry, every value handed back is rounded to float --
// while the arithmetic stays fp64.  Plain fp32 VALU instructions issue at the fp64 rate on gfx950 (only packed fp32 is
// faster), so float arithmetic would buy no time; it would only coarsen the callbacks below the fixed-point weights.
static inline double f32r(double v) { return (double)(float)v; }
static void round_out(const ssme_pf_handle h, double* p, size_t n) {
    if (h->cfg.dtype == SSME_F32 && p) for (size_t i = 0; i < n; ++i) p[i] = f32r(p[i]);
}

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
static int ceil_log2(int n) { int k = 0; while ((1ll << k) < n) ++k; return k; }

// Derived constants; operation order mirrors oracle/ssme_oracle.cpp derive().
static ModelConst derive(int model, const double* th) {
    ModelConst c{};
#if SSME_HAS_USER_MODEL
    if (model == SSME_MODEL_USER0) return ssme_user_model0::derive(th);
#endif
    if (model == SSME_MODEL_SVOL) {            // (beta, phi, sigma)
        const double beta = th[0], phi = th[1], sigma = th[2];
        c.a0 = phi; c.a1 = sigma;
        c.a2 = sigma / dsqrt(1.0 - phi * phi);
        c.a3 = dlog(beta);
        c.a4 = 1.0 / (beta * beta);
        c.bad = !(beta > 0.0);
    } else if (model == SSME_MODEL_SVOL_LEVERAGE) {   // (phi, mu, sigma, rho)
        const double phi = th[0], mu = th[1], sigma = th[2], rho = th[3];
        c.a0 = phi; c.a1 = mu;
        c.a2 = sigma / dsqrt(1.0 - phi * phi);
        c.a3 = sigma * dsqrt(1.0 - phi * phi);
        c.a4 = rho * sigma;
        c.bad = 0;
    } else {                                    // (phi, sigma, tau)
        const double phi = th[0], sigma = th[1], tau = th[2];
        c.a0 = phi; c.a1 = sigma;
        c.a2 = sigma / dsqrt(1.0 - phi * phi);
        c.a3 = dlog(tau);
        c.a4 = 1.0 / tau;
        c.bad = !(tau > 0.0);
    }
    return c;
}

constexpr int kStepGammaChunk = 64;
// state / observation dimension of a model: 1 for the built-in models, dim_x / dim_y of a compiled-in user model
static void dims_of(int model, int* dx, int* dy) {
    *dx = 1; *dy = 1;
#if SSME_HAS_USER_MODEL
    if (model == SSME_MODEL_USER0) { *dx = user_dims<ssme_user_model0>::dx; *dy = user_dims<ssme_user_model0>::dy; }
#endif
}
static int n_theta_of(int model) {
#if SSME_HAS_USER_MODEL
    if (model == SSME_MODEL_USER0) return ssme_user_model0::n_theta;
#endif
    return model == SSME_MODEL_SVOL_LEVERAGE ? 4 : 3;
}
static bool logw_needed(ssme_pf_handle h) { return h->keep_logw || h->cfg.resamp_sched > 1; }

// arguments of the step that reads buffers `cur` and writes `cur ^ 1`
static StepArgs step_args(ssme_pf_handle h) {
    StepArgs a{};
    const int i = h->cur, o = h->cur ^ 1;
    a.x_in = h->x[i]; a.x_out = h->x[o]; a.xplane = (size_t)h->R * h->Npad;
    a.cdf_in = h->cdf[i]; a.cdf_out = h->cdf[o];
    a.tsum_in = h->tsum[i]; a.tsum_out = h->tsum[o];
    a.tmax_in = h->tmax[i]; a.tmax_out = h->tmax[o];
    a.logw = logw_needed(h) ? h->logw : nullptr;
    a.anc = h->debug_anc ? h->anc : nullptr;
    a.scal = h->scal; a.mc = h->mc; a.y = h->ybuf; a.z = nullptr; a.per_step = nullptr;
    a.gam = h->gam; a.pgam = h->pgam; a.gtot = h->gtot;
    a.N = h->N; a.Npad = h->Npad; a.B = h->B; a.Bs = h->Bs; a.Bpow2 = h->Bpow2; a.rshift = h->rshift; a.R = h->R;
    a.tile = h->tile;
    a.Tcap = h->tcap;
    a.resampler = h->cfg.resampler; a.resamp_sched = h->cfg.resamp_sched;
    a.keyp = h->keybuf; a.first_filter = h->cfg.first_filter_id;
    a.logN = dlog((double)h->N);
    a.l2_T = h->l2_T; a.l2_R = h->l2_R; a.l2_lo = h->l2_lo; a.l2_hi = h->l2_hi; a.l2_work = h->l2_work;
    a.l2_inkernel = (h->split_l2 && h->shard_world == 0 && !h->l2_tables) ? 1 : 0; a.l2_ticket = h->l2_ticket;
    {
        double* tail = h->l2_work + (size_t)h->R * h->Bs;                     // [R][kL2Scratch] and 32 zeros after them
        a.l2_tsrc = a.l2_inkernel ? h->l2_work : h->l2_T;
        a.l2_offs = a.l2_inkernel ? tail + kL2Offsets : tail + (size_t)h->R * kL2Scratch;
        a.l2_offs_stride = a.l2_inkernel ? kL2Scratch : 0;
    }
    {
        // two 512-thread workgroups fit a CU (LDS): is the whole grid resident at once?
        const long blocks = (long)(h->shard_world > 0 ? h->sh_Bown : h->B) * h->R;
        a.prio_mode = (blocks <= 2L * h->num_cus) ? 1 : 2;
        static const char* force = getenv("SSME_PRIO_MODE");         // measurement aid: 0 none, 1 / 2 the two schedules, 3.. experimental
        if (force) a.prio_mode = atoi(force);
        // measured (profiles/r02_stream_stores.txt): 2048-particle tiles gain 7-10 % up to 3 workgroups per CU, nothing
        // or -3 % beyond; the smaller tiles (mid-size handles, latency bound) are indifferent
        a.stream_stores = (h->tile == kTile && blocks <= 3L * h->num_cus) ? 1 : 0;
        static const char* force_ss = getenv("SSME_STREAM_STORES");  // measurement aid: 0 / 1
        if (force_ss) a.stream_stores = atoi(force_ss);
    }
#ifdef SSME_ABLATE
    { const char* e = getenv("SSME_ABLATE_MASK"); a.ablate = e ? atoi(e) : 0; }
    {
        static unsigned long long* g_stamps = nullptr;       // diagnostic build only
        if (getenv("SSME_STAMPS")) {
            if (!g_stamps) { hipMalloc(&g_stamps, sizeof(unsigned long long) * 16 * (size_t)h->B * h->R); }
            a.stamps = g_stamps;
            if (getenv("SSME_STAMPS_DUMP")) {   // dump the stamps of the last launch (call after a sync)
                std::vector<unsigned long long> hs(16 * (size_t)h->B * h->R);
                hipMemcpy(hs.data(), g_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
                FILE* f = fopen(getenv("SSME_STAMPS_DUMP"), "w");
                for (size_t i = 0; i < hs.size(); i += 16) { for (int k = 0; k < 16; ++k) fprintf(f, "%llu ", hs[i + k]); fprintf(f, "\n"); }
                fclose(f);
            }
        }
    }
#endif
    return a;
}

// Tile size when the caller does not choose (part of the arithmetic specification: the oracle applies the same rule).
// 2048-particle tiles are the efficient shape at full occupancy (fewer, fatter workgroups; N <= 2048 is the whole-series
// kernel).  Smaller handles get the smallest tile that still leaves every workgroup resident at once, so that mid-size
// filters spread over the 256 CUs: 512-particle tiles while they make at most 256 workgroups (one filter of 2^16: 128
// workgroups instead of 32, 10.3 -> 7.4 us per step), 1024-particle tiles while those make at most 512 (one filter of
// 2^18: 10.7 -> 8.4 us; 2^19: 11.5 -> 10.8 us), 2048 beyond (profiles/r02_tile_sweep.txt).  The rule reads N and the
// size of the BANK of filters (ssme_pf_config::n_filters_total, else the handle's n_filters): a bank that is split over
// GPUs or over one handle per member declares its total, so a filter's bits do not depend on the split.
static int default_tile(int n_particles, int n_filters) {
    if (n_particles <= kTile) return kTile;
    const long R = n_filters;
    if (R * ((n_particles + kTileSmall - 1) / kTileSmall) <= 256) return kTileSmall;   // at most one 256-thread block per CU
    if (R * ((n_particles + kTileMid - 1) / kTileMid) <= 512) return kTileMid;          // at most two 512-thread blocks per CU
    return kTile;
}

// One launcher per instantiation of the step kernel.  The dynamic-LDS ceiling of a kernel is process-wide state: it is
// only ever raised (a handle with few tiles must not lower what a handle with many tiles was granted).
static thread_local int g_grant_only = 0;      // != 0: launch_k only raises the LDS ceiling (handle creation); 1 general kernel, 2 / 3 the RS = 0 / 1 variants
template <int MODEL, int NT, bool BIG, int TILE, int RS, bool WL2 = false>
static void launch_k(ssme_pf_handle h, const StepArgs& a, dim3 grid, size_t lds) {
    static std::atomic<size_t> granted{0};
    auto kern = &k_filter_step<MODEL, NT, BIG, TILE, RS, WL2>;
    if (lds > granted.load(std::memory_order_relaxed)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted.store(lds, std::memory_order_relaxed);
    }
    if (g_grant_only) return;
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, h->stream, a);
}

// RS: the common configurations (multinomial -- the reference's -- or systematic resampling, every step, t > 0, no debug
// outputs) have their own instantiations; returns the RS template argument or -1 for the general kernel
static int hot_config(ssme_pf_handle h, const StepArgs& a) {
    if (g_grant_only) return g_grant_only - 2;                       // 1 -> -1 (general), 2 -> 0, 3 -> 1
    if (h->cfg.resamp_sched != 1 || a.t <= 0 || a.anc || a.logw || a.ticket) return -1;
    return h->cfg.resampler == SSME_RESAMP_MULTINOMIAL ? 0 : (h->cfg.resampler == SSME_RESAMP_SYSTEMATIC ? 1 : -1);
}

template <int MODEL, int NT, bool BIG, int TILE, bool WL2>
static void launch_rs_w(ssme_pf_handle h, const StepArgs& a, dim3 grid, size_t lds, int rs) {
    if (rs == 0) launch_k<MODEL, NT, BIG, TILE, 0, WL2>(h, a, grid, lds);
    else if (rs == 1) launch_k<MODEL, NT, BIG, TILE, 1, WL2>(h, a, grid, lds);
    else launch_k<MODEL, NT, BIG, TILE, -1, WL2>(h, a, grid, lds);
}
// WL2: filters of at most 128 tiles take the level-2 wave by wave (k_filter_step); its own instantiations
template <int MODEL, int NT, bool BIG, int TILE>
static void launch_rs(ssme_pf_handle h, const StepArgs& a, dim3 grid, size_t lds, int rs) {
    if (BIG) { launch_rs_w<MODEL, NT, BIG, TILE, false>(h, a, grid, lds, rs); return; }
    if (g_grant_only || a.B <= 128) launch_rs_w<MODEL, NT, false, TILE, true>(h, a, grid, lds, rs);
    if (g_grant_only || a.B > 128) launch_rs_w<MODEL, NT, false, TILE, false>(h, a, grid, lds, rs);
}

template <int MODEL>
static void launch_step_grid(ssme_pf_handle h, const StepArgs& a, dim3 grid) {
    const int rs = hot_config(h, a);
    if (h->tile == kTileSmall) {
        if (h->split_l2) launch_rs<MODEL, 256, true, kTileSmall>(h, a, grid, h->lds_bytes_big, rs);
        else launch_rs<MODEL, 256, false, kTileSmall>(h, a, grid, h->lds_bytes, rs);
        return;
    }
    if (h->tile == kTileMid) {      // 1024-particle tiles: 512 threads, one particle pair each
        if (h->split_l2) launch_rs<MODEL, 512, true, kTileMid>(h, a, grid, h->lds_bytes_big, rs);
        else launch_rs<MODEL, 512, false, kTileMid>(h, a, grid, h->lds_bytes, rs);
        return;
    }
    if (h->split_l2) { launch_rs<MODEL, 512, true, kTile>(h, a, grid, h->lds_bytes_big, rs); return; }
    switch (h->nt) {
        case 256: launch_k<MODEL, 256, false, kTile, -1>(h, a, grid, h->lds_bytes); break;
        case 512: launch_rs<MODEL, 512, false, kTile>(h, a, grid, h->lds_bytes, rs); break;
        default: launch_k<MODEL, 1024, false, kTile, -1>(h, a, grid, h->lds_bytes); break;
    }
}
static void launch_step_on(ssme_pf_handle h, const StepArgs& a, dim3 grid) {
    switch (h->cfg.model) {
        case SSME_MODEL_SVOL: launch_step_grid<MODEL_SVOL>(h, a, grid); break;
        case SSME_MODEL_SVOL_LEVERAGE: launch_step_grid<MODEL_SVOL_LEVERAGE>(h, a, grid); break;
#if SSME_HAS_USER_MODEL
        case SSME_MODEL_USER0: launch_step_grid<MODEL_USER0>(h, a, grid); break;
#endif
        default: launch_step_grid<MODEL_LIN_GAUSS>(h, a, grid); break;
    }
}
static void launch_step(ssme_pf_handle h, const StepArgs& a) { launch_step_on(h, a, dim3(h->B, h->R)); }
// every instantiation this handle can launch gets its LDS ceiling now (not inside a stream capture)
static void grant_step_lds(ssme_pf_handle h) {
    const int nt0 = h->nt, sp0 = h->split_l2;
    StepArgs a{};
    for (int hot = 1; hot <= 3; ++hot)
        for (int sp = 0; sp < 2; ++sp)
            for (int nt : {256, 512, 1024}) {
                g_grant_only = hot; h->split_l2 = sp; h->nt = nt;
                launch_step_on(h, a, dim3(1, 1));
            }
    g_grant_only = 0; h->nt = nt0; h->split_l2 = sp0;
}
// Gamma tables for time indices t0 .. t0+nT-1 into table rows 0 .. nT-1
static void launch_gamma(ssme_pf_handle h, int t0, int nT) {
    if (h->cfg.resampler != SSME_RESAMP_MULTINOMIAL) return;
    hipLaunchKernelGGL(k_gamma_draw, dim3((h->B + kThreads - 1) / kThreads, nT, h->R), dim3(kThreads), 0, h->stream,
                       h->gam, h->N, h->B, h->R, t0, (const uint32_t*)h->keybuf, h->cfg.first_filter_id, (uint32_t)STREAM_GAMMA, h->tile);
    if (h->B <= 64)
        hipLaunchKernelGGL(k_gamma_prefix_rows, dim3((nT * h->R + kThreads - 1) / kThreads), dim3(kThreads), 0, h->stream,
                           h->gam, h->pgam, h->gtot, h->B, h->R, nT, t0, (const uint32_t*)h->keybuf, h->cfg.first_filter_id,
                           (uint32_t)STREAM_RESAMP_EXTRA);
    else
        hipLaunchKernelGGL(k_gamma_prefix, dim3(nT * h->R), dim3(kThreads), 0, h->stream,
                           h->gam, h->pgam, h->gtot, h->B, h->R, nT, t0, (const uint32_t*)h->keybuf, h->cfg.first_filter_id,
                           (uint32_t)STREAM_RESAMP_EXTRA);
}
// the split level-2 of one draw: one workgroup per filter up to 1024 tiles, several above (k_l2_scan_blocks + k_l2_ranges)
static void launch_level2(hipStream_t st, const StepArgs& a, int n_filters, size_t lds, int ranges) {
    if (a.l2_inkernel) {
        // unsharded bootstrap filter: one launch; k_filter_step<.., true> finds the source-tile ranges itself
        hipLaunchKernelGGL(k_l2_scan_blocks, dim3((a.B + 1023) / 1024, n_filters), dim3(1024), 0, st, a);
    } else if (a.l2_work && a.B > 1024) {
        const dim3 grid((a.B + 1023) / 1024, n_filters);
        hipLaunchKernelGGL(k_l2_scan_blocks, grid, dim3(1024), 0, st, a);
        hipLaunchKernelGGL(k_l2_ranges, grid, dim3(1024), lds, st, a, ranges);
    } else
        hipLaunchKernelGGL(k_level2_plan, dim3(n_filters), dim3(1024), lds, st, a, ranges);
}
// accounts the log conditional likelihood of step t from the buffers the step wrote (now `cur`)
// split level-2 of the buffers `cur` holds: for the step t about to run (plan_ranges) or only the accounting of step t-1
static void launch_plan(ssme_pf_handle h, int t, int gi, bool plan_ranges, bool finalize_prev, bool record_per_step,
                        double* ll_host = nullptr) {
    StepArgs a = step_args(h);
    a.t = t; a.gi = gi; a.finalize_prev = finalize_prev ? 1 : 0;
    a.ll_host = ll_host;
    a.per_step = record_per_step ? h->per_step : nullptr;
    launch_level2(h->stream, a, h->R, h->lds_bytes_plan, plan_ranges ? 1 : 0);
}
static void launch_kf(ssme_pf_handle h, int t, bool record_per_step, double* ll_host = nullptr) {
    if (h->split_l2) { launch_plan(h, t + 1, 0, false, true, record_per_step, ll_host); return; }
    StepArgs a = step_args(h);
    a.ll_host = ll_host;
    a.t = t;
    a.per_step = record_per_step ? h->per_step : nullptr;
    hipLaunchKernelGGL(kf_finalize, dim3(h->R), dim3(kThreads), 0, h->stream, a);
}

// enqueue one filter step at time index t reading y[yi] and gamma-table row gi
static void enqueue_step(ssme_pf_handle h, int t, int yi, int gi, bool has_z, bool finalize_prev, bool record_per_step,
                         bool from_step_staging = false, bool results_to_host = true) {
    StepArgs a = step_args(h);
    a.z = has_z ? h->zbuf : nullptr;
    if (from_step_staging) {
        a.by_value = 1; a.y_now = h->pin[0]; a.z_now = has_z ? h->pin[1] : 0.0;
        for (int d = 1; d < h->dy; ++d) a.y_now_v[d - 1] = h->y_step_v[d - 1];
        if (!h->split_l2) { a.ticket = h->ticket; a.ll_host = results_to_host ? h->pin_dev + 2 : nullptr; }     // accounting inside the step kernel (its last workgroup)
    }
    a.per_step = record_per_step ? h->per_step : nullptr;
    a.t = t; a.yi = yi; a.gi = gi; a.finalize_prev = finalize_prev ? 1 : 0;
    if (h->split_l2 && t > 0) launch_plan(h, t, gi, true, finalize_prev, record_per_step);
    launch_step(h, a);
    h->cur ^= 1;
}

// whole series in one launch for one-tile filters (pf_small.h); reads buffers `cur`-independent, writes x[1] etc.
template <int MODEL>
static void launch_small_m(ssme_pf_handle h, const StepArgs& a, int T) {
    const dim3 grid(h->R);
    // one particle per lane up to 512 particles (twice the waves, about half the chain per wave: N = 100 3.0 -> 2.2 us per step,
    // N = 500 3.3 -> 3.1); pairs above (N = 1000: 4.6 against 4.9 us; profiles/r02_small_series.txt)
    if (h->N <= 64) hipLaunchKernelGGL((k_filter_series_lane<MODEL, 64>), grid, dim3(64), 0, h->stream, a, T);
    else if (h->N <= 128) hipLaunchKernelGGL((k_filter_series_lane<MODEL, 128>), grid, dim3(128), 0, h->stream, a, T);
    else if (h->N <= 256) hipLaunchKernelGGL((k_filter_series_lane<MODEL, 256>), grid, dim3(256), 0, h->stream, a, T);
    else if (h->N <= 512) hipLaunchKernelGGL((k_filter_series_lane<MODEL, 512>), grid, dim3(512), 0, h->stream, a, T);
    else if (h->N <= 1024) hipLaunchKernelGGL((k_filter_series_small<MODEL, 512, 1>), grid, dim3(512), 0, h->stream, a, T);
    else hipLaunchKernelGGL((k_filter_series_small<MODEL, 512, 2>), grid, dim3(512), 0, h->stream, a, T);
}
static void enqueue_series_small(ssme_pf_handle h, int T, bool has_z) {
    launch_gamma(h, 0, T);
    h->cur = 0;
    StepArgs a = step_args(h);
    a.z = has_z ? h->zbuf : nullptr;
    a.per_step = h->per_step;
    a.small_ms = h->small_ms;
    switch (h->cfg.model) {
        case SSME_MODEL_SVOL: launch_small_m<MODEL_SVOL>(h, a, T); break;
        case SSME_MODEL_SVOL_LEVERAGE: launch_small_m<MODEL_SVOL_LEVERAGE>(h, a, T); break;
#if SSME_HAS_USER_MODEL
        case SSME_MODEL_USER0: launch_small_m<MODEL_USER0>(h, a, T); break;
#endif
        default: launch_small_m<MODEL_LIN_GAUSS>(h, a, T); break;
    }
    h->cur = 1;
}

// Gamma tables of the multinomial resampler for T time rows (nothing else moves: the step API grows them on its own)
static int ensure_gamma_capacity(ssme_pf_handle h, int T) {
    if (h->cfg.resampler == SSME_RESAMP_MULTINOMIAL && T > h->gcap) {
        if (h->gam) hipFree(h->gam);
        if (h->pgam) hipFree(h->pgam);
        if (h->gtot) hipFree(h->gtot);
        h->gam = h->pgam = h->gtot = nullptr;
        HIPCHK(hipMalloc(&h->gam, sizeof(double) * (size_t)T * h->R * h->B));
        HIPCHK(hipMalloc(&h->pgam, sizeof(double) * (size_t)T * h->R * h->B));
        HIPCHK(hipMalloc(&h->gtot, sizeof(double) * (size_t)T * h->R));
        h->gcap = T;
        if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    }
    return SSME_OK;
}

static int ensure_series_capacity(ssme_pf_handle h, int T) {
    if (T > h->ycap) {
        if (h->ybuf) hipFree(h->ybuf);
        if (h->zbuf) hipFree(h->zbuf);
        h->ybuf = h->zbuf = nullptr;
        HIPCHK(hipMalloc(&h->ybuf, sizeof(double) * T * h->dy));
        HIPCHK(hipMalloc(&h->zbuf, sizeof(double) * T));
        h->ycap = T;
        if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    }
    int rcg = ensure_gamma_capacity(h, T);
    if (rcg != SSME_OK) return rcg;
    if (T > h->tcap) {
        if (h->per_step) hipFree(h->per_step);
        if (h->small_ms) hipFree(h->small_ms);
        h->per_step = h->small_ms = nullptr;
        HIPCHK(hipMalloc(&h->per_step, sizeof(double) * (size_t)T * h->R));
        if (h->B == 1) HIPCHK(hipMalloc(&h->small_ms, sizeof(double) * (size_t)T * h->R * 2));
        h->tcap = T;
        if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    }
    return SSME_OK;
}

static int ensure_logw(ssme_pf_handle h) {
    if (logw_needed(h) && !h->logw) {
        const size_t np = (size_t)h->R * h->Npad;
        HIPCHK(hipMalloc(&h->logw, sizeof(double) * np));
        HIPCHK(hipMemset(h->logw, 0, sizeof(double) * np));
    }
    return SSME_OK;
}

static int upload_key(ssme_pf_handle h) {
    const uint32_t k[2] = {(uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32)};
    HIPCHK(hipMemcpyAsync(h->keybuf, k, sizeof(k), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // k is a temporary
    return SSME_OK;
}

static int do_reset(ssme_pf_handle h) {
    std::vector<FilterScalars> sc(h->R);
    const double logN = dlog((double)h->N);
    for (auto& s : sc) { std::memset(&s, 0, sizeof(s)); s.prev = logN; }
    HIPCHK(hipMemcpyAsync(h->scal, sc.data(), sizeof(FilterScalars) * h->R, hipMemcpyHostToDevice, h->stream));
    // arrival counters of the in-step accounting and of the one-launch level-2: every launch leaves them at zero; a reset does too
    if (h->ticket) HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int32_t) * h->R, h->stream));
    if (h->l2_ticket) HIPCHK(hipMemsetAsync(h->l2_ticket, 0, sizeof(int32_t) * h->R, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // sc is a temporary
    h->t = 0; h->cur = 0;
    h->gamma_rows = 0;                          // tables are redrawn for the new stream / the new series
    return SSME_OK;
}

extern "C" {

int ssme_pf_version(void) { return 332; }

const char* ssme_pf_strerror(int s) {
    switch (s) {
        case SSME_OK: return "ok";
        case SSME_ERR_INVALID_ARG: return "invalid argument";
        case SSME_ERR_LENGTH: return "length error (empty data)";
        case SSME_ERR_UNSUPPORTED: return "unsupported configuration";
        case SSME_ERR_HIP: return "HIP runtime error";
        case SSME_ERR_STATE: return "invalid call order";
        default: return "unknown status";
    }
}
const char* ssme_pf_last_error(ssme_pf_handle h) { return h ? h->err.c_str() : ""; }

static int create_impl(const ssme_pf_config* cfg, int shard_rank, int shard_world, ssme_pf_handle* out) {
    if (!cfg || !out) return SSME_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->n_particles < 1 || cfg->n_filters < 1 || cfg->n_filters > 65535) return SSME_ERR_INVALID_ARG;
    if (cfg->model == SSME_MODEL_USER0 && !SSME_HAS_USER_MODEL) return SSME_ERR_UNSUPPORTED;   // this library was built without a user model (model_api.h)
    if (cfg->model < 0 || cfg->model > SSME_MODEL_USER0) return SSME_ERR_INVALID_ARG;
    if (cfg->resampler < 0 || cfg->resampler > SSME_RESAMP_MULTINOMIAL_IID) return SSME_ERR_INVALID_ARG;
    if (cfg->resamp_sched < 1) return SSME_ERR_INVALID_ARG;
    if (cfg->dtype != SSME_F64 && cfg->dtype != SSME_F32) return SSME_ERR_INVALID_ARG;
    if (cfg->dtype == SSME_F32 && shard_world > 0) return SSME_ERR_UNSUPPORTED;     // the sharded entry points exchange raw fp64 arrays
    {
        int dx = 1, dy = 1;
        dims_of(cfg->model, &dx, &dy);
        if ((dx > 1 || dy > 1) && (shard_world > 0 || cfg->dtype == SSME_F32)) return SSME_ERR_UNSUPPORTED;   // vector models: unsharded, fp64 boundary
    }
    if (cfg->tile_particles != 0 && cfg->tile_particles != kTile && cfg->tile_particles != kTileSmall && cfg->tile_particles != kTileMid) return SSME_ERR_INVALID_ARG;
    if (cfg->n_filters_total < 0 || (cfg->n_filters_total > 0 && cfg->n_filters_total < cfg->n_filters)) return SSME_ERR_INVALID_ARG;
    const int tile = shard_world > 0 ? kTile : (cfg->tile_particles ? cfg->tile_particles
                                                : default_tile(cfg->n_particles, cfg->n_filters_total > 0 ? cfg->n_filters_total : cfg->n_filters));
    const int B = (cfg->n_particles + tile - 1) / tile;
    if (B > kMaxTilesSplit) return SSME_ERR_UNSUPPORTED;   // at most 16384 tiles per filter (N <= 2^25 with 2048-particle tiles)
    ssme_pf_handle h = new (std::nothrow) ssme_pf_s();
    if (!h) return SSME_ERR_INVALID_ARG;
    h->cfg = *cfg;
    h->shard_rank = shard_rank; h->shard_world = shard_world;
    if (shard_world > 0) {
        h->sh_Bl = (B + shard_world - 1) / shard_world;
        h->sh_Bown = B - shard_rank * h->sh_Bl < h->sh_Bl ? B - shard_rank * h->sh_Bl : h->sh_Bl;
    }
    h->tile = tile;
    h->N = cfg->n_particles; h->R = cfg->n_filters; h->B = B; h->Npad = B * tile;
    h->Bs = (B + 1) & ~1; h->Bpow2 = next_pow2(B);
    h->rshift = 52 - ceil_log2(h->Npad);
    h->split_l2 = B > kSplitLevel2Above ? 1 : 0;
    // in-kernel level-2 keeps T' and A/A' of all tiles in LDS (possible up to 2048 tiles, whichever policy is the default)
    h->lds_bytes = sizeof(double) * (2 * (size_t)(B > kMaxTilesPerFilter ? 2 : (h->Bpow2 < 2 ? 2 : h->Bpow2)) + (size_t)kStageTiles * tile);
    h->lds_bytes_big = sizeof(double) * (4 + (size_t)kStageTiles * tile);
    h->lds_bytes_plan = sizeof(double) * (size_t)(h->Bpow2 < 2 ? 2 : h->Bpow2);
    h->graph_mode = 1;
    h->small_series = 1;
    dims_of(cfg->model, &h->dx, &h->dy);
    if (h->dx > 1 || h->dy > 1) h->small_series = 0;             // vector models run the tiled step kernel at every N
    h->nt = tile == kTileSmall ? 256 : 512;
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { delete h; return SSME_ERR_HIP; }
    {
        int cus = 0;
        h->num_cus = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) ? cus : 256;
    }
    int rc = [&]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = h->stream;
        HIPCHK(hipEventCreate(&h->ev0));
        HIPCHK(hipEventCreate(&h->ev1));
        const size_t np = (size_t)h->R * h->Npad, nb = (size_t)h->R * h->Bs;
        if (h->shard_world > 0) HIPCHK(hipMalloc(&h->plan_dev, sizeof(int32_t) * 2 * h->shard_world));
        for (int i = 0; i < 2 && h->shard_world == 0; ++i) {   // a sharded handle works on the caller's buffers
            HIPCHK(hipMalloc(&h->x[i], sizeof(double) * np * h->dx));          // dim_x planes of [R][Npad]
            HIPCHK(hipMalloc(&h->cdf[i], sizeof(double) * np));
            HIPCHK(hipMalloc(&h->tsum[i], sizeof(double) * nb));
            HIPCHK(hipMalloc(&h->tmax[i], sizeof(double) * nb));
            HIPCHK(hipMemset(h->x[i], 0, sizeof(double) * np * h->dx));
            HIPCHK(hipMemset(h->cdf[i], 0, sizeof(double) * np));
            HIPCHK(hipMemset(h->tsum[i], 0, sizeof(double) * nb));
            HIPCHK(hipMemset(h->tmax[i], 0, sizeof(double) * nb));
        }
        grant_step_lds(h);
        HIPCHK(hipGetLastError());
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level2_plan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lds_bytes_plan));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_l2_ranges), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lds_bytes_plan));
        HIPCHK(hipMalloc(&h->l2_work, sizeof(double) * ((size_t)h->R * h->Bs + (size_t)h->R * kL2Scratch + 32)));
        HIPCHK(hipMemset(h->l2_work, 0, sizeof(double) * ((size_t)h->R * h->Bs + (size_t)h->R * kL2Scratch + 32)));
        HIPCHK(hipMalloc(&h->l2_T, sizeof(double) * (size_t)h->R * h->Bs));
        HIPCHK(hipMalloc(&h->l2_R, sizeof(double) * (size_t)h->R * h->Bs));
        HIPCHK(hipMalloc(&h->l2_lo, sizeof(int32_t) * (size_t)h->R * h->Bs));
        HIPCHK(hipMalloc(&h->l2_hi, sizeof(int32_t) * (size_t)h->R * h->Bs));
        HIPCHK(hipMemset(h->l2_lo, 0, sizeof(int32_t) * (size_t)h->R * h->Bs));
        HIPCHK(hipMemset(h->l2_hi, 0, sizeof(int32_t) * (size_t)h->R * h->Bs));
        HIPCHK(hipMalloc(&h->scal, sizeof(FilterScalars) * h->R));
        HIPCHK(hipMalloc(&h->mc, sizeof(ModelConst) * h->R));
        HIPCHK(hipMalloc(&h->scratchR, sizeof(double) * h->R));
        HIPCHK(hipMalloc(&h->exp_part, sizeof(double) * (size_t)h->R * h->Bs * kMaxFunctionals));
        HIPCHK(hipMalloc(&h->exp_out, sizeof(double) * ((size_t)(kMaxFunctionals + 1) * h->R + kMaxFunctionals + 1)));
        HIPCHK(hipMalloc(&h->keybuf, sizeof(uint32_t) * 2));
        HIPCHK(hipMalloc(&h->yz_step, sizeof(double) * 2));
        HIPCHK(hipMalloc(&h->ticket, sizeof(int32_t) * h->R));
        HIPCHK(hipMalloc(&h->l2_ticket, sizeof(int32_t) * h->R));
        HIPCHK(hipMemsetAsync(h->l2_ticket, 0, sizeof(int32_t) * h->R, h->stream));
        HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int32_t) * h->R, h->stream));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->pin), sizeof(double) * (2 + (size_t)h->R + 64 + 8), hipHostMallocMapped));   // + 128 ints for the shard plan + 8 swarm aggregates
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_dev), h->pin, 0));
        int rc2 = upload_key(h);
        if (rc2 != SSME_OK) return rc2;
        if (h->shard_world > 0) {
            rc2 = ensure_logw(h);                // resamp_sched > 1: the carried log-weights of this rank's particles (local offsets)
            if (rc2 != SSME_OK) return rc2;
            return ensure_series_capacity(h, 1);
        }
        rc2 = ensure_logw(h);
        if (rc2 != SSME_OK) return rc2;
        return ensure_series_capacity(h, 1);
    }();
    if (rc != SSME_OK) { ssme_pf_destroy(h); return rc; }
    *out = h;
    return SSME_OK;
}

int ssme_pf_destroy(ssme_pf_handle h) {
    if (!h) return SSME_ERR_INVALID_ARG;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    h->stream = h->own_stream;
    if (h->gexec) hipGraphExecDestroy(h->gexec);
    void* bufs[] = {h->x[0], h->x[1], h->cdf[0], h->cdf[1], h->tsum[0], h->tsum[1], h->tmax[0], h->tmax[1], h->logw,
                    h->ybuf, h->zbuf, h->per_step, h->scratchR, h->anc, h->scal, h->mc, h->gam, h->pgam, h->gtot, h->keybuf, h->plan_dev, h->l2_T, h->l2_R, h->l2_work, h->l2_lo, h->l2_hi, h->yz_step, h->ticket, h->l2_ticket, h->small_ms, h->exp_part, h->exp_out, h->wscratch,
                    h->sh_x[0], h->sh_x[1], h->sh_c[0], h->sh_c[1], h->sh_loc, h->sh_raw, h->sh_tsum, h->sh_tmax, h->sh_winx, h->sh_winc, h->sh_flag};
    for (void* p : bufs) if (p) hipFree(p);
    if (h->pin) hipHostFree(h->pin);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return SSME_OK;
}

int ssme_pf_create(const ssme_pf_config* cfg, ssme_pf_handle* out) { return create_impl(cfg, 0, 0, out); }
int ssme_pf_user_model_n_theta(void) {
#if SSME_HAS_USER_MODEL
    return ssme_user_model0::n_theta;
#else
    return 0;
#endif
}
int ssme_pf_user_model_dims(int32_t* dim_x, int32_t* dim_y) {
    if (!dim_x || !dim_y) return SSME_ERR_INVALID_ARG;
#if SSME_HAS_USER_MODEL
    int dx = 1, dy = 1;
    dims_of(SSME_MODEL_USER0, &dx, &dy);
    *dim_x = dx; *dim_y = dy;
    return SSME_OK;
#else
    *dim_x = 0; *dim_y = 0;
    return SSME_ERR_UNSUPPORTED;
#endif
}
int ssme_pf_default_tile(int32_t n_particles, int32_t bank_filters) { return default_tile(n_particles, bank_filters < 1 ? 1 : bank_filters); }

// ---- particle-sharded filter: one filter of cfg->n_particles particles over `world` GPUs ------------------------------
//   This handle owns rank `rank`'s share: B/world consecutive tiles.  Particle buffers
// are the caller's (device pointers), so that the host side can hand the same memory to its collective library.
int ssme_pf_shard_create(const ssme_pf_config* cfg, int32_t rank, int32_t world, ssme_pf_handle* out) {
    if (!cfg || !out) return SSME_ERR_INVALID_ARG;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return SSME_ERR_INVALID_ARG;
    if (cfg->n_filters != 1 || cfg->resamp_sched < 1) return SSME_ERR_UNSUPPORTED;
    if (cfg->n_particles < 1) return SSME_ERR_UNSUPPORTED;
    {
        // ceil(B / world) tiles per rank; the last rank takes what is left (fewer tiles, a ragged last tile) and must own at least one
        const int B = (cfg->n_particles + kTile - 1) / kTile, Bl = (B + world - 1) / world;
        if ((world - 1) * Bl >= B) return SSME_ERR_UNSUPPORTED;
    }
    return create_impl(cfg, rank, world, out);
}

int ssme_pf_shard_layout(ssme_pf_handle h, int32_t* out4) {
    if (!h || !out4) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    const long first = (long)h->shard_rank * h->sh_Bl * kTile, own = (long)h->sh_Bown * kTile;
    out4[0] = h->B; out4[1] = h->sh_Bl; out4[2] = h->sh_Bown; out4[3] = (int32_t)(h->N - first < own ? h->N - first : own);
    return SSME_OK;
}

int ssme_pf_set_stream(ssme_pf_handle h, void* hip_stream) {
    if (!h) return SSME_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    return SSME_OK;
}

int ssme_pf_shard_prepare(ssme_pf_handle h, const double* y, const double* z, int32_t T) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->params_set) return SSME_ERR_STATE;
    if (T < 1) return SSME_ERR_LENGTH;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = ensure_series_capacity(h, T);
    if (rc != SSME_OK) return rc;
    HIPCHK(hipMemcpyAsync(h->ybuf, y, sizeof(double) * T * h->dy, hipMemcpyHostToDevice, h->stream));
    if (z) HIPCHK(hipMemcpyAsync(h->zbuf, z, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    h->g_has_z = z != nullptr;
    rc = do_reset(h);
    if (rc != SSME_OK) return rc;
    launch_gamma(h, 0, T);                       // global tables: every rank holds all B tiles' draws
    HIPCHK(hipGetLastError());
    return SSME_OK;
}

static StepArgs shard_args(ssme_pf_handle h, int t, const double* tsum_all, const double* tmax_all) {
    StepArgs a = step_args(h);
    a.tsum_in = tsum_all; a.tmax_in = tmax_all;
    a.z = h->g_has_z ? h->zbuf : nullptr;
    a.per_step = h->per_step;
    a.t = t; a.yi = t; a.gi = t;
    a.logw = h->cfg.resamp_sched > 1 ? h->logw : nullptr;      // steps without resampling carry the log-weights (this rank's own, local offsets)
    a.anc = nullptr;
    return a;
}

// the plan of step t on the device: every rank's window into h->plan_dev (up to 1024 tiles) or every tile's range into
// l2_lo / l2_hi (split level-2, which also accounts log p(y_{t-1} | .))
static void shard_plan_device(ssme_pf_handle h, int t, const double* tsum_all, const double* tmax_all, int margin = 0,
                              int32_t* flag = nullptr) {
    StepArgs a = shard_args(h, t, tsum_all, tmax_all);
    if (h->split_l2) {
        a.finalize_prev = 1;
        launch_level2(h->stream, a, 1, h->lds_bytes_plan, 1);
        if (flag) hipLaunchKernelGGL(k_shard_window_check, dim3(1), dim3(64), 0, h->stream, (const int32_t*)nullptr, (const int32_t*)h->l2_lo,
                                     (const int32_t*)h->l2_hi, h->shard_world, h->sh_Bl, h->B, margin, flag, flag + 1);
    } else {
        hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(512), sizeof(double) * (h->Bpow2 < 2 ? 2 : h->Bpow2), h->stream, a,
                           h->shard_world, h->sh_Bl, h->plan_dev, margin, flag);
    }
}

int ssme_pf_shard_plan(ssme_pf_handle h, const double* tsum_all, const double* tmax_all, int32_t t, int32_t* lo_hi) {
    if (!h || !tsum_all || !tmax_all || !lo_hi || t < 1) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    shard_plan_device(h, t, tsum_all, tmax_all);
    HIPCHK(hipGetLastError());
    int32_t* stage = reinterpret_cast<int32_t*>(h->pin + 2 + h->R);            // pinned
    if (h->split_l2) {
        // more than 1024 tiles in total: the split level-2 plans every tile; a rank's window is [lo of its first tile, hi of its last]
        const int Bl = h->sh_Bl;
        const bool sorted = h->cfg.resampler != SSME_RESAMP_MULTINOMIAL_IID;
        for (int d = 0; d < h->shard_world; ++d) {
            stage[2 * d] = 0; stage[2 * d + 1] = h->B - 1;
            if (sorted) {
                HIPCHK(hipMemcpyAsync(stage + 2 * d, h->l2_lo + (size_t)d * Bl, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
                const size_t last = (size_t)(d + 1) * Bl - 1 < (size_t)h->B - 1 ? (size_t)(d + 1) * Bl - 1 : (size_t)h->B - 1;     // the last rank may own fewer tiles
                HIPCHK(hipMemcpyAsync(stage + 2 * d + 1, h->l2_hi + last, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            }
        }
    } else {
        HIPCHK(hipMemcpyAsync(stage, h->plan_dev, sizeof(int32_t) * 2 * h->shard_world, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(wait_stream_low_latency(h->stream));
    for (int d = 0; d < 2 * h->shard_world; ++d) lo_hi[d] = stage[d];
    return SSME_OK;
}

int ssme_pf_shard_step(ssme_pf_handle h, int32_t t, const double* x_win, const double* cdf_win, int32_t win_tile0,
                       const double* tsum_all, const double* tmax_all, double* x_out, double* cdf_out, double* tsum_out,
                       double* tmax_out, uint32_t* anc_out) {
    if (!h || !x_out || !cdf_out || !tsum_out || !tmax_out || t < 0) return SSME_ERR_INVALID_ARG;
    // win_tile0 may be negative on the C++ driver's fixed-halo path: rank 0's halo buffer starts `margin` (never-read) rows before tile 0
    // (on a step without resampling -- t % resamp_sched != 0 -- x_win is this rank's OWN particles of step t - 1, stored as the
    //  outputs are: the kernel reads them in place, cdf_win is not read)
    if (t > 0 && (!x_win || !cdf_win || !tsum_all || !tmax_all || win_tile0 < -h->sh_margin)) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->params_set) return SSME_ERR_STATE;
    if (t >= h->tcap) return SSME_ERR_STATE;            // ssme_pf_shard_prepare sizes the series
    HIPCHK(hipSetDevice(h->cfg.device));
    StepArgs a = shard_args(h, t, tsum_all, tmax_all);
    a.x_in = x_win; a.cdf_in = cdf_win; a.win_tile0 = win_tile0;
    if (h->sh_check && (t % h->cfg.resamp_sched) == 0) { a.win_tiles = h->sh_rows; a.win_flag = h->sh_flag; }
    a.x_out = x_out; a.cdf_out = cdf_out; a.tsum_out = tsum_out; a.tmax_out = tmax_out;
    a.anc = anc_out;
    a.finalize_prev = t > 0 ? 1 : 0;
    a.tile0 = h->shard_rank * h->sh_Bl;
    launch_step_on(h, a, dim3(h->sh_Bown, 1));
    HIPCHK(hipGetLastError());
    h->t = t + 1;
    return SSME_OK;
}

int ssme_pf_shard_finalize(ssme_pf_handle h, int32_t t, const double* tsum_all, const double* tmax_all) {
    if (!h || !tsum_all || !tmax_all || t < 0) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    StepArgs a = shard_args(h, t, tsum_all, tmax_all);
    if (h->split_l2) {
        a.t = t + 1; a.finalize_prev = 1;
        launch_level2(h->stream, a, 1, h->lds_bytes_plan, 0);
    } else
        hipLaunchKernelGGL(kf_finalize, dim3(1), dim3(kThreads), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return SSME_OK;
}

// ---- C++ driver of the sharded filter over RCCL (shard_driver.h) ----------------------------------------------------------
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    if (h) h->err = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); return SSME_ERR_HIP; } } while (0)

int ssme_shard_comm_get_unique_id(void* id128) {
    ssme_pf_handle h = nullptr;
    if (!id128) return SSME_ERR_INVALID_ARG;
    if (!rccl().ok) return SSME_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    NCCLCHK(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
    return SSME_OK;
}
int ssme_shard_comm_init(const void* id128, int32_t rank, int32_t world, int32_t device, void** comm_out) {
    ssme_pf_handle h = nullptr;
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return SSME_ERR_INVALID_ARG;
    if (!rccl().ok) return SSME_ERR_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess) return SSME_ERR_HIP;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(rccl().CommInitRank(&c, world, id, rank));
    *comm_out = c;
    return SSME_OK;
}
int ssme_shard_comm_destroy(void* comm) {
    ssme_pf_handle h = nullptr;
    if (!comm) return SSME_ERR_INVALID_ARG;
    if (!rccl().ok) return SSME_ERR_UNSUPPORTED;
    NCCLCHK(rccl().CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
    return SSME_OK;
}

static int shard_alloc(ssme_pf_handle h, bool exact) {
    const int world = h->shard_world, Bl = h->sh_Bl;
    if (!h->sh_x[0]) {
        // halo margin: a rank's resampling window normally reaches a tile or two into its neighbours (the cumulative tile
        // weights wander like sqrt(tiles) around the uniform split); 4 tiles or 1/64 of the share, never more than the share
        int m = Bl / 64 > 4 ? Bl / 64 : 4;
        if (m > Bl) m = Bl;
        if (world == 1) m = 0;
        h->sh_margin = m; h->sh_rows = Bl + 2 * m;
        const size_t nb = sizeof(double) * (size_t)h->sh_rows * kTile;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc(&h->sh_x[i], nb)); HIPCHK(hipMemset(h->sh_x[i], 0, nb));
            HIPCHK(hipMalloc(&h->sh_c[i], nb)); HIPCHK(hipMemset(h->sh_c[i], 0, nb));
        }
        HIPCHK(hipMalloc(&h->sh_loc, sizeof(double) * 2 * Bl));
        const size_t nall = (size_t)world * Bl > (size_t)h->Bs ? (size_t)world * Bl : (size_t)h->Bs;   // gathered: world x Bl entries, the first B are tiles
        HIPCHK(hipMalloc(&h->sh_raw, sizeof(double) * 2 * nall));
        HIPCHK(hipMalloc(&h->sh_tsum, sizeof(double) * nall));
        HIPCHK(hipMalloc(&h->sh_tmax, sizeof(double) * nall));
        HIPCHK(hipMemset(h->sh_loc, 0, sizeof(double) * 2 * Bl));
        HIPCHK(hipMalloc(&h->sh_flag, sizeof(int32_t) * 4));
    }
    if (exact && !h->sh_winx) {
        HIPCHK(hipMalloc(&h->sh_winx, sizeof(double) * (size_t)h->B * kTile));
        HIPCHK(hipMalloc(&h->sh_winc, sizeof(double) * (size_t)h->B * kTile));
    }
    return SSME_OK;
}

// tile sums and tile maxima of all ranks, each straight into its final [B] array: two all-gathers in one group (one launch)
static int shard_gather(ssme_pf_handle h, ncclComm_t comm) {
    const int world = h->shard_world, Bl = h->sh_Bl;
    (void)world;
    NCCLCHK(rccl().GroupStart());
    NCCLCHK(rccl().AllGather(h->sh_loc, h->sh_tsum, (size_t)Bl, ncclDouble, comm, h->stream));
    NCCLCHK(rccl().AllGather(h->sh_loc + Bl, h->sh_tmax, (size_t)Bl, ncclDouble, comm, h->stream));
    NCCLCHK(rccl().GroupEnd());
    return SSME_OK;
}

// One series on one path.  fast: fixed halo exchange, no host synchronisation inside the loop; exact: planned exchange.
static int shard_series(ssme_pf_handle h, ncclComm_t comm, const double* y, const double* z, int T, bool fast, bool* overflow) {
    const int world = h->shard_world, rank = h->shard_rank, Bl = h->sh_Bl, m = h->sh_margin, tile0 = rank * Bl;
    const size_t TL = kTile;
    int rc = ssme_pf_shard_prepare(h, y, z, T);
    if (rc != SSME_OK) return rc;
    HIPCHK(hipMemsetAsync(h->sh_flag, 0, sizeof(int32_t) * 4, h->stream));
    h->sh_exchanged = 0;
    int cur = 0;
    std::vector<int32_t> lo_hi(2 * (size_t)world);
    for (int t = 0; t < T; ++t) {
        double* xs = h->sh_x[cur]; double* cs = h->sh_c[cur];                     // sources: own rows + halos
        double* xo = h->sh_x[cur ^ 1] + (size_t)m * TL; double* co = h->sh_c[cur ^ 1] + (size_t)m * TL;   // outputs: the own rows of the other pair
        const double *xw = nullptr, *cw = nullptr;
        int win0 = 0;
        const bool resampled = t > 0 && (t % h->cfg.resamp_sched) == 0;     // the draw that closes step t - 1 runs now (lazily)
        if (t > 0) {
            rc = shard_gather(h, comm);
            if (rc != SSME_OK) return rc;
            if (!resampled) {
                // no draw: every particle continues itself with its carried log-weight -- sources = this rank's own tiles, no exchange;
                // the gathered tile sums still give every rank the step's log p(y_{t-1} | .)
                if (h->split_l2) { shard_plan_device(h, t, h->sh_tsum, h->sh_tmax); HIPCHK(hipGetLastError()); }
                xw = xs + (size_t)m * TL; cw = cs + (size_t)m * TL; win0 = tile0;
            } else if (fast) {
                // up to 1024 tiles the step kernel runs level-2 itself and checks its own source tiles against the fixed
                // halo (StepArgs::win_flag): no plan launch at all; above, k_level2_plan is needed anyway and a small check follows it
                if (h->split_l2) {
                    shard_plan_device(h, t, h->sh_tsum, h->sh_tmax, m, h->sh_flag);
                    HIPCHK(hipGetLastError());
                }
                if (world > 1) {
                    // fixed halo: my first m tiles are the left neighbour's right halo, my last m tiles the right neighbour's left halo
                    NCCLCHK(rccl().GroupStart());
                    for (double* buf : {xs, cs}) {
                        if (rank > 0) {
                            NCCLCHK(rccl().Send(buf + (size_t)m * TL, (size_t)m * TL, ncclDouble, rank - 1, comm, h->stream));
                            NCCLCHK(rccl().Recv(buf, (size_t)m * TL, ncclDouble, rank - 1, comm, h->stream));
                        }
                        if (rank + 1 < world) {
                            NCCLCHK(rccl().Send(buf + (size_t)Bl * TL, (size_t)m * TL, ncclDouble, rank + 1, comm, h->stream));
                            NCCLCHK(rccl().Recv(buf + (size_t)(m + Bl) * TL, (size_t)m * TL, ncclDouble, rank + 1, comm, h->stream));
                        }
                    }
                    NCCLCHK(rccl().GroupEnd());
                    h->sh_exchanged += (long)m * ((rank > 0) + (rank + 1 < world));
                }
                xw = xs; cw = cs; win0 = tile0 - m;                                   // row 0 of the halo buffer is global tile tile0 - m
            } else {
                rc = ssme_pf_shard_plan(h, h->sh_tsum, h->sh_tmax, t, lo_hi.data());   // synchronises: the exact path is host-planned
                if (rc != SSME_OK) return rc;
                const int lo = lo_hi[2 * rank], hi = lo_hi[2 * rank + 1];
                NCCLCHK(rccl().GroupStart());
                for (int pass = 0; pass < 2; ++pass) {
                    const double* own = (pass == 0 ? xs : cs) + (size_t)m * TL;          // my tiles tile0 .. tile0 + Bl - 1
                    double* win = pass == 0 ? h->sh_winx : h->sh_winc;               // window: global tiles lo .. hi
                    for (int p = 0; p < world; ++p) {
                        // what I need from rank p: [lo, hi] x p's tiles
                        const int a1 = lo > p * Bl ? lo : p * Bl, b1 = hi < (p + 1) * Bl - 1 ? hi : (p + 1) * Bl - 1;
                        if (b1 >= a1) {
                            if (p == rank) HIPCHK(hipMemcpyAsync(win + (size_t)(a1 - lo) * TL, own + (size_t)(a1 - tile0) * TL, sizeof(double) * (size_t)(b1 - a1 + 1) * TL,
                                                                 hipMemcpyDeviceToDevice, h->stream));
                            else { NCCLCHK(rccl().Recv(win + (size_t)(a1 - lo) * TL, (size_t)(b1 - a1 + 1) * TL, ncclDouble, p, comm, h->stream)); if (pass == 0) h->sh_exchanged += b1 - a1 + 1; }
                        }
                        // what rank p needs from me: [lo_p, hi_p] x my tiles
                        if (p != rank) {
                            const int lp = lo_hi[2 * p], hp = lo_hi[2 * p + 1];
                            const int a2 = lp > tile0 ? lp : tile0, b2 = hp < tile0 + Bl - 1 ? hp : tile0 + Bl - 1;
                            if (b2 >= a2) NCCLCHK(rccl().Send(own + (size_t)(a2 - tile0) * TL, (size_t)(b2 - a2 + 1) * TL, ncclDouble, p, comm, h->stream));
                        }
                    }
                }
                NCCLCHK(rccl().GroupEnd());
                xw = h->sh_winx; cw = h->sh_winc; win0 = lo;
            }
        }
        h->sh_check = (fast && t > 0) ? 1 : 0;                                         // the step kernel checks its sources against the halo
        rc = ssme_pf_shard_step(h, t, xw, cw, win0, h->sh_tsum, h->sh_tmax, xo, co, h->sh_loc, h->sh_loc + Bl, nullptr);
        h->sh_check = 0;
        if (rc != SSME_OK) return rc;
        cur ^= 1;
    }
    rc = shard_gather(h, comm);
    if (rc != SSME_OK) return rc;
    rc = ssme_pf_shard_finalize(h, T - 1, h->sh_tsum, h->sh_tmax);                    // synchronises
    if (rc != SSME_OK) return rc;
    h->cur = cur;                                                                     // sh_x[cur] holds the final particles
    // The fallback decision must be the SAME on every rank: a rank's own flag only says what its own workgroups saw (up to
    // 1024 tiles no plan kernel runs), so the flags are reduced over the ranks before anyone reads them.  One int, once per series.
    *overflow = false;
    if (fast) {           // (the exact path has no halo to leave; sh_stats keeps the record of the fixed-halo pass it may be the rerun of)
        NCCLCHK(rccl().AllReduce(h->sh_flag, h->sh_flag + 3, 1, ncclInt32, ncclMax, comm, h->stream));
        HIPCHK(hipMemcpyAsync(h->sh_stats, h->sh_flag, sizeof(h->sh_stats), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        *overflow = h->sh_stats[3] != 0;
    }
    return SSME_OK;
}

int ssme_pf_shard_run_series(ssme_pf_handle h, void* nccl_comm, const double* y, const double* z, int32_t T, int32_t mode, double* loglik_out) {
    if (!h || !nccl_comm || !y || mode < 0 || mode > 2) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->params_set) return SSME_ERR_STATE;
    if (T < 1) return SSME_ERR_LENGTH;
    if (!rccl().ok) { h->err = "RCCL (librccl.so) not found in this process"; return SSME_ERR_UNSUPPORTED; }
    if (h->cfg.resampler == SSME_RESAMP_MULTINOMIAL_IID && mode == 1) return SSME_ERR_UNSUPPORTED;      // unsorted targets need every tile
    HIPCHK(hipSetDevice(h->cfg.device));
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    const bool want_fast = mode != 2 && h->cfg.resampler != SSME_RESAMP_MULTINOMIAL_IID;
    int rc = shard_alloc(h, !want_fast);
    if (rc != SSME_OK) return rc;
    bool overflow = false;
    std::memset(h->sh_stats, 0, sizeof(h->sh_stats));
    h->sh_path = want_fast ? 1 : 2;
    rc = shard_series(h, comm, y, z, T, want_fast, &overflow);
    if (rc != SSME_OK) return rc;
    if (overflow) {
        // a window left the fixed halo on SOME rank (the reduced flag: every rank reads the same value and takes the same branch):
        // mode 1 reports it, mode 0 runs the series again on the exact path
        if (mode == 1) { h->err = "a resampling window left the fixed halo"; return SSME_ERR_STATE; }
        rc = shard_alloc(h, true);
        if (rc != SSME_OK) return rc;
        h->sh_path = 2;
        rc = shard_series(h, comm, y, z, T, false, &overflow);
        if (rc != SSME_OK) return rc;
    }
    if (loglik_out) return ssme_pf_get_loglik(h, loglik_out);
    return SSME_OK;
}

// this rank's particles and integer cdf after the last native series (N / world values each); path / statistics
int ssme_pf_shard_download(ssme_pf_handle h, double* x_local, uint64_t* cdf_local, int32_t* path, int64_t* exchanged_tiles) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->sh_x[0]) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    // this rank's particles: Bl tiles, fewer on the last rank (and its last tile may be ragged)
    const size_t first = (size_t)h->shard_rank * h->sh_Bl * kTile, off = (size_t)h->sh_margin * kTile;
    const size_t n = (size_t)h->N - first < (size_t)h->sh_Bown * kTile ? (size_t)h->N - first : (size_t)h->sh_Bown * kTile;
    if (x_local) HIPCHK(hipMemcpy(x_local, h->sh_x[h->cur] + off, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (cdf_local) {
        HIPCHK(hipMemcpy(cdf_local, h->sh_c[h->cur] + off, sizeof(double) * n, hipMemcpyDeviceToHost));
        double dv;
        for (size_t i = 0; i < n; ++i) { std::memcpy(&dv, &cdf_local[i], 8); cdf_local[i] = (uint64_t)dv; }
    }
    if (path) *path = h->sh_path;
    if (exchanged_tiles) *exchanged_tiles = h->sh_exchanged;
    return SSME_OK;
}

int ssme_pf_shard_stats(ssme_pf_handle h, int32_t* out4) {
    if (!h || !out4) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->sh_x[0]) return SSME_ERR_STATE;
    out4[0] = h->sh_stats[3]; out4[1] = h->sh_stats[0]; out4[2] = h->sh_stats[1]; out4[3] = h->sh_stats[2];
    return SSME_OK;
}

int ssme_pf_set_params(ssme_pf_handle h, const double* theta, int32_t n_theta, int32_t n_rows) {
    if (!h || !theta) return SSME_ERR_INVALID_ARG;
    if (n_theta != n_theta_of(h->cfg.model)) return SSME_ERR_INVALID_ARG;
    if (n_rows != 1 && n_rows != h->R) return SSME_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    h->h_mc.resize(h->R);
    for (int r = 0; r < h->R; ++r) {
        double th[8];
        for (int d = 0; d < n_theta; ++d) {
            const double v = theta[(size_t)(n_rows == 1 ? 0 : r) * n_theta + d];
            th[d] = h->cfg.dtype == SSME_F32 ? f32r(v) : v;
        }
        h->h_mc[r] = derive(h->cfg.model, th);
    }
    HIPCHK(hipMemcpyAsync(h->mc, h->h_mc.data(), sizeof(ModelConst) * h->R, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->params_set = true;
    return do_reset(h);
}

// New random stream for the next evaluation (PMMH draws a fresh likelihood estimate per proposal; the reference
// seeds every model object from the clock).  The key lives in device memory: a captured graph stays valid.
int ssme_pf_set_seed(ssme_pf_handle h, uint64_t seed) {
    if (!h) return SSME_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    h->cfg.seed = seed;
    int rc = upload_key(h);
    if (rc != SSME_OK) return rc;
    return h->params_set ? do_reset(h) : SSME_OK;
}

int ssme_pf_reset(ssme_pf_handle h) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (!h->params_set) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    return do_reset(h);
}

// flags: bit 0 = record ancestor indices, bit 1 = keep the log-weights in memory, bits 2-4 = level-2 policy (below)
int ssme_pf_set_debug(ssme_pf_handle h, int32_t flags) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((flags & 1) && !h->anc) {
        HIPCHK(hipMalloc(&h->anc, sizeof(uint32_t) * (size_t)h->R * h->Npad));
        HIPCHK(hipMemset(h->anc, 0, sizeof(uint32_t) * (size_t)h->R * h->Npad));
    }
    h->debug_anc = (flags & 1) ? 1 : 0;
    h->keep_logw = (flags & 2) ? 1 : 0;
    {
        // bit 2 forces the split level-2, bit 3 the in-kernel one (where it exists: <= 2048 tiles); default by size.
        // bit 4: the split level-2 by the table kernels (two launches above 1024 tiles) instead of one launch + ranges in the step kernel
        const int want = (h->B > kMaxTilesPerFilter || (flags & 4)) ? 1 : ((flags & 8) ? 0 : (h->B > kSplitLevel2Above ? 1 : 0));
        const int tables = (flags & 16) ? 1 : 0;
        if ((want != h->split_l2 || tables != h->l2_tables) && h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        h->split_l2 = want;
        h->l2_tables = tables;
    }
    return ensure_logw(h);
}

int ssme_pf_set_tuning(ssme_pf_handle h, int32_t threads_per_tile) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (threads_per_tile != 256 && threads_per_tile != 512 && threads_per_tile != 1024) return SSME_ERR_INVALID_ARG;
    if (h->tile == kTileSmall && threads_per_tile != 256) return SSME_ERR_UNSUPPORTED;     // 512-particle tiles run 256 threads
    if (h->tile == kTileMid && threads_per_tile != 512) return SSME_ERR_UNSUPPORTED;       // 1024-particle tiles run 512 threads
    h->nt = threads_per_tile;
    return SSME_OK;
}

int ssme_pf_set_small_series(ssme_pf_handle h, int32_t enable) {
    if (!h || enable < 0 || enable > 1) return SSME_ERR_INVALID_ARG;
    h->small_series = (h->dx > 1 || h->dy > 1) ? 0 : enable;      // vector models have no whole-series kernel
    return SSME_OK;
}

int ssme_pf_set_graph_mode(ssme_pf_handle h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) return SSME_ERR_INVALID_ARG;
    h->graph_mode = mode;
    return SSME_OK;
}

int ssme_pf_step(ssme_pf_handle h, const double* y, const double* z, double* out) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles are driven by ssme_pf_shard_*
    if (!h->params_set) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->cfg.resampler == SSME_RESAMP_MULTINOMIAL && h->gcap < kStepGammaChunk) {
        int rc = ensure_gamma_capacity(h, kStepGammaChunk);
        if (rc != SSME_OK) return rc;
    }
    // y and z travel in the kernel arguments; the R results come back through device-mapped pinned memory, written by the
    // last workgroup of the step kernel itself: a filter() call is one launch and a poll (two launches with the split
    // level-2 of very large filters), no copy operation in either direction
    h->pin[0] = *y; h->pin[1] = z ? *z : 0.0;
    for (int d = 1; d < h->dy; ++d) h->y_step_v[d - 1] = y[d];
    if (h->cfg.dtype == SSME_F32) { h->pin[0] = f32r(h->pin[0]); h->pin[1] = f32r(h->pin[1]); }
    // Gamma tables are drawn kStepGammaChunk time steps at a time (data independent), so that the two table launches are paid
    // once per chunk and not once per filter() call
    int gi = 0;
    if (h->cfg.resampler == SSME_RESAMP_MULTINOMIAL) {
        if (h->t < h->gamma_t0 || h->t >= h->gamma_t0 + h->gamma_rows) {
            launch_gamma(h, h->t, kStepGammaChunk);
            h->gamma_t0 = h->t; h->gamma_rows = kStepGammaChunk;
        }
        gi = h->t - h->gamma_t0;
    }
    // out == NULL (the swarm classes: the aggregation that follows hands the data back): the step is only queued -- no result
    // slots, no wait; anything that reads the handle afterwards is ordered behind it on the stream
    const bool want = out != nullptr;
    if (want) mark_results_pending(h->pin + 2, h->R);
    enqueue_step(h, h->t, 0, gi, z != nullptr, /*finalize_prev=*/false, /*per_step=*/false, /*from_step_staging=*/true, want);
    if (h->split_l2) launch_kf(h, h->t, false, want ? h->pin_dev + 2 : nullptr);
    HIPCHK(hipGetLastError());
    h->t += 1;
    if (want) HIPCHK(wait_results(h->stream, h->pin + 2, h->R));
    if (out) for (int r = 0; r < h->R; ++r) out[r] = h->pin[2 + r];
    round_out(h, out, h->R);
    return SSME_OK;
}

static void enqueue_series(ssme_pf_handle h, int T, bool has_z) {
    launch_gamma(h, 0, T);
    for (int t = 0; t < T; ++t) enqueue_step(h, t, t, t, has_z, /*finalize_prev=*/t > 0, /*per_step=*/true);
    launch_kf(h, T - 1, true);
}

int ssme_pf_run_series(ssme_pf_handle h, const double* y, const double* z, int32_t T, double* loglik_out) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles are driven by ssme_pf_shard_*
    if (T < 1) return SSME_ERR_LENGTH;
    if (!h->params_set) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = ensure_series_capacity(h, T);
    if (rc != SSME_OK) return rc;
    std::vector<double> y32, z32;
    if (h->cfg.dtype == SSME_F32) {
        y32.resize(T);
        for (int t = 0; t < T; ++t) y32[t] = f32r(y[t]);
        y = y32.data();
        if (z) { z32.resize(T); for (int t = 0; t < T; ++t) z32[t] = f32r(z[t]); z = z32.data(); }
    }
    HIPCHK(hipMemcpyAsync(h->ybuf, y, sizeof(double) * T * h->dy, hipMemcpyHostToDevice, h->stream));
    if (z) HIPCHK(hipMemcpyAsync(h->zbuf, z, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.dtype == SSME_F32) HIPCHK(hipStreamSynchronize(h->stream));      // the staging vectors go out of scope
    rc = do_reset(h);
    if (rc != SSME_OK) return rc;
    const bool has_z = z != nullptr;
    const int lw = logw_needed(h) ? 1 : 0;
    if (h->B == 1 && h->tile == kTile && h->small_series) {
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        enqueue_series_small(h, T, has_z);
        HIPCHK(hipEventRecord(h->ev1, h->stream));
    } else if (h->graph_mode) {
        if (!h->gexec || h->g_T != T || h->g_has_z != (int)has_z || h->g_debug != h->debug_anc || h->g_logw != lw ||
            h->g_nt != h->nt) {
            if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            h->cur = 0;
            enqueue_series(h, T, has_z);
            HIPCHK(hipStreamEndCapture(h->stream, &g));
            HIPCHK(hipGraphInstantiate(&h->gexec, g, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(g));
            h->g_T = T; h->g_has_z = has_z; h->g_debug = h->debug_anc; h->g_logw = lw; h->g_nt = h->nt;
        }
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        HIPCHK(hipGraphLaunch(h->gexec, h->stream));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        h->cur = (T & 1);
    } else {
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        h->cur = 0;
        enqueue_series(h, T, has_z);
        HIPCHK(hipEventRecord(h->ev1, h->stream));
    }
    HIPCHK(hipGetLastError());
    h->t = T;
    std::vector<FilterScalars> sc(h->R);
    HIPCHK(hipMemcpyAsync(sc.data(), h->scal, sizeof(FilterScalars) * h->R, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if (loglik_out) for (int r = 0; r < h->R; ++r) loglik_out[r] = sc[r].loglik;
    round_out(h, loglik_out, h->R);
    return SSME_OK;
}

int ssme_pf_get_per_step(ssme_pf_handle h, double* out, int32_t T) {
    if (!h || !out || T < 1 || T > h->tcap) return SSME_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    // device layout is [R][tcap]; return [R][T]
    for (int r = 0; r < h->R; ++r)
        HIPCHK(hipMemcpyAsync(out + (size_t)r * T, h->per_step + (size_t)r * h->tcap, sizeof(double) * T,
                              hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    round_out(h, out, (size_t)h->R * T);
    return SSME_OK;
}

int ssme_pf_get_loglik(ssme_pf_handle h, double* out) {
    if (!h || !out) return SSME_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    std::vector<FilterScalars> sc(h->R);
    HIPCHK(hipMemcpyAsync(sc.data(), h->scal, sizeof(FilterScalars) * h->R, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->R; ++r) out[r] = sc[r].loglik;
    round_out(h, out, h->R);
    return SSME_OK;
}

int ssme_pf_log_mean_exp(ssme_pf_handle h, double* out) {
    if (!h || !out) return SSME_ERR_INVALID_ARG;
    std::vector<double> ll(h->R);
    int rc = ssme_pf_get_loglik(h, ll.data());
    if (rc != SSME_OK) return rc;
    // thread_pool.h:263-268, host side (R is small): m + log(sum exp(v - m)) - log R
    double m = ll[0];
    for (double v : ll) if (v > m) m = v;
    double s = 0.0;
    for (double v : ll) s += std::exp(v - m);
    *out = m + std::log(s) - std::log((double)h->R);
    round_out(h, out, 1);
    return SSME_OK;
}

// per-filter expectations of n built-in functionals into exp_out rows 0..n-1 (device); three small launches, no sync
static int enqueue_expectations(ssme_pf_handle h, const int32_t* functionals, int32_t n) {
    if (n < 1 || n > kMaxFunctionals) return SSME_ERR_INVALID_ARG;
    FunctionalIds fs{};
    fs.n = n;
    for (int i = 0; i < n; ++i) {
        if (functionals[i] < 0 || functionals[i] > SSME_H_CONST42) return SSME_ERR_INVALID_ARG;
        fs.id[i] = functionals[i];
    }
    hipLaunchKernelGGL(k_expect_partials, dim3(h->B, h->R), dim3(kThreads), 0, h->stream, h->x[h->cur], h->cdf[h->cur], h->N,
                       h->Npad, h->Bs, h->tile, fs, h->exp_part);
    hipLaunchKernelGGL(k_expect_final, dim3(h->R), dim3(kThreads), 0, h->stream, h->exp_part, h->tsum[h->cur], h->tmax[h->cur],
                       h->B, h->Bs, h->R, fs, h->exp_out);
    HIPCHK(hipGetLastError());
    return SSME_OK;
}

int ssme_pf_get_expectations_multi(ssme_pf_handle h, const int32_t* functionals, int32_t n, double* out) {
    if (!h || !out || !functionals) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (h->t < 1) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = enqueue_expectations(h, functionals, n);
    if (rc != SSME_OK) return rc;
    HIPCHK(hipMemcpyAsync(out, h->exp_out, sizeof(double) * (size_t)n * h->R, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    round_out(h, out, (size_t)n * h->R);
    return SSME_OK;
}

int ssme_pf_get_expectations(ssme_pf_handle h, int32_t functional, double* out) {
    return ssme_pf_get_expectations_multi(h, &functional, 1, out);
}

// Swarm::update's aggregation (pswarm_filter.h:96-160,233-235): the plain means over the R member filters of the last
// step's log conditional likelihoods and of the expectations, reduced ON THE DEVICE; one download of n + 1 doubles.
int ssme_pf_swarm_aggregate(ssme_pf_handle h, const int32_t* functionals, int32_t n, double* mean_logcondlike,
                            double* mean_expectations) {
    return ssme_pf_swarm_aggregate_threads(h, functionals, n, 0, mean_logcondlike, mean_expectations);
}

int ssme_pf_swarm_aggregate_threads(ssme_pf_handle h, const int32_t* functionals, int32_t n, int32_t num_threads, double* mean_logcondlike,
                                    double* mean_expectations) {
    if (!h || !mean_logcondlike || n < 0 || n > kMaxFunctionals || (n > 0 && (!functionals || !mean_expectations)))
        return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (h->t < 1) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (n > 0) {
        int rc = enqueue_expectations(h, functionals, n);
        if (rc != SSME_OK) return rc;
    }
    // one launch writes the n + 1 means straight into mapped host memory, where the host waits for them (see wait_results)
    double* pin = h->pin + 2 + h->R + 64;
    mark_results_pending(pin, n);
    mark_results_pending(pin + kMaxFunctionals, 1);
    hipLaunchKernelGGL(k_swarm_means, dim3(n + 1), dim3(kThreads), 0, h->stream, (const double*)h->exp_out, (const FilterScalars*)h->scal,
                       h->R, n, kMaxFunctionals, h->pin_dev + 2 + h->R + 64, (int)num_threads);
    HIPCHK(hipGetLastError());
    HIPCHK(wait_results(h->stream, pin + kMaxFunctionals, 1));
    if (n > 0) HIPCHK(wait_results(h->stream, pin, n));
    *mean_logcondlike = pin[kMaxFunctionals];
    for (int i = 0; i < n; ++i) mean_expectations[i] = pin[i];
    round_out(h, mean_logcondlike, 1);
    round_out(h, mean_expectations, (size_t)n);
    return SSME_OK;
}

// Particles and normalisable weights of one filter after the last step, for functionals that cannot run on the device
// (arbitrary std::function h, pswarm_filter.h:44): w_j = exp(logw_j - max logw) up to the 2^-41 fixed point -- the same
// weights the device expectations and the resampler use.  No debug mode needed (the weights are rebuilt from the cdf).
int ssme_pf_download_weights(ssme_pf_handle h, int32_t f, double* x, double* w) {
    if (!h || f < 0 || f >= h->R || !w) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (h->t < 1) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (!h->wscratch) HIPCHK(hipMalloc(&h->wscratch, sizeof(double) * (size_t)h->Npad));
    const size_t off = (size_t)f * h->Npad;
    hipLaunchKernelGGL(k_weights, dim3(h->B), dim3(kThreads), 0, h->stream, (const double*)(h->cdf[h->cur] + off),
                       (const double*)(h->tmax[h->cur] + (size_t)f * h->Bs), h->N, h->B, h->tile, h->wscratch);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(w, h->wscratch, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    if (x) for (int d = 0; d < h->dx; ++d)                 // vector models: dim_x planes, x[d * N + i]
        HIPCHK(hipMemcpyAsync(x + (size_t)d * h->N, h->x[h->cur] + (size_t)d * h->R * h->Npad + off, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    round_out(h, x, h->N);
    round_out(h, w, h->N);
    return SSME_OK;
}

int ssme_pf_download_state(ssme_pf_handle h, int32_t f, double* x, double* logw, uint64_t* cdf, uint32_t* anc) {
    if (!h || f < 0 || f >= h->R) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t off = (size_t)f * h->Npad;
    if (x) for (int d = 0; d < h->dx; ++d)                 // vector models: dim_x planes, x[d * N + i]
        HIPCHK(hipMemcpyAsync(x + (size_t)d * h->N, h->x[h->cur] + (size_t)d * h->R * h->Npad + off, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    if (logw) {
        if (!h->logw || !logw_needed(h)) return SSME_ERR_STATE;      // set_debug(2) before stepping
        HIPCHK(hipMemcpyAsync(logw, h->logw + off, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    }
    if (cdf) HIPCHK(hipMemcpyAsync(cdf, h->cdf[h->cur] + off, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    if (anc) {
        if (!h->anc) return SSME_ERR_STATE;
        HIPCHK(hipMemcpyAsync(anc, h->anc + off, sizeof(uint32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (cdf) {   // the device holds the integers in fp64 registers/memory; hand them out as integers
        double dv;
        for (int i = 0; i < h->N; ++i) { std::memcpy(&dv, &cdf[i], 8); cdf[i] = (uint64_t)dv; }
    }
    return SSME_OK;
}

int ssme_pf_download_scalars(ssme_pf_handle h, int32_t f, double* max_logw, uint64_t* sum_q, uint64_t* tile_sums,
                             double* tile_max, int32_t* rshift) {
    if (!h || f < 0 || f >= h->R) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    FilterScalars sc;
    HIPCHK(hipMemcpyAsync(&sc, h->scal + f, sizeof(sc), hipMemcpyDeviceToHost, h->stream));
    if (tile_sums) HIPCHK(hipMemcpyAsync(tile_sums, h->tsum[h->cur] + (size_t)f * h->Bs, sizeof(double) * h->B,
                                         hipMemcpyDeviceToHost, h->stream));
    if (tile_max) HIPCHK(hipMemcpyAsync(tile_max, h->tmax[h->cur] + (size_t)f * h->Bs, sizeof(double) * h->B,
                                        hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (tile_sums) { double dv; for (int i = 0; i < h->B; ++i) { std::memcpy(&dv, &tile_sums[i], 8); tile_sums[i] = (uint64_t)dv; } }
    if (max_logw) *max_logw = sc.m;
    if (sum_q) *sum_q = (sc.S == sc.S && sc.S > 0.0) ? (uint64_t)sc.S : 0;
    if (rshift) *rshift = h->rshift;
    return SSME_OK;
}

int ssme_pf_get_layout(ssme_pf_handle h, int32_t* tile_particles, int32_t* n_tiles) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (tile_particles) *tile_particles = h->tile;
    if (n_tiles) *n_tiles = h->B;
    return SSME_OK;
}

int ssme_pf_last_elapsed_ms(ssme_pf_handle h, float* ms) {
    if (!h || !ms) return SSME_ERR_INVALID_ARG;
    *ms = h->last_ms;
    return SSME_OK;
}

int ssme_pf_profile_series(ssme_pf_handle h, const double* y, const double* z, int32_t T, double* mean_us,
                           int32_t* launches) {
    if (!h || !y || !mean_us || !launches) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (T < 1) return SSME_ERR_LENGTH;
    if (!h->params_set) return SSME_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = ensure_series_capacity(h, T);
    if (rc != SSME_OK) return rc;
    HIPCHK(hipMemcpyAsync(h->ybuf, y, sizeof(double) * T * h->dy, hipMemcpyHostToDevice, h->stream));
    if (z) HIPCHK(hipMemcpyAsync(h->zbuf, z, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    rc = do_reset(h);
    if (rc != SSME_OK) return rc;
    // one event every kGroup launches: an event pair around a single 20-us kernel adds ~3 us of its own
    const int kGroup = 32;
    const int nev = (T + kGroup - 1) / kGroup + 1;
    std::vector<hipEvent_t> ev((size_t)nev);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    const bool has_z = z != nullptr;
    h->cur = 0;
    launch_gamma(h, 0, T);
    HIPCHK(hipEventRecord(ev[0], h->stream));
    for (int t = 0; t < T; ++t) {
        enqueue_step(h, t, t, t, has_z, t > 0, false);
        if ((t + 1) % kGroup == 0 || t + 1 == T) HIPCHK(hipEventRecord(ev[(t + kGroup) / kGroup], h->stream));
    }
    launch_kf(h, T - 1, false);
    HIPCHK(hipStreamSynchronize(h->stream));
    double sa = 0;
    for (int g = 0; g + 1 < nev; ++g) {
        float m1 = 0;
        HIPCHK(hipEventElapsedTime(&m1, ev[g], ev[g + 1]));
        sa += m1;
    }
    for (auto& e : ev) hipEventDestroy(e);
    mean_us[0] = sa * 1000.0 / T;
    launches[0] = T;
    h->t = T;
    return SSME_OK;
}

}  // extern "C"

// ---- device primitives for bit-parity tests ------------------------------------------------
__global__ void k_test_math(int fn, const double* in, double* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double r, s, c;
    switch (fn) {
        case 0: r = dexp(x); break;
        case 1: r = dlog(x); break;
        case 2: dsincos2pi(x, &s, &c); r = s; break;
        case 3: dsincos2pi(x, &s, &c); r = c; break;
        case 4: r = dsqrt(x); break;
        case 6: r = dlog_u(x, kLogTable); break;
        case 7: r = dexp_scaled_t(x, 0, kExpTable); break;
        case 8: dsincos_k24((uint32_t)x, kSinCosTable, &s, &c); r = s; break;
        case 9: dsincos_k24((uint32_t)x, kSinCosTable, &s, &c); r = c; break;
        case 10: r = dlog_u32(x, kLogTable); break;
        case 11: r = dsqrt_pn(x); break;
        default: r = dlog_pn(x); break;
    }
    out[i] = r;
}
__global__ void k_test_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    const u32x4 o = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = o.v0; out[1] = o.v1; out[2] = o.v2; out[3] = o.v3;
}
__global__ void k_test_quantize(const double* in, int shift, u64* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u64)__builtin_rint(dexp_scaled_t(in[i], shift, kExpTable));
}
__global__ void k_test_rescale(const u64* A, const double* dm, int shift, u64* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u64)__builtin_rint((double)A[i] * dexp_scaled_t(dm[i], shift, kExpTable));
}
template <int NT>
__global__ __launch_bounds__(NT) void k_test_block_scan(const u64* in, u64* incl, u64* total) {
    __shared__ double lds_seg[16];
    constexpr int NK = 1024 / NT;
    const int tid = threadIdx.x;
    double q[NK][2], inc[NK][2], tot;
    for (int k = 0; k < NK; ++k) { q[k][0] = (double)in[(k * NT + tid) * 2]; q[k][1] = (double)in[(k * NT + tid) * 2 + 1]; }
    block_scan_f64<NT, 1024 / NT>(q, inc, tot, lds_seg);
    for (int k = 0; k < NK; ++k) { incl[(k * NT + tid) * 2] = (u64)inc[k][0]; incl[(k * NT + tid) * 2 + 1] = (u64)inc[k][1]; }
    if (tid == 0) *total = (u64)tot;
}
// streaming copy with the step kernel's access shape (16 B per lane): calibrates FETCH_SIZE / WRITE_SIZE
__global__ __launch_bounds__(512) void k_calib_copy(const double* in, double* out, long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<double2*>(out)[i] = reinterpret_cast<const double2*>(in)[i];
}
__global__ void k_test_gamma(uint32_t key0, uint32_t key1, uint32_t rep, int t, double shape, int n, double* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = gamma_draw((uint32_t)b, (uint32_t)t, rep, key0, key1, shape);
}

extern "C" {

#define HIPCHK0(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = SSME_ERR_HIP; goto done; } } while (0)

int ssme_pf_test_math(int32_t device, int32_t fn, const double* in, double* out, int64_t n) {
    if (!in || !out || n < 1) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    double *din = nullptr, *dout = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&din, sizeof(double) * n));
    HIPCHK0(hipMalloc(&dout, sizeof(double) * n));
    HIPCHK0(hipMemcpy(din, in, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, din, dout, (long)n);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
done:
    if (din) hipFree(din);
    if (dout) hipFree(dout);
    return rc;
}

int ssme_pf_test_philox(int32_t device, const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
    if (!ctr4 || !key2 || !out4) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    uint32_t* d = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&d, sizeof(uint32_t) * 10));
    HIPCHK0(hipMemcpy(d, ctr4, 16, hipMemcpyHostToDevice));
    HIPCHK0(hipMemcpy(d + 4, key2, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_philox, dim3(1), dim3(1), 0, 0, d, d + 4, d + 6);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(out4, d + 6, 16, hipMemcpyDeviceToHost));
done:
    if (d) hipFree(d);
    return rc;
}

int ssme_pf_test_quantize(int32_t device, const double* in, int32_t shift, uint64_t* out, int64_t n) {
    if (!in || !out || n < 1) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    double* din = nullptr; u64* dout = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&din, sizeof(double) * n));
    HIPCHK0(hipMalloc(&dout, sizeof(u64) * n));
    HIPCHK0(hipMemcpy(din, in, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, din, shift, dout, (long)n);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(out, dout, sizeof(u64) * n, hipMemcpyDeviceToHost));
done:
    if (din) hipFree(din);
    if (dout) hipFree(dout);
    return rc;
}

int ssme_pf_test_rescale(int32_t device, const uint64_t* tile_sums, const double* dm, int32_t shift, uint64_t* out, int64_t n) {
    if (!tile_sums || !dm || !out || n < 1) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    double* ddm = nullptr; u64 *dA = nullptr, *dout = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&ddm, sizeof(double) * n));
    HIPCHK0(hipMalloc(&dA, sizeof(u64) * n));
    HIPCHK0(hipMalloc(&dout, sizeof(u64) * n));
    HIPCHK0(hipMemcpy(ddm, dm, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK0(hipMemcpy(dA, tile_sums, sizeof(u64) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_rescale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dA, ddm, shift, dout, (long)n);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(out, dout, sizeof(u64) * n, hipMemcpyDeviceToHost));
done:
    if (ddm) hipFree(ddm);
    if (dA) hipFree(dA);
    if (dout) hipFree(dout);
    return rc;
}

int ssme_pf_test_block_scan(int32_t device, int32_t threads, const uint64_t* in, uint64_t* incl, uint64_t* total) {
    if (!in || !incl || !total) return SSME_ERR_INVALID_ARG;
    if (threads != 256 && threads != 512 && threads != 1024) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    u64* d = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&d, sizeof(u64) * (2 * kTile + 1)));
    HIPCHK0(hipMemcpy(d, in, sizeof(u64) * kTile, hipMemcpyHostToDevice));
    if (threads == 256) hipLaunchKernelGGL(k_test_block_scan<256>, dim3(1), dim3(256), 0, 0, d, d + kTile, d + 2 * kTile);
    else if (threads == 512) hipLaunchKernelGGL(k_test_block_scan<512>, dim3(1), dim3(512), 0, 0, d, d + kTile, d + 2 * kTile);
    else hipLaunchKernelGGL(k_test_block_scan<1024>, dim3(1), dim3(1024), 0, 0, d, d + kTile, d + 2 * kTile);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(incl, d + kTile, sizeof(u64) * kTile, hipMemcpyDeviceToHost));
    HIPCHK0(hipMemcpy(total, d + 2 * kTile, sizeof(u64), hipMemcpyDeviceToHost));
done:
    if (d) hipFree(d);
    return rc;
}

int ssme_pf_test_copy(int32_t device, int64_t n_doubles, int32_t repeats) {
    if (n_doubles < 2 || repeats < 1) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    double *a = nullptr, *b = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&a, sizeof(double) * n_doubles));
    HIPCHK0(hipMalloc(&b, sizeof(double) * n_doubles));
    HIPCHK0(hipMemset(a, 0, sizeof(double) * n_doubles));
    for (int i = 0; i < repeats; ++i)
        hipLaunchKernelGGL(k_calib_copy, dim3(512), dim3(512), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, (long)(n_doubles / 2));
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipDeviceSynchronize());
done:
    if (a) hipFree(a);
    if (b) hipFree(b);
    return rc;
}

int ssme_pf_test_gamma(int32_t device, uint64_t seed, uint32_t rep, int32_t t, double shape, int32_t n, double* out) {
    if (!out || n < 1) return SSME_ERR_INVALID_ARG;
    int rc = SSME_OK;
    double* d = nullptr;
    HIPCHK0(hipSetDevice(device));
    HIPCHK0(hipMalloc(&d, sizeof(double) * n));
    hipLaunchKernelGGL(k_test_gamma, dim3((n + 255) / 256), dim3(256), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), rep, t,
                       shape, n, d);
    HIPCHK0(hipGetLastError());
    HIPCHK0(hipMemcpy(out, d, sizeof(double) * n, hipMemcpyDeviceToHost));
done:
    if (d) hipFree(d);
    return rc;
}

}  // extern "C"

// =============================================================================================
// Liu-West filter (LWFilterWithCovs, include/ssme/liu_west_filter.h:971-1159)
// =============================================================================================
struct ssme_lw_s {
    ssme_lw_config cfg;
    int N, R, Npad, B, Bs, Bpow2, rshift;
    size_t lds_bytes;
    int t, debug;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    float last_ms;
    double *xB, *thB, *xr, *thr, *lw1, *cdfA, *tsumA, *tmaxA, *cdfB, *tsumB, *tmaxB, *mom, *prop;
    double* momtot;                  // [R][16] moment totals of k_lw_mom_totals (filters of more than 1024 tiles)
    double* lwB;                     // carried second-stage log-weights (resamp_sched > 1 only)
    double* wscratch;                // [5][Npad] weights + untransformed parameters of one filter (host-side functionals)
    int form, rs;                    // 0 auxiliary form / 1 SISR form; resampling schedule m_rs
    double *ybuf, *zbuf, *per_step, *scratch;
    double *gamA, *pgamA, *gtotA, *gamB, *pgamB, *gtotB;
    uint32_t *anc, *kidx, *keybuf;
    uint32_t* ancbuf;        // unsharded handles: this step's resampling ancestors, stage 1 -> stage 2 (compose mode, lw_kernels.h)
    int shard_rank, shard_world;     // particle-sharded filter (world = 0: unsharded)
    int fixed_trans;                 // the transform set is (logit, null, log, twice_fisher): stage kernels with it compiled in (lw_trans_kind)
    int th_plane_tiles;              // sharded: rows (tiles) per theta plane of the caller's OUTPUT buffers (default Bl)
    int sh_Bl, sh_Bown;              // Bl = ceil(B / world) tiles per rank in every layout; the last rank owns B - (world-1) Bl >= 1 of them
    hipStream_t own_stream;
    int32_t* plan_dev;
    int32_t* plan_pin;               // pinned staging of the plan download
    double *pin, *pin_dev;           // step API: device-mapped pinned buffer for the R log conditional likelihoods
    // C++ shard driver (ssme_lw_shard_run_series): halo buffers ([margin | own | margin] rows of 2048 doubles; theta rows of 8192),
    // this rank's stage outputs for the gathers, the gathered arrays, flag
    double *sh_xB, *sh_thB, *sh_cdfB, *sh_xr, *sh_thr, *sh_g1, *sh_cdfA;
    double *sh_locB, *sh_locA, *sh_allB_s, *sh_allB_m, *sh_allA_s, *sh_allA_m, *sh_mom_all;
    int32_t* sh_flag;        // [0] a window left the halo on this rank, [3] max of [0] over all ranks
    int32_t sh_stats[4];
    int sh_margin, sh_rows, sh_check;
    long sh_exchanged;
    int gamma_t0, gamma_rows;        // step API: the Gamma tables hold time indices gamma_t0 .. gamma_t0 + gamma_rows - 1
    int split_l2;                    // level-2 of both draws by the split level-2 kernels (more than 1024 tiles)
    double *l2T[2], *l2R[2];         // [draw: 0 resampling (B), 1 k draw (A)][R][Bs]
    double* l2_work;                 // scratch of the multi-workgroup level-2 (the two draws run one after the other)
    int32_t *l2lo[2], *l2hi[2];
    FilterScalars* l2s[2];
    size_t lds_bytes_big, lds_bytes_plan;
    LwScalars* scal;
    int ycap, tcap, gcap;
    std::string err;
};

static int lw_fail(ssme_lw_handle h, const char* what, hipError_t e) {
    if (h) h->err = std::string(what) + ": " + hipGetErrorString(e);
    return SSME_ERR_HIP;
}
#define LWCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return lw_fail(h, #call, e_); } while (0)

static LwArgs lw_args(ssme_lw_handle h) {
    LwArgs a{};
    a.xB = h->xB; a.thB = h->thB; a.xr = h->xr; a.thr = h->thr; a.lw1 = h->lw1;
    a.lwB = h->rs > 1 ? h->lwB : nullptr; a.form = h->form; a.resamp_sched = h->rs;
    a.cdfA = h->cdfA; a.tsumA = h->tsumA; a.tmaxA = h->tmaxA; a.cdfB = h->cdfB; a.tsumB = h->tsumB; a.tmaxB = h->tmaxB;
    a.mom = h->mom; a.prop = h->prop;
    a.anc = (h->debug & 1) ? h->anc : nullptr; a.kidx = (h->debug & 1) ? h->kidx : nullptr;
    a.ancbuf = h->ancbuf; a.compose = (h->shard_world == 0 && h->ancbuf) ? 1 : 0;
    a.scal = h->scal; a.y = h->ybuf; a.z = h->zbuf; a.per_step = nullptr;
    a.gamA = h->gamA; a.pgamA = h->pgamA; a.gtotA = h->gtotA; a.gamB = h->gamB; a.pgamB = h->pgamB; a.gtotB = h->gtotB;
    a.N = h->N; a.Npad = h->Npad; a.B = h->B; a.Bs = h->Bs; a.Bpow2 = h->Bpow2; a.rshift = h->rshift; a.R = h->R;
    a.Tcap = h->tcap;
    a.key0 = (uint32_t)h->cfg.seed; a.key1 = (uint32_t)(h->cfg.seed >> 32); a.first_filter = h->cfg.first_filter_id;
    a.logN = dlog((double)h->N);
    a.a_shrink = (3.0 * h->cfg.delta - 1.0) / (2.0 * h->cfg.delta);          // liu_west_filter.h:960
    a.tile0 = 0; a.win_tile0 = 0;
    a.l2B_T = h->l2T[0]; a.l2B_R = h->l2R[0]; a.l2B_lo = h->l2lo[0]; a.l2B_hi = h->l2hi[0]; a.l2B_s = h->l2s[0];
    a.l2A_T = h->l2T[1]; a.l2A_R = h->l2R[1]; a.l2A_lo = h->l2lo[1]; a.l2A_hi = h->l2hi[1]; a.l2A_s = h->l2s[1];
    for (int d = 0; d < kDP; ++d) { a.trans[d] = h->cfg.transforms[d]; a.lo[d] = h->cfg.prior_lo[d]; a.hi[d] = h->cfg.prior_hi[d]; }
    return a;
}

static int lw_ensure_gamma(ssme_lw_handle h, int T) {
    if (T > h->gcap) {
        double** tabs[] = {&h->gamA, &h->pgamA, &h->gtotA, &h->gamB, &h->pgamB, &h->gtotB};
        for (auto t : tabs) { if (*t) hipFree(*t); *t = nullptr; }
        const size_t nb = sizeof(double) * (size_t)T * h->R * h->B, nr = sizeof(double) * (size_t)T * h->R;
        LWCHK(hipMalloc(&h->gamA, nb)); LWCHK(hipMalloc(&h->pgamA, nb)); LWCHK(hipMalloc(&h->gtotA, nr));
        LWCHK(hipMalloc(&h->gamB, nb)); LWCHK(hipMalloc(&h->pgamB, nb)); LWCHK(hipMalloc(&h->gtotB, nr));
        h->gcap = T;
    }
    return SSME_OK;
}

static int lw_ensure_capacity(ssme_lw_handle h, int T) {
    if (T > h->ycap) {
        if (h->ybuf) hipFree(h->ybuf);
        if (h->zbuf) hipFree(h->zbuf);
        LWCHK(hipMalloc(&h->ybuf, sizeof(double) * T));
        LWCHK(hipMalloc(&h->zbuf, sizeof(double) * T));
        LWCHK(hipMemset(h->zbuf, 0, sizeof(double) * T));
        h->ycap = T;
    }
    { int rcg = lw_ensure_gamma(h, T); if (rcg != SSME_OK) return rcg; }
    if (T > h->tcap) {
        if (h->per_step) hipFree(h->per_step);
        LWCHK(hipMalloc(&h->per_step, sizeof(double) * (size_t)T * h->R));
        h->tcap = T;
    }
    return SSME_OK;
}

// Gamma tables of both draws for time indices t0 .. t0+nT-1 into rows 0 .. nT-1
static void lw_launch_gamma(ssme_lw_handle h, int t0, int nT) {
    const uint32_t* kp = h->keybuf;
    const dim3 g1((h->B + kThreads - 1) / kThreads, nT, h->R);
    const bool rows = h->B <= 64;                    // short rows: one thread per row; else one workgroup per row
    const dim3 g2(rows ? (nT * h->R + kThreads - 1) / kThreads : nT * h->R);
    auto prefix = [&](double* gam, double* pgam, double* gtot, uint32_t extra) {
        if (rows) hipLaunchKernelGGL(k_gamma_prefix_rows, g2, dim3(kThreads), 0, h->stream, gam, pgam, gtot, h->B, h->R, nT, t0, kp,
                                     h->cfg.first_filter_id, extra);
        else hipLaunchKernelGGL(k_gamma_prefix, g2, dim3(kThreads), 0, h->stream, gam, pgam, gtot, h->B, h->R, nT, t0, kp,
                                h->cfg.first_filter_id, extra);
    };
    hipLaunchKernelGGL(k_gamma_draw, g1, dim3(kThreads), 0, h->stream, h->gamB, h->N, h->B, h->R, t0, kp,
                       h->cfg.first_filter_id, (uint32_t)STREAM_GAMMA, kTile);
    prefix(h->gamB, h->pgamB, h->gtotB, (uint32_t)STREAM_RESAMP_EXTRA);
    hipLaunchKernelGGL(k_gamma_draw, g1, dim3(kThreads), 0, h->stream, h->gamA, h->N, h->B, h->R, t0, kp,
                       h->cfg.first_filter_id, (uint32_t)STREAM_GAMMA_K, kTile);
    prefix(h->gamA, h->pgamA, h->gtotA, (uint32_t)STREAM_LW_K_EXTRA);
}

// the stage kernels, with the reference test models' transform set compiled in when the handle has it
template <bool BIG>
static void lw_launch_stage1(ssme_lw_handle h, dim3 grid, size_t lds, const LwArgs& a) {
    if (h->fixed_trans) hipLaunchKernelGGL((k_lw_stage1<BIG, true>), grid, dim3(kLwNT), lds, h->stream, a);
    else hipLaunchKernelGGL((k_lw_stage1<BIG, false>), grid, dim3(kLwNT), lds, h->stream, a);
}
template <bool BIG>
static void lw_launch_stage2(ssme_lw_handle h, dim3 grid, size_t lds, const LwArgs& a) {
    if (h->fixed_trans) hipLaunchKernelGGL((k_lw_stage2<BIG, true>), grid, dim3(kLwNT), lds, h->stream, a);
    else hipLaunchKernelGGL((k_lw_stage2<BIG, false>), grid, dim3(kLwNT), lds, h->stream, a);
}

static void lw_launch_plan(ssme_lw_handle h, int draw, int t, int gi, const double* tsum, const double* tmax, bool ranges) {
    StepArgs a{};
    a.tsum_in = tsum; a.tmax_in = tmax;
    a.l2_T = h->l2T[draw]; a.l2_R = h->l2R[draw]; a.l2_lo = h->l2lo[draw]; a.l2_hi = h->l2hi[draw];
    a.scal = h->l2s[draw];
    a.B = h->B; a.Bs = h->Bs; a.Bpow2 = h->Bpow2; a.rshift = h->rshift; a.R = h->R; a.N = h->N; a.tile = kTile;
    a.resampler = RESAMP_MULTINOMIAL; a.resamp_sched = 1;
    a.t = t; a.gi = gi; a.finalize_prev = 0;
    a.pgam = draw ? h->pgamA : h->pgamB; a.gtot = draw ? h->gtotA : h->gtotB;
    a.keyp = h->keybuf; a.first_filter = h->cfg.first_filter_id;
    a.l2_work = h->l2_work;
    launch_level2(h->stream, a, h->shard_world > 0 ? 1 : h->R, h->lds_bytes_plan, ranges ? 1 : 0);
}

static void lw_enqueue_step(ssme_lw_handle h, int t, int yi, int gi, bool record, bool finalize_prev, const double* yz_now = nullptr) {
    LwArgs a = lw_args(h);
    if (yz_now) { a.by_value = 1; a.y_now = yz_now[0]; a.z_now = yz_now[1]; }
    a.t = t; a.yi = yi; a.gi = gi; a.finalize_prev = finalize_prev ? 1 : 0;
    a.per_step = record ? h->per_step : nullptr;
    const dim3 grid(h->B, h->R);
    const bool resampled = (t % h->rs) == 0;           // a resampling draw closes step t-1 (lazily: it runs at the start of step t)
    if (t == 0) {
        hipLaunchKernelGGL(k_lw_init, grid, dim3(kLwNT), 0, h->stream, a);
    } else if (h->split_l2) {
        lw_launch_plan(h, 0, t, gi, h->tsumB, h->tmaxB, resampled);
        lw_launch_stage1<true>(h, grid, h->lds_bytes_big, a);
        if (h->form == 0) lw_launch_plan(h, 1, t, gi, h->tsumA, h->tmaxA, true);
        a.momtot = h->momtot;
        hipLaunchKernelGGL(k_lw_mom_totals, dim3(kNMom, h->R), dim3(64), 0, h->stream, a);
        hipLaunchKernelGGL(k_lw_mid<true>, dim3(h->R), dim3(kThreads), 0, h->stream, a);
        lw_launch_stage2<true>(h, grid, h->lds_bytes_big, a);
        if (a.compose) { std::swap(h->xB, h->xr); std::swap(h->thB, h->thr); }      // the new population is where stage 2 wrote it
    } else {
        // two launches when the tile partials ([B][14] doubles) fit the window area of stage 2's LDS: every workgroup of stage 2
        // then takes theta-bar and the Cholesky factor from them itself
        a.fuse_mid = ((size_t)h->B * kNMom * sizeof(double) <= h->lds_bytes) ? 1 : 0;
        lw_launch_stage1<false>(h, grid, h->lds_bytes, a);
        if (!a.fuse_mid) {
            a.momtot = h->momtot;                // 586 .. 1024 tiles: the totals by one wave per moment, then the one-workgroup rest
            hipLaunchKernelGGL(k_lw_mom_totals, dim3(kNMom, h->R), dim3(64), 0, h->stream, a);
            hipLaunchKernelGGL(k_lw_mid<false>, dim3(h->R), dim3(kThreads), 0, h->stream, a);
        }
        lw_launch_stage2<false>(h, grid, h->lds_bytes, a);
        if (a.compose) { std::swap(h->xB, h->xr); std::swap(h->thB, h->thr); }
    }
}
static void lw_enqueue_finalize(ssme_lw_handle h, int t, bool record, double* ll_host = nullptr) {
    LwArgs a = lw_args(h);
    a.t = t; a.ll_host = ll_host;
    a.per_step = record ? h->per_step : nullptr;
    if (h->split_l2) {
        lw_launch_plan(h, 0, t + 1, 0, h->tsumB, h->tmaxB, false);
        hipLaunchKernelGGL(k_lw_finalize<true>, dim3(h->R), dim3(kThreads), 0, h->stream, a);
    } else
        hipLaunchKernelGGL(k_lw_finalize<false>, dim3(h->R), dim3(kThreads), 0, h->stream, a);
}

static int lw_reset(ssme_lw_handle h) {
    std::vector<LwScalars> sc(h->R);
    const double logN = dlog((double)h->N);
    for (auto& s : sc) { std::memset(&s, 0, sizeof(s)); s.prev = logN; }
    LWCHK(hipMemcpyAsync(h->scal, sc.data(), sizeof(LwScalars) * h->R, hipMemcpyHostToDevice, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    h->t = 0;
    h->gamma_rows = 0;
    return SSME_OK;
}

extern "C" {

int ssme_lw_destroy(ssme_lw_handle h) {
    if (!h) return SSME_ERR_INVALID_ARG;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    h->stream = h->own_stream;
    void* bufs[] = {h->xB, h->thB, h->xr, h->thr, h->lw1, h->cdfA, h->tsumA, h->tmaxA, h->cdfB, h->tsumB, h->tmaxB, h->mom, h->prop, h->momtot,
                    h->ybuf, h->zbuf, h->per_step, h->scratch, h->gamA, h->pgamA, h->gtotA, h->gamB, h->pgamB, h->gtotB, h->anc, h->ancbuf,
                    h->kidx, h->scal, h->keybuf, h->plan_dev, h->l2_work, h->l2T[0], h->l2T[1], h->l2R[0], h->l2R[1], h->l2lo[0], h->l2lo[1],
                    h->l2hi[0], h->l2hi[1], h->l2s[0], h->l2s[1], h->lwB, h->wscratch,
                    h->sh_xB, h->sh_thB, h->sh_cdfB, h->sh_xr, h->sh_thr, h->sh_g1, h->sh_cdfA, h->sh_locB, h->sh_locA, h->sh_allB_s, h->sh_allB_m,
                    h->sh_allA_s, h->sh_allA_m, h->sh_mom_all, h->sh_flag};
    for (void* p : bufs) if (p) hipFree(p);
    if (h->plan_pin) hipHostFree(h->plan_pin);
    if (h->pin) hipHostFree(h->pin);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return SSME_OK;
}

static int lw_create_impl(const ssme_lw_config* cfg, int shard_rank, int shard_world, ssme_lw_handle* out) {
    if (!cfg || !out) return SSME_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->n_particles < 1 || cfg->n_filters < 1 || cfg->n_filters > 65535) return SSME_ERR_INVALID_ARG;
    if (!(cfg->delta > 0.0 && cfg->delta <= 1.0)) return SSME_ERR_INVALID_ARG;
    if (cfg->form < 0 || cfg->form > 1 || cfg->resamp_sched < 0) return SSME_ERR_INVALID_ARG;
    for (int d = 0; d < kDP; ++d) {
        if (cfg->transforms[d] < 0 || cfg->transforms[d] > 3) return SSME_ERR_INVALID_ARG;     // parameters.h:283 invalid_argument
        if (!(cfg->prior_lo[d] <= cfg->prior_hi[d])) return SSME_ERR_INVALID_ARG;
    }
    const int B = (cfg->n_particles + kTile - 1) / kTile;
    if (B > kMaxTilesSplit) return SSME_ERR_UNSUPPORTED;          // N <= 2^25
    ssme_lw_handle h = new (std::nothrow) ssme_lw_s();
    if (!h) return SSME_ERR_INVALID_ARG;
    h->cfg = *cfg;
    h->form = cfg->form; h->rs = cfg->resamp_sched < 1 ? 1 : cfg->resamp_sched;
    h->shard_rank = shard_rank; h->shard_world = shard_world;
    h->fixed_trans = (cfg->transforms[0] == TR_LOGIT && cfg->transforms[1] == TR_NULL && cfg->transforms[2] == TR_LOG && cfg->transforms[3] == TR_TWICE_FISHER) ? 1 : 0;
    if (shard_world > 0) {
        h->sh_Bl = (B + shard_world - 1) / shard_world;
        h->sh_Bown = B - shard_rank * h->sh_Bl < h->sh_Bl ? B - shard_rank * h->sh_Bl : h->sh_Bl;
    }
    h->N = cfg->n_particles; h->R = cfg->n_filters; h->B = B; h->Npad = B * kTile;
    h->Bs = (B + 1) & ~1; h->Bpow2 = next_pow2(B);
    h->rshift = 52 - ceil_log2(h->Npad);
    h->split_l2 = B > kSplitLevel2Above ? 1 : 0;
    h->lds_bytes = sizeof(double) * (2 * (size_t)(B > kMaxTilesPerFilter ? 2 : (h->Bpow2 < 2 ? 2 : h->Bpow2)) + (size_t)kStageTiles * kTile);
    h->lds_bytes_big = sizeof(double) * (4 + (size_t)kStageTiles * kTile);
    h->lds_bytes_plan = sizeof(double) * (size_t)(h->Bpow2 < 2 ? 2 : h->Bpow2);
    if (hipSetDevice(cfg->device) != hipSuccess) { delete h; return SSME_ERR_HIP; }
    int rc = [&]() -> int {
        LWCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = h->stream;
        LWCHK(hipMalloc(&h->l2_work, sizeof(double) * ((size_t)h->R * h->Bs + (size_t)h->R * kL2Scratch)));
        for (int d = 0; d < 2; ++d) {
            LWCHK(hipMalloc(&h->l2T[d], sizeof(double) * (size_t)h->R * h->Bs));
            LWCHK(hipMalloc(&h->l2R[d], sizeof(double) * (size_t)h->R * h->Bs));
            LWCHK(hipMalloc(&h->l2lo[d], sizeof(int32_t) * (size_t)h->R * h->Bs));
            LWCHK(hipMalloc(&h->l2hi[d], sizeof(int32_t) * (size_t)h->R * h->Bs));
            LWCHK(hipMalloc(&h->l2s[d], sizeof(FilterScalars) * (size_t)h->R));
            LWCHK(hipMemset(h->l2lo[d], 0, sizeof(int32_t) * (size_t)h->R * h->Bs));
            LWCHK(hipMemset(h->l2hi[d], 0, sizeof(int32_t) * (size_t)h->R * h->Bs));
            LWCHK(hipMemset(h->l2s[d], 0, sizeof(FilterScalars) * (size_t)h->R));
        }
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level2_plan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)h->lds_bytes_plan));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_l2_ranges), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)h->lds_bytes_plan));
        LWCHK(hipEventCreate(&h->ev0));
        LWCHK(hipEventCreate(&h->ev1));
        const size_t np = (size_t)h->R * h->Npad, nb = (size_t)h->R * h->Bs;
        if (h->shard_world == 0) {               // a sharded handle works on the caller's buffers
            double** big[] = {&h->xB, &h->xr, &h->lw1, &h->cdfA, &h->cdfB};
            for (auto p : big) { LWCHK(hipMalloc(p, sizeof(double) * np)); LWCHK(hipMemset(*p, 0, sizeof(double) * np)); }
            if (h->rs > 1) { LWCHK(hipMalloc(&h->lwB, sizeof(double) * np)); LWCHK(hipMemset(h->lwB, 0, sizeof(double) * np)); }
            double** big4[] = {&h->thB, &h->thr};
            for (auto p : big4) { LWCHK(hipMalloc(p, sizeof(double) * np * kDP)); LWCHK(hipMemset(*p, 0, sizeof(double) * np * kDP)); }
            double** small[] = {&h->tsumA, &h->tmaxA, &h->tsumB, &h->tmaxB};
            for (auto p : small) { LWCHK(hipMalloc(p, sizeof(double) * nb)); LWCHK(hipMemset(*p, 0, sizeof(double) * nb)); }
            LWCHK(hipMalloc(&h->ancbuf, sizeof(uint32_t) * np));
            LWCHK(hipMemset(h->ancbuf, 0, sizeof(uint32_t) * np));
        } else {
            LWCHK(hipMalloc(&h->plan_dev, sizeof(int32_t) * 2 * h->shard_world));
            LWCHK(hipHostMalloc(reinterpret_cast<void**>(&h->plan_pin), sizeof(int32_t) * 2 * h->shard_world, hipHostMallocDefault));
            h->th_plane_tiles = h->sh_Bl;
            if (h->rs > 1) {                     // the carried second-stage log-weights of this rank's own particles (local offsets)
                LWCHK(hipMalloc(&h->lwB, sizeof(double) * (size_t)h->sh_Bl * kTile));
                LWCHK(hipMemset(h->lwB, 0, sizeof(double) * (size_t)h->sh_Bl * kTile));
            }
        }
        LWCHK(hipMalloc(&h->mom, sizeof(double) * (size_t)h->R * h->B * 16));
        LWCHK(hipMalloc(&h->prop, sizeof(double) * (size_t)h->R * 16));
        LWCHK(hipMalloc(&h->momtot, sizeof(double) * (size_t)h->R * 16));
        LWCHK(hipMemset(h->prop, 0, sizeof(double) * (size_t)h->R * 16));
        LWCHK(hipMalloc(&h->scal, sizeof(LwScalars) * h->R));
        LWCHK(hipMalloc(&h->scratch, sizeof(double) * h->R * kLwNExp));
        LWCHK(hipHostMalloc(reinterpret_cast<void**>(&h->pin), sizeof(double) * (size_t)h->R, hipHostMallocMapped));
        LWCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_dev), h->pin, 0));
        LWCHK(hipMalloc(&h->keybuf, sizeof(uint32_t) * 2));
        {
            const uint32_t k[2] = {(uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32)};
            LWCHK(hipMemcpy(h->keybuf, k, sizeof(k), hipMemcpyHostToDevice));
        }
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage1<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage2<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage1<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_big));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage2<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_big));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage1<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage2<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage1<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_big));
        LWCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_stage2<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_big));
        int rc2 = lw_ensure_capacity(h, 1);
        if (rc2 != SSME_OK) return rc2;
        return lw_reset(h);
    }();
    if (rc != SSME_OK) { ssme_lw_destroy(h); return rc; }
    *out = h;
    return SSME_OK;
}

int ssme_lw_create(const ssme_lw_config* cfg, ssme_lw_handle* out) { return lw_create_impl(cfg, 0, 0, out); }

// ---- particle-sharded Liu-West filter: one filter over `world` GPUs (BASELINE.json configs[4]) ------------------------
// Same scheme as ssme_pf_shard_* with two exchanges per step: the resampling draw reads windows of (cdfB, x, theta) and the
// k draw windows of (cdfA, lw1, x, theta); the 14 moment partials of every tile are gathered like the tile sums and summed
// in tile order by every rank (k_lw_mid), so theta-bar, the Cholesky factor and all draws equal the unsharded filter's.
int ssme_lw_shard_create(const ssme_lw_config* cfg, int32_t rank, int32_t world, ssme_lw_handle* out) {
    if (!cfg || !out) return SSME_ERR_INVALID_ARG;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return SSME_ERR_INVALID_ARG;
    if (cfg->n_filters != 1) return SSME_ERR_UNSUPPORTED;
    if (cfg->n_particles < 1) return SSME_ERR_UNSUPPORTED;
    {
        // ceil(B / world) tiles per rank; the last rank takes what is left and must own at least one tile (ssme_pf_shard_create)
        const int B = (cfg->n_particles + kTile - 1) / kTile, Bl = (B + world - 1) / world;
        if ((world - 1) * Bl >= B) return SSME_ERR_UNSUPPORTED;
    }
    return lw_create_impl(cfg, rank, world, out);
}

int ssme_lw_shard_layout(ssme_lw_handle h, int32_t* out4) {
    if (!h || !out4) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    const long first = (long)h->shard_rank * h->sh_Bl * kTile, own = (long)h->sh_Bown * kTile;
    out4[0] = h->B; out4[1] = h->sh_Bl; out4[2] = h->sh_Bown; out4[3] = (int32_t)(h->N - first < own ? h->N - first : own);
    return SSME_OK;
}

int ssme_lw_set_stream(ssme_lw_handle h, void* hip_stream) {
    if (!h) return SSME_ERR_INVALID_ARG;
    LWCHK(hipSetDevice(h->cfg.device));
    LWCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    return SSME_OK;
}

int ssme_lw_shard_set_plane_tiles(ssme_lw_handle h, int32_t tiles) {
    if (!h || tiles < 1) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    if (tiles < h->sh_Bl) return SSME_ERR_INVALID_ARG;
    h->th_plane_tiles = tiles;
    return SSME_OK;
}

int ssme_lw_shard_prepare(ssme_lw_handle h, const double* y, const double* z, int32_t T) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    if (T < 1) return SSME_ERR_LENGTH;
    LWCHK(hipSetDevice(h->cfg.device));
    int rc = lw_ensure_capacity(h, T);
    if (rc != SSME_OK) return rc;
    LWCHK(hipMemcpyAsync(h->ybuf, y, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    if (z) LWCHK(hipMemcpyAsync(h->zbuf, z, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    else LWCHK(hipMemsetAsync(h->zbuf, 0, sizeof(double) * T, h->stream));
    rc = lw_reset(h);
    if (rc != SSME_OK) return rc;
    lw_launch_gamma(h, 0, T);
    LWCHK(hipGetLastError());
    return SSME_OK;
}

static LwArgs lw_shard_args(ssme_lw_handle h, int t) {
    LwArgs a = lw_args(h);
    const int Bl = h->sh_Bl;
    a.t = t; a.yi = t; a.gi = t; a.finalize_prev = t > 1 || t == 1 ? 1 : 0;
    a.per_step = h->per_step;
    a.tile0 = h->shard_rank * Bl;
    a.anc = nullptr; a.kidx = nullptr;
    if (h->sh_check) { a.win_tiles = h->sh_rows; a.win_flag = h->sh_flag; }
    return a;
}

int ssme_lw_shard_init(ssme_lw_handle h, double* xB, double* thB, double* cdfB, double* tsumB, double* tmaxB) {
    if (!h || !xB || !thB || !cdfB || !tsumB || !tmaxB) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    LwArgs a = lw_shard_args(h, 0);
    a.xB = xB; a.thB = thB; a.cdfB = cdfB; a.tsumB = tsumB; a.tmaxB = tmaxB;
    hipLaunchKernelGGL(k_lw_init, dim3(h->sh_Bown, 1), dim3(kLwNT), 0, h->stream, a);
    LWCHK(hipGetLastError());
    h->t = 1;
    return SSME_OK;
}

// which = 0: the resampling draw of stage 1 (second-stage weights, Gamma tables B); 1: the k draw of stage 2 (first-stage weights)
int ssme_lw_shard_plan(ssme_lw_handle h, int32_t which, int32_t t, const double* tsum_all, const double* tmax_all, int32_t* lo_hi) {
    if (!h || !tsum_all || !tmax_all || !lo_hi || t < 1 || which < 0 || which > 1) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    if (h->split_l2) {
        lw_launch_plan(h, which, t, t, tsum_all, tmax_all, true);
        LWCHK(hipGetLastError());
        const int Bl = h->sh_Bl;
        for (int d = 0; d < h->shard_world; ++d) {
            const size_t last = (size_t)(d + 1) * Bl - 1 < (size_t)h->B - 1 ? (size_t)(d + 1) * Bl - 1 : (size_t)h->B - 1;
            LWCHK(hipMemcpyAsync(h->plan_pin + 2 * d, h->l2lo[which] + (size_t)d * Bl, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            LWCHK(hipMemcpyAsync(h->plan_pin + 2 * d + 1, h->l2hi[which] + last, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        }
        LWCHK(wait_stream_low_latency(h->stream));
        for (int d = 0; d < 2 * h->shard_world; ++d) lo_hi[d] = h->plan_pin[d];
        return SSME_OK;
    }
    StepArgs a{};                                   // the fields k_shard_plan reads
    a.tsum_in = tsum_all; a.tmax_in = tmax_all;
    a.B = h->B; a.Bs = h->Bs; a.Bpow2 = h->Bpow2; a.rshift = h->rshift; a.R = 1; a.N = h->N; a.tile = kTile;
    a.resampler = RESAMP_MULTINOMIAL; a.resamp_sched = 1;
    a.t = t; a.gi = t;
    a.pgam = which ? h->pgamA : h->pgamB; a.gtot = which ? h->gtotA : h->gtotB;
    a.keyp = h->keybuf; a.first_filter = h->cfg.first_filter_id;
    hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(512), sizeof(double) * (h->Bpow2 < 2 ? 2 : h->Bpow2), h->stream, a,
                       h->shard_world, h->sh_Bl, h->plan_dev);
    LWCHK(hipGetLastError());
    LWCHK(hipMemcpyAsync(h->plan_pin, h->plan_dev, sizeof(int32_t) * 2 * h->shard_world, hipMemcpyDeviceToHost, h->stream));
    LWCHK(wait_stream_low_latency(h->stream));
    for (int d = 0; d < 2 * h->shard_world; ++d) lo_hi[d] = h->plan_pin[d];
    return SSME_OK;
}

int ssme_lw_shard_stage1(ssme_lw_handle h, int32_t t, int32_t win_tile0, int32_t win_tiles, const double* w_xB, const double* w_thB,
                         const double* w_cdfB, const double* tsumB_all, const double* tmaxB_all, double* xr, double* thr, double* lw1,
                         double* cdfA, double* tsumA, double* tmaxA, double* mom, uint32_t* anc) {
    if (!h || t < 1 || win_tile0 < -h->sh_margin || win_tiles < 1 || !w_xB || !w_thB || !w_cdfB || !tsumB_all || !tmaxB_all || !xr || !thr || !lw1 ||
        !cdfA || !tsumA || !tmaxA || !mom)
        return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    if (t >= h->tcap) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    LwArgs a = lw_shard_args(h, t);
    a.xB = const_cast<double*>(w_xB); a.thB = const_cast<double*>(w_thB); a.cdfB = const_cast<double*>(w_cdfB);
    a.tsumB = const_cast<double*>(tsumB_all); a.tmaxB = const_cast<double*>(tmaxB_all);
    a.win_tile0 = win_tile0; (void)win_tiles;
    a.xr = xr; a.thr = thr; a.lw1 = lw1; a.cdfA = cdfA; a.tsumA = tsumA; a.tmaxA = tmaxA; a.mom = mom;
    a.anc = anc;
    if (h->split_l2) lw_launch_stage1<true>(h, dim3(h->sh_Bown, 1), h->lds_bytes_big, a);
    else lw_launch_stage1<false>(h, dim3(h->sh_Bown, 1), h->lds_bytes, a);
    LWCHK(hipGetLastError());
    return SSME_OK;
}

int ssme_lw_shard_mid(ssme_lw_handle h, int32_t t, const double* tsumA_all, const double* tmaxA_all, const double* mom_all) {
    if (!h || t < 1 || !tsumA_all || !tmaxA_all || !mom_all) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    LwArgs a = lw_shard_args(h, t);
    a.tsumA = const_cast<double*>(tsumA_all); a.tmaxA = const_cast<double*>(tmaxA_all); a.mom = const_cast<double*>(mom_all);
    // split level-2: the plan of the k draw (ssme_lw_shard_plan(which = 1)) must precede this call -- it provides m and S
    if (h->split_l2) {
        a.momtot = h->momtot;
        hipLaunchKernelGGL(k_lw_mom_totals, dim3(kNMom, 1), dim3(64), 0, h->stream, a);
        hipLaunchKernelGGL(k_lw_mid<true>, dim3(1), dim3(kThreads), 0, h->stream, a);
    }
    else if (h->B > 256) {
        a.momtot = h->momtot;
        hipLaunchKernelGGL(k_lw_mom_totals, dim3(kNMom, 1), dim3(64), 0, h->stream, a);
        hipLaunchKernelGGL(k_lw_mid<false>, dim3(1), dim3(kThreads), 0, h->stream, a);
    } else hipLaunchKernelGGL(k_lw_mid<false>, dim3(1), dim3(kThreads), 0, h->stream, a);
    LWCHK(hipGetLastError());
    return SSME_OK;
}

int ssme_lw_shard_stage2(ssme_lw_handle h, int32_t t, int32_t win_tile0, int32_t win_tiles, const double* w_xr, const double* w_thr,
                         const double* w_lw1, const double* w_cdfA, const double* tsumA_all, const double* tmaxA_all, double* xB,
                         double* thB, double* cdfB, double* tsumB, double* tmaxB, uint32_t* kidx) {
    if (!h || t < 1 || win_tile0 < -h->sh_margin || win_tiles < 1 || !w_xr || !w_thr || !w_lw1 || !w_cdfA || !tsumA_all || !tmaxA_all || !xB || !thB ||
        !cdfB || !tsumB || !tmaxB)
        return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    LwArgs a = lw_shard_args(h, t);
    a.xr = const_cast<double*>(w_xr); a.thr = const_cast<double*>(w_thr); a.lw1 = const_cast<double*>(w_lw1);
    a.cdfA = const_cast<double*>(w_cdfA); a.tsumA = const_cast<double*>(tsumA_all); a.tmaxA = const_cast<double*>(tmaxA_all);
    a.win_tile0 = win_tile0; (void)win_tiles;
    a.xB = xB; a.thB = thB; a.cdfB = cdfB; a.tsumB = tsumB; a.tmaxB = tmaxB;
    a.kidx = kidx;
    if (h->split_l2) lw_launch_stage2<true>(h, dim3(h->sh_Bown, 1), h->lds_bytes_big, a);
    else lw_launch_stage2<false>(h, dim3(h->sh_Bown, 1), h->lds_bytes, a);
    LWCHK(hipGetLastError());
    h->t = t + 1;
    return SSME_OK;
}

int ssme_lw_shard_finalize(ssme_lw_handle h, int32_t t, const double* tsumB_all, const double* tmaxB_all) {
    if (!h || t < 0 || !tsumB_all || !tmaxB_all) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    LwArgs a = lw_shard_args(h, t);
    a.tsumB = const_cast<double*>(tsumB_all); a.tmaxB = const_cast<double*>(tmaxB_all);
    if (h->split_l2) {
        lw_launch_plan(h, 0, t + 1, 0, tsumB_all, tmaxB_all, false);
        hipLaunchKernelGGL(k_lw_finalize<true>, dim3(1), dim3(kThreads), 0, h->stream, a);
    } else
        hipLaunchKernelGGL(k_lw_finalize<false>, dim3(1), dim3(kThreads), 0, h->stream, a);
    LWCHK(hipGetLastError());
    LWCHK(hipStreamSynchronize(h->stream));
    return SSME_OK;
}

// ---- C++ driver of the sharded Liu-West filter over RCCL (BASELINE.json configs[4]; fixed-halo path) ------------------------
// Per step, on one HIP stream, no host synchronisation inside the time loop:
//   grouped all-gather (tsumB, tmaxB)  [-> plan(0) above 1024 tiles]  -> halo exchange of (xB, theta B, cdfB)  -> stage 1
//   grouped all-gather (tsumA, tmaxA, 16 moment slots per tile)  [-> plan(1)]  -> mid  -> halo exchange of (xr, theta r, g1, cdfA)  -> stage 2
// The stage kernels check their own source tiles against the exchanged window and raise a flag (read once, after the
// series): SSME_ERR_STATE then tells the caller to run the exact, host-planned loop (ssme_amd/sharded.py: ShardedLiuWest).
#define LWNCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    h->err = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); return SSME_ERR_HIP; } } while (0)

static int lw_shard_alloc(ssme_lw_handle h) {
    if (h->sh_xB) return SSME_OK;
    const int world = h->shard_world, Bl = h->sh_Bl;
    int m = Bl / 64 > 4 ? Bl / 64 : 4;
    if (m > Bl) m = Bl;
    if (world == 1) m = 0;
    h->sh_margin = m; h->sh_rows = Bl + 2 * m;
    const size_t row = sizeof(double) * kTile, rows = (size_t)h->sh_rows;
    double** one[] = {&h->sh_xB, &h->sh_cdfB, &h->sh_xr, &h->sh_g1, &h->sh_cdfA};
    for (auto p : one) { LWCHK(hipMalloc(p, row * rows)); LWCHK(hipMemset(*p, 0, row * rows)); }
    double** four[] = {&h->sh_thB, &h->sh_thr};
    for (auto p : four) { LWCHK(hipMalloc(p, row * rows * kDP)); LWCHK(hipMemset(*p, 0, row * rows * kDP)); }
    LWCHK(hipMalloc(&h->sh_locB, sizeof(double) * 2 * Bl));
    LWCHK(hipMalloc(&h->sh_locA, sizeof(double) * 18 * Bl));                 // tile sums | tile maxima | 16 moment slots per tile
    LWCHK(hipMemset(h->sh_locA, 0, sizeof(double) * 18 * Bl));
    double** all[] = {&h->sh_allB_s, &h->sh_allB_m, &h->sh_allA_s, &h->sh_allA_m};
    const size_t nall = (size_t)world * Bl > (size_t)h->Bs ? (size_t)world * Bl : (size_t)h->Bs;        // gathered: world x Bl entries, the first B are tiles
    for (auto p : all) LWCHK(hipMalloc(p, sizeof(double) * nall));
    LWCHK(hipMalloc(&h->sh_mom_all, sizeof(double) * nall * 16));
    LWCHK(hipMemset(h->sh_locB, 0, sizeof(double) * 2 * Bl));
    LWCHK(hipMalloc(&h->sh_flag, sizeof(int32_t) * 4));
    return SSME_OK;
}

// halo exchange with the two neighbouring ranks for a list of buffers (rows of `width` doubles per tile)
static int lw_halo_exchange(ssme_lw_handle h, ncclComm_t comm, std::initializer_list<std::pair<double*, size_t>> bufs) {
    const int world = h->shard_world, rank = h->shard_rank, Bl = h->sh_Bl, m = h->sh_margin;
    if (world == 1 || m == 0) return SSME_OK;
    LWNCCL(rccl().GroupStart());
    for (auto& bw : bufs) {
        double* buf = bw.first;
        const size_t w = bw.second;
        if (rank > 0) {
            LWNCCL(rccl().Send(buf + (size_t)m * w, (size_t)m * w, ncclDouble, rank - 1, comm, h->stream));
            LWNCCL(rccl().Recv(buf, (size_t)m * w, ncclDouble, rank - 1, comm, h->stream));
        }
        if (rank + 1 < world) {
            LWNCCL(rccl().Send(buf + (size_t)Bl * w, (size_t)m * w, ncclDouble, rank + 1, comm, h->stream));
            LWNCCL(rccl().Recv(buf + (size_t)(m + Bl) * w, (size_t)m * w, ncclDouble, rank + 1, comm, h->stream));
        }
    }
    LWNCCL(rccl().GroupEnd());
    h->sh_exchanged += (long)m * ((rank > 0) + (rank + 1 < world));
    return SSME_OK;
}

int ssme_lw_shard_run_series(ssme_lw_handle h, void* nccl_comm, const double* y, const double* z, int32_t T, double* loglik_out) {
    if (!h || !nccl_comm || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1) return SSME_ERR_STATE;
    if (T < 1) return SSME_ERR_LENGTH;
    if (!rccl().ok) { h->err = "RCCL (librccl.so) not found in this process"; return SSME_ERR_UNSUPPORTED; }
    LWCHK(hipSetDevice(h->cfg.device));
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    int rc = lw_shard_alloc(h);
    if (rc != SSME_OK) return rc;
    const int world = h->shard_world, Bl = h->sh_Bl, m = h->sh_margin, tile0 = h->shard_rank * Bl;
    (void)world;
    const size_t TL = kTile, off = (size_t)m * TL;
    rc = ssme_lw_shard_set_plane_tiles(h, h->sh_rows);
    if (rc != SSME_OK) return rc;
    rc = ssme_lw_shard_prepare(h, y, z, T);
    if (rc != SSME_OK) return rc;
    LWCHK(hipMemsetAsync(h->sh_flag, 0, sizeof(int32_t) * 4, h->stream));
    h->sh_exchanged = 0;
    double *tsB = h->sh_locB, *tmB = h->sh_locB + Bl;
    double *tsA = h->sh_locA, *tmA = h->sh_locA + Bl, *momL = h->sh_locA + 2 * Bl;
    auto gatherB = [&]() -> int {
        LWNCCL(rccl().GroupStart());
        LWNCCL(rccl().AllGather(tsB, h->sh_allB_s, (size_t)Bl, ncclDouble, comm, h->stream));
        LWNCCL(rccl().AllGather(tmB, h->sh_allB_m, (size_t)Bl, ncclDouble, comm, h->stream));
        LWNCCL(rccl().GroupEnd());
        return SSME_OK;
    };
    rc = ssme_lw_shard_init(h, h->sh_xB + off, h->sh_thB + off * kDP, h->sh_cdfB + off, tsB, tmB);
    if (rc != SSME_OK) return rc;
    for (int t = 1; t < T; ++t) {
        // resampling schedule m_rs (liu_west_filter.h:1139-1140): a draw closes step t - 1 only if t % m_rs == 0; otherwise every
        // particle continues itself with its carried second-stage weight -- stage 1 reads this rank's own rows, nothing travels
        const bool resampled = (t % h->rs) == 0;
        rc = gatherB();
        if (rc != SSME_OK) return rc;
        if (h->split_l2) { lw_launch_plan(h, 0, t, t, h->sh_allB_s, h->sh_allB_m, resampled); LWCHK(hipGetLastError()); }
        if (resampled) {
            rc = lw_halo_exchange(h, comm, {{h->sh_xB, TL}, {h->sh_thB, TL * kDP}, {h->sh_cdfB, TL}});
            if (rc != SSME_OK) return rc;
            h->sh_check = 1;
            rc = ssme_lw_shard_stage1(h, t, tile0 - m, h->sh_rows, h->sh_xB, h->sh_thB, h->sh_cdfB, h->sh_allB_s, h->sh_allB_m,
                                      h->sh_xr + off, h->sh_thr + off * kDP, h->sh_g1 + off, h->sh_cdfA + off, tsA, tmA, momL, nullptr);
            h->sh_check = 0;
        } else {
            rc = ssme_lw_shard_stage1(h, t, tile0, Bl, h->sh_xB + off, h->sh_thB + off * kDP, h->sh_cdfB + off, h->sh_allB_s, h->sh_allB_m,
                                      h->sh_xr + off, h->sh_thr + off * kDP, h->sh_g1 + off, h->sh_cdfA + off, tsA, tmA, momL, nullptr);
        }
        if (rc != SSME_OK) return rc;
        LWNCCL(rccl().GroupStart());
        LWNCCL(rccl().AllGather(tsA, h->sh_allA_s, (size_t)Bl, ncclDouble, comm, h->stream));
        LWNCCL(rccl().AllGather(tmA, h->sh_allA_m, (size_t)Bl, ncclDouble, comm, h->stream));
        LWNCCL(rccl().AllGather(momL, h->sh_mom_all, (size_t)Bl * 16, ncclDouble, comm, h->stream));
        LWNCCL(rccl().GroupEnd());
        // the plan of the k draw first: above 1024 tiles it provides the (m, S) that mid turns into the first-stage log-sum-exp
        // (SISR form, form 1: no first-stage weights and no k draw -- every particle continues itself, so stage 2 reads this
        //  rank's own stage-1 outputs and the second exchange does not exist: ONE exchange per step)
        if (h->split_l2 && h->form == 0) { lw_launch_plan(h, 1, t, t, h->sh_allA_s, h->sh_allA_m, true); LWCHK(hipGetLastError()); }
        rc = ssme_lw_shard_mid(h, t, h->sh_allA_s, h->sh_allA_m, h->sh_mom_all);
        if (rc != SSME_OK) return rc;
        if (h->form == 0) {
            rc = lw_halo_exchange(h, comm, {{h->sh_xr, TL}, {h->sh_thr, TL * kDP}, {h->sh_g1, TL}, {h->sh_cdfA, TL}});
            if (rc != SSME_OK) return rc;
        }
        h->sh_check = 1;
        rc = ssme_lw_shard_stage2(h, t, tile0 - m, h->sh_rows, h->sh_xr, h->sh_thr, h->sh_g1, h->sh_cdfA, h->sh_allA_s, h->sh_allA_m,
                                  h->sh_xB + off, h->sh_thB + off * kDP, h->sh_cdfB + off, tsB, tmB, nullptr);
        h->sh_check = 0;
        if (rc != SSME_OK) return rc;
    }
    rc = gatherB();
    if (rc != SSME_OK) return rc;
    rc = ssme_lw_shard_finalize(h, T - 1, h->sh_allB_s, h->sh_allB_m);           // synchronises
    if (rc != SSME_OK) return rc;
    // every rank must take the same branch (see shard_series): reduce the per-rank flags before reading them
    LWNCCL(rccl().AllReduce(h->sh_flag, h->sh_flag + 3, 1, ncclInt32, ncclMax, comm, h->stream));
    LWCHK(hipMemcpyAsync(h->sh_stats, h->sh_flag, sizeof(h->sh_stats), hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    if (h->sh_stats[3]) { h->err = "a resampling window left the fixed halo on some rank: run the exact host-planned loop"; return SSME_ERR_STATE; }
    if (loglik_out) return ssme_lw_get_loglik(h, loglik_out);
    return SSME_OK;
}

// this rank's particles and transformed parameters (theta[d * n + i]) after ssme_lw_shard_run_series
int ssme_lw_shard_download(ssme_lw_handle h, double* x_local, double* theta_local, int64_t* exchanged_tiles) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->sh_xB) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    const size_t first = (size_t)h->shard_rank * h->sh_Bl * kTile, off = (size_t)h->sh_margin * kTile;
    const size_t n = (size_t)h->N - first < (size_t)h->sh_Bown * kTile ? (size_t)h->N - first : (size_t)h->sh_Bown * kTile;      // fewer on the last rank
    if (x_local) LWCHK(hipMemcpy(x_local, h->sh_xB + off, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (theta_local) {
        std::vector<double> rec(n * kDP);
        LWCHK(hipMemcpy(rec.data(), h->sh_thB + off * kDP, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) for (int d = 0; d < kDP; ++d) theta_local[(size_t)d * n + i] = rec[i * kDP + d];
    }
    if (exchanged_tiles) *exchanged_tiles = h->sh_exchanged;
    return SSME_OK;
}

int ssme_lw_shard_stats(ssme_lw_handle h, int32_t* out4) {
    if (!h || !out4) return SSME_ERR_INVALID_ARG;
    if (h->shard_world < 1 || !h->sh_xB) return SSME_ERR_STATE;
    out4[0] = h->sh_stats[3]; out4[1] = h->sh_stats[0]; out4[2] = h->sh_stats[1]; out4[3] = h->sh_stats[2];
    return SSME_OK;
}

int ssme_lw_get_loglik(ssme_lw_handle h, double* out) {
    if (!h || !out) return SSME_ERR_INVALID_ARG;
    LWCHK(hipSetDevice(h->cfg.device));
    std::vector<LwScalars> sc(h->R);
    LWCHK(hipMemcpyAsync(sc.data(), h->scal, sizeof(LwScalars) * h->R, hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->R; ++r) out[r] = sc[r].loglik;
    return SSME_OK;
}

int ssme_lw_reset(ssme_lw_handle h) {
    if (!h) return SSME_ERR_INVALID_ARG;
    LWCHK(hipSetDevice(h->cfg.device));
    return lw_reset(h);
}

int ssme_lw_set_debug(ssme_lw_handle h, int32_t flags) {
    if (!h) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0 && (flags & 1)) return SSME_ERR_STATE;   // index recording needs the handle's own buffers
    LWCHK(hipSetDevice(h->cfg.device));
    if ((flags & 1) && !h->anc) {
        LWCHK(hipMalloc(&h->anc, sizeof(uint32_t) * (size_t)h->R * h->Npad));
        LWCHK(hipMalloc(&h->kidx, sizeof(uint32_t) * (size_t)h->R * h->Npad));
        LWCHK(hipMemset(h->anc, 0, sizeof(uint32_t) * (size_t)h->R * h->Npad));
        LWCHK(hipMemset(h->kidx, 0, sizeof(uint32_t) * (size_t)h->R * h->Npad));
    }
    h->debug = flags;
    h->split_l2 = (h->B > kMaxTilesPerFilter || (flags & 4)) ? 1 : ((flags & 8) ? 0 : (h->B > kSplitLevel2Above ? 1 : 0));
    return SSME_OK;
}

int ssme_lw_step(ssme_lw_handle h, const double* y, const double* z, double* out) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles own no particle buffers: ssme_lw_shard_* drives them
    LWCHK(hipSetDevice(h->cfg.device));
    const double z0 = z ? *z : 0.0;
    if (h->gcap < kStepGammaChunk) {
        int rc = lw_ensure_gamma(h, kStepGammaChunk);
        if (rc != SSME_OK) return rc;
    }
    // y and z travel in the kernel arguments, the R results come back through device-mapped pinned memory written by the
    // accounting kernel: no copy operation in either direction
    const double yz[2] = {*y, z0};
    int gi = 0;
    if (h->t > 0) {                              // Gamma tables of both draws, kStepGammaChunk steps at a time
        if (h->t < h->gamma_t0 || h->t >= h->gamma_t0 + h->gamma_rows) {
            lw_launch_gamma(h, h->t, kStepGammaChunk);
            h->gamma_t0 = h->t; h->gamma_rows = kStepGammaChunk;
        }
        gi = h->t - h->gamma_t0;
    }
    mark_results_pending(h->pin, h->R);
    lw_enqueue_step(h, h->t, 0, gi, false, /*finalize_prev=*/false, yz);
    lw_enqueue_finalize(h, h->t, false, h->pin_dev);          // the step API accounts each step right away
    LWCHK(hipGetLastError());
    LWCHK(wait_results(h->stream, h->pin, h->R));
    if (out) for (int r = 0; r < h->R; ++r) out[r] = h->pin[r];
    h->t += 1;
    return SSME_OK;
}

int ssme_lw_run_series(ssme_lw_handle h, const double* y, const double* z, int32_t T, double* loglik_out) {
    if (!h || !y) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles own no particle buffers: ssme_lw_shard_* drives them
    if (T < 1) return SSME_ERR_LENGTH;
    LWCHK(hipSetDevice(h->cfg.device));
    int rc = lw_ensure_capacity(h, T);
    if (rc != SSME_OK) return rc;
    LWCHK(hipMemcpyAsync(h->ybuf, y, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    if (z) LWCHK(hipMemcpyAsync(h->zbuf, z, sizeof(double) * T, hipMemcpyHostToDevice, h->stream));
    else LWCHK(hipMemsetAsync(h->zbuf, 0, sizeof(double) * T, h->stream));
    rc = lw_reset(h);
    if (rc != SSME_OK) return rc;
    LWCHK(hipEventRecord(h->ev0, h->stream));
    lw_launch_gamma(h, 0, T);
    for (int t = 0; t < T; ++t) lw_enqueue_step(h, t, t, t, true, /*finalize_prev=*/t > 0);
    lw_enqueue_finalize(h, T - 1, true);
    LWCHK(hipEventRecord(h->ev1, h->stream));
    LWCHK(hipGetLastError());
    h->t = T;
    std::vector<LwScalars> sc(h->R);
    LWCHK(hipMemcpyAsync(sc.data(), h->scal, sizeof(LwScalars) * h->R, hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    LWCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if (loglik_out) for (int r = 0; r < h->R; ++r) loglik_out[r] = sc[r].loglik;
    return SSME_OK;
}

int ssme_lw_get_per_step(ssme_lw_handle h, double* out, int32_t T) {
    if (!h || !out || T < 1 || T > h->tcap) return SSME_ERR_INVALID_ARG;
    LWCHK(hipSetDevice(h->cfg.device));
    for (int r = 0; r < h->R; ++r)
        LWCHK(hipMemcpyAsync(out + (size_t)r * T, h->per_step + (size_t)r * h->tcap, sizeof(double) * T, hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    return SSME_OK;
}

// [R][8] on the host: E[theta_d] (4), E[x], E[x^2], E[exp(x/2)], E[42] under the last step's weights
static int lw_expect_table(ssme_lw_handle h, std::vector<double>& tab) {
    LwArgs a = lw_args(h);
    // the per-tile moment scratch is free between steps (stage 1 rewrites it before k_lw_mid reads it)
    hipLaunchKernelGGL(k_lw_param_partials, dim3(h->B, h->R), dim3(kThreads), 0, h->stream, a);
    hipLaunchKernelGGL(k_lw_param_means, dim3(h->R), dim3(kWave), 0, h->stream, a, h->scratch);
    LWCHK(hipGetLastError());
    tab.resize((size_t)h->R * kLwNExp);
    LWCHK(hipMemcpyAsync(tab.data(), h->scratch, sizeof(double) * tab.size(), hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    return SSME_OK;
}

int ssme_lw_get_param_means(ssme_lw_handle h, double* out) {
    if (!h || !out) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles own no particle buffers: ssme_lw_shard_* drives them
    if (h->t < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    std::vector<double> tab;
    int rc = lw_expect_table(h, tab);
    if (rc != SSME_OK) return rc;
    for (int r = 0; r < h->R; ++r) for (int d = 0; d < kDP; ++d) out[(size_t)r * kDP + d] = tab[(size_t)r * kLwNExp + d];
    return SSME_OK;
}

int ssme_lw_get_expectations(ssme_lw_handle h, const int32_t* functionals, int32_t n, double* out) {
    if (!h || !out || !functionals || n < 1) return SSME_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) if (functionals[i] < 0 || functionals[i] > 7) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (h->t < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    std::vector<double> tab;
    int rc = lw_expect_table(h, tab);
    if (rc != SSME_OK) return rc;
    static const int slot[8] = {4, 5, 6, 7, 0, 1, 2, 3};      // ids 0-3: x, x^2, exp(x/2), 42; 4-7: phi, mu, sigma, rho
    for (int i = 0; i < n; ++i) for (int r = 0; r < h->R; ++r) out[(size_t)i * h->R + r] = tab[(size_t)r * kLwNExp + slot[functionals[i]]];
    return SSME_OK;
}

int ssme_lw_download_weights(ssme_lw_handle h, int32_t f, double* x, double* theta_untrans, double* w) {
    if (!h || f < 0 || f >= h->R || !w) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;
    if (h->t < 1) return SSME_ERR_STATE;
    LWCHK(hipSetDevice(h->cfg.device));
    if (!h->wscratch) LWCHK(hipMalloc(&h->wscratch, sizeof(double) * (size_t)h->Npad * (1 + kDP)));
    LwArgs a = lw_args(h);
    hipLaunchKernelGGL(k_lw_weights, dim3(h->B), dim3(kThreads), 0, h->stream, a, (int)f, h->wscratch);
    LWCHK(hipGetLastError());
    LWCHK(hipMemcpyAsync(w, h->wscratch, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    if (theta_untrans) for (int d = 0; d < kDP; ++d)
        LWCHK(hipMemcpyAsync(theta_untrans + (size_t)d * h->N, h->wscratch + (size_t)(1 + d) * h->Npad, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    if (x) LWCHK(hipMemcpyAsync(x, h->xB + (size_t)f * h->Npad, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    return SSME_OK;
}

int ssme_lw_download_state(ssme_lw_handle h, int32_t f, double* x, double* theta, uint32_t* kidx, uint32_t* anc, double* thetabar,
                           double* chol) {
    if (!h || f < 0 || f >= h->R) return SSME_ERR_INVALID_ARG;
    if (h->shard_world > 0) return SSME_ERR_STATE;     // sharded handles own no particle buffers: ssme_lw_shard_* drives them
    LWCHK(hipSetDevice(h->cfg.device));
    const size_t off = (size_t)f * h->Npad;
    if (x) LWCHK(hipMemcpyAsync(x, h->xB + off, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    std::vector<double> rec;                       // the device keeps one [4]-record per particle; callers get parameter planes
    if (theta) {
        rec.resize((size_t)h->N * kDP);
        LWCHK(hipMemcpyAsync(rec.data(), h->thB + (size_t)f * h->Npad * kDP, sizeof(double) * rec.size(), hipMemcpyDeviceToHost, h->stream));
    }
    if (kidx || anc) {
        if (!h->anc) return SSME_ERR_STATE;
        if (kidx) LWCHK(hipMemcpyAsync(kidx, h->kidx + off, sizeof(uint32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
        if (anc) LWCHK(hipMemcpyAsync(anc, h->anc + off, sizeof(uint32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
    }
    double p[16];
    LWCHK(hipMemcpyAsync(p, h->prop + (size_t)f * 16, sizeof(p), hipMemcpyDeviceToHost, h->stream));
    LWCHK(hipStreamSynchronize(h->stream));
    if (theta) for (int i = 0; i < h->N; ++i) for (int d = 0; d < kDP; ++d) theta[(size_t)d * h->N + i] = rec[(size_t)i * kDP + d];
    if (thetabar) for (int d = 0; d < kDP; ++d) thetabar[d] = p[d];
    if (chol) { int q = kDP; for (int d = 0; d < kDP; ++d) for (int e = 0; e < kDP; ++e) chol[d * kDP + e] = (e <= d) ? p[q++] : 0.0; }
    return SSME_OK;
}

int ssme_lw_last_elapsed_ms(ssme_lw_handle h, float* ms) {
    if (!h || !ms) return SSME_ERR_INVALID_ARG;
    *ms = h->last_ms;
    return SSME_OK;
}

const char* ssme_lw_last_error(ssme_lw_handle h) { return h ? h->err.c_str() : ""; }

}  // extern "C"
