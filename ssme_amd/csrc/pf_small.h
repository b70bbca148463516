// pf_small.h -- whole-series kernel for filters that fit one tile (N <= 2048): the T steps of
// BSFilter::filter (call site example/estimate_univ_svol.h:121-127) run inside ONE launch, one workgroup
// per filter.  Particles and the integer weight cdf stay in LDS, log-weights in registers; there is no
// kernel boundary and no HBM round trip per step, which is what bounds small filters (a step of the
// tiled kernel costs ~10 us of launch + dependent-latency chain however small N is).
//
// The arithmetic is the tiled kernel's (k_filter_step with B = 1) operation for operation: same Philox
// counters, same spacings / targets / count-search, same tile max and exact fp64-integer scan.  The two
// paths are bit-identical (tests/test_parity_gpu.py::test_small_series_kernel_*); the oracle does not
// distinguish them.  Used by ssme_pf_run_series when the filter has one tile (the shipped example's
// N = 500, example/main.cpp:9) -- the step API and multi-tile filters use k_filter_step.
#pragma once
#include "pf_kernels.h"

namespace ssme {

// min(#{ j < P : tile[j] < target }, P-1) for two targets at once by a radix-8 descent: the probes of one level are
// independent loads, so a search costs log8(P) dependent LDS round trips instead of log2(P) -- this kernel runs one
// wave per SIMD and is bound by dependent latency, not by issue slots.  Same count as count_less_pow2 (monotone tile).
template <int P>
__device__ __forceinline__ void count_less_radix8_x2(const double* tile, double t0, double t1, int& j0, int& j1) {
    int p0 = 0, p1 = 0;
#pragma unroll
    for (int w = P; w > 1;) {
        const int radix = (w >= 8) ? 8 : w;          // P is a power of two: the last level may be radix 2 or 4
        const int s = w / radix;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int k = 1; k < radix; ++k) {
            c0 += (tile[p0 + k * s - 1] < t0) ? 1 : 0;
            c1 += (tile[p1 + k * s - 1] < t1) ? 1 : 0;
        }
        p0 += c0 * s; p1 += c1 * s;
        w = s;
    }
    j0 = p0; j1 = p1;
}

// Log conditional likelihoods of a whole series outside the time loop (both one-tile kernels): the loop only records
// (m_t, S_t) (thread 0, ms[2t], ms[2t+1]); here the T logarithms are evaluated in parallel and one wave replays the
// sequential accounting (ll_t = lse_t - prev_t, loglik += ll_t) in loop order, 64 steps per load.  Same arithmetic and
// order as kf_finalize step by step, so the same bits.  Called by every thread after a barrier that follows the last record.
template <int NT>
__device__ __forceinline__ void series_accounting(const StepArgs& a, int r, int T, double* ms, double m_last, double S_last,
                                                  double A_prev, double mb_prev) {
    const int tid = threadIdx.x;
    for (int t = tid; t < T; t += NT) {
        const double m = ms[2 * t], S = ms[2 * t + 1];
        const double Sd = (S > 0.0) ? dldexp(S, -a.rshift) : dnan();
        ms[2 * t] = m + dlog(Sd);                     // lse_t
    }
    __syncthreads();
    if (tid < 64) {
        double prev = a.scal[r].prev, loglik = a.scal[r].loglik, last = 0.0;
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + tid;
            const double lse = (t < T) ? ms[2 * t] : 0.0;
            double mine = 0.0;
            const int nb = (T - t0 < 64) ? T - t0 : 64;
            for (int j = 0; j < nb; ++j) {
                const double lj = readlane_f64(lse, j);
                const double ll = lj - prev;
                loglik = loglik + ll;
                prev = (((t0 + j + 1) % a.resamp_sched) == 0) ? a.logN : lj;
                last = ll;
                if (tid == j) mine = ll;
            }
            if (t < T && a.per_step) a.per_step[(size_t)r * a.Tcap + t] = mine;
        }
        if (tid == 0) {
            FilterScalars* o = a.scal + r;
            o->m = m_last; o->S = S_last; o->prev = prev; o->loglik = loglik; o->last_ll = last;
            a.tsum_out[(size_t)r * a.Bs] = A_prev;
            a.tmax_out[(size_t)r * a.Bs] = mb_prev;
        }
    }
}

// grid = (R filters), block = NT, covers P = 2*NT*NK >= N particle slots (P a power of two, 256 .. 2048).
// a.x_out / cdf_out / tsum_out / tmax_out / logw receive the state after the last step (as k_filter_step
// leaves it), a.scal the accumulated log-likelihood; a.t / yi / gi are ignored (t = yi = gi = 0 .. T-1).
template <int MODEL, int NT, int NK>
__global__ __launch_bounds__(NT) void k_filter_series_small(const StepArgs a, const int T) {
    constexpr int P = 2 * NT * NK;
    __shared__ __attribute__((aligned(16))) double lds_x[P];
    __shared__ __attribute__((aligned(16))) double lds_cdf[P];
    __shared__ double lds_seg_a[16];
    __shared__ double lds_seg_c[16];
    __shared__ double lds_d2[16];
    __shared__ __attribute__((aligned(16))) DrawTabs lds_dtab;
    __shared__ __attribute__((aligned(16))) ExpTabEntry lds_etab[SSME_EXP_TABLE_SIZE];

    const int tid = threadIdx.x;
    load_log_table<NT>(&lds_dtab);
    load_exp_table<NT>(lds_etab);
    __syncthreads();
    const int r = blockIdx.x;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
    const size_t rowoff = (size_t)r * a.Npad;
    const ModelConst mc = a.mc[r];
    const bool multinomial_kind = a.resampler == RESAMP_MULTINOMIAL;

    double* ms = a.small_ms + (size_t)r * a.Tcap * 2;
    double xcur[NK][2], lwcur[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) { xcur[k][0] = 0.0; xcur[k][1] = 0.0; lwcur[k][0] = 0.0; lwcur[k][1] = 0.0; }
    double A_prev = 0.0, mb_prev = 0.0;

    // next step's inputs are requested one step ahead (uniform scalar loads)
    double y_n = a.y[0], z_n = a.z ? a.z[0] : 0.0;
    double gam_n = 0.0, pgam_n = 0.0, G_n = 1.0;

    for (int t = 0; t < T; ++t) {
        const double y = y_n, zcov = z_n;
        const double gam = gam_n, pgam = pgam_n, G = G_n;
        if (t + 1 < T) {
            y_n = a.y[t + 1];
            if (a.z) z_n = a.z[t + 1];
            if (multinomial_kind && ((t + 1) % a.resamp_sched == 0)) {
                const size_t gidx = ((size_t)(t + 1) * a.R + r) * a.B;          // B = 1, tile 0
                gam_n = a.gam[gidx]; pgam_n = a.pgam[gidx]; G_n = a.gtot[(size_t)(t + 1) * a.R + r];
            }
        }
        const bool resampled = (t > 0) && (t % a.resamp_sched == 0);
        const bool multinomial = resampled && multinomial_kind;

        // --- level-2 with one tile (level2_scan for B = 1) and the log conditional likelihood of step t-1 ---
        double S = 0.0, R0 = 0.0;
        if (t > 0) {
            const double m = (mb_prev != mb_prev) ? dnan() : mb_prev;
            // one tile: m_b - m is +0 unless the max is NaN or infinite, and the shift is 0 (Npad = 2048); exp(0) 2^0 = 1 exactly
            const double dm = mb_prev - m;
            const int sh = a.rshift - kTileShift;
            const double Ap = (dm == 0.0 && sh == 0) ? __builtin_rint(A_prev) : __builtin_rint(A_prev * dexp_scaled_t(dm, sh, lds_etab));
            S = Ap;
            R0 = A_prev / Ap;
            if (tid == 0) { ms[2 * (t - 1)] = m; ms[2 * (t - 1) + 1] = S; }       // its logarithm is taken after the loop
        }

        // --- standard normals of this step: independent of the resampling chain, issued next to the spacings so that the
        //     two instruction streams interleave ---
        // one Philox call per pair feeds the normals (words 0-1) and the pair's two spacings (words 2-3): pair_words
        double zn[NK][2];
        double le[NK][2], se = 1.0;
        {
            double qe[NK][2];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const u32x4 o = pair_words((uint32_t)(k * NT + tid), (uint32_t)t, rep, key0, key1);
                pair_normals(o.v0, o.v1, &lds_dtab, &zn[k][0], &zn[k][1]);
                qe[k][0] = 0.0; qe[k][1] = 0.0;
                if (multinomial) {
                    const int i0 = (k * NT + tid) * 2;
                    double e0, e1;
                    pair_spacings(o, lds_dtab.log, &e0, &e1);
                    qe[k][0] = (i0 < a.N) ? __builtin_rint(e0 * 34359738368.0 /* 2^35 */) : 0.0;
                    qe[k][1] = (i0 + 1 < a.N) ? __builtin_rint(e1 * 34359738368.0) : 0.0;
                }
            }
            // --- exponential spacings (multinomial), exact scan ---
            if (multinomial) block_scan_f64<NT, NK>(qe, le, se, lds_seg_a);
        }

        double xin[NK][2], lw_old[NK][2];
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k) { xin[k][0] = 0.0; xin[k][1] = 0.0; lw_old[k][0] = 0.0; lw_old[k][1] = 0.0; }
        } else if (!resampled) {
#pragma unroll
            for (int k = 0; k < NK; ++k) { xin[k][0] = xcur[k][0]; xin[k][1] = xcur[k][1]; lw_old[k][0] = lwcur[k][0]; lw_old[k][1] = lwcur[k][1]; }
        } else {
            double t_scale, u0 = 0.0;
            if (a.resampler == RESAMP_MULTINOMIAL) t_scale = S / G;
            else {
                t_scale = S / (double)a.N;
                if (a.resampler == RESAMP_SYSTEMATIC) {
                    const u32x4 ox = philox4x32_10(0u, (uint32_t)t, rep, STREAM_RESAMP_EXTRA, key0, key1);
                    u0 = u01_co(ox.v0, ox.v1);
                }
            }
            const double ratio = gam / se;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i0 = (k * NT + tid) * 2;
                double tau[2];
                if (a.resampler == RESAMP_MULTINOMIAL) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double t1 = ratio * le[k][c];
                        const double t2 = pgam + t1;
                        tau[c] = __builtin_ceil(t2 * t_scale);
                    }
                } else if (a.resampler == RESAMP_SYSTEMATIC) {
                    tau[0] = __builtin_ceil(((double)i0 + u0) * t_scale);
                    tau[1] = __builtin_ceil(((double)(i0 + 1) + u0) * t_scale);
                } else {
                    const u32x4 o = philox4x32_10((uint32_t)(i0 >> 1), (uint32_t)t, rep, STREAM_RESAMP, key0, key1);
                    const double v0 = u01_co(o.v0, o.v1), v1 = u01_co(o.v2, o.v3);
                    if (a.resampler == RESAMP_STRATIFIED) {
                        tau[0] = __builtin_ceil(((double)i0 + v0) * t_scale);
                        tau[1] = __builtin_ceil(((double)(i0 + 1) + v1) * t_scale);
                    } else {
                        tau[0] = __builtin_ceil(v0 * S);
                        tau[1] = __builtin_ceil(v1 * S);
                    }
                }
                const double tl0 = __builtin_ceil((tau[0] - 0.0) * R0), tl1 = __builtin_ceil((tau[1] - 0.0) * R0);
                int jj[2];
                count_less_radix8_x2<P>(lds_cdf, tl0, tl1, jj[0], jj[1]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int anc = jj[c] < a.N - 1 ? jj[c] : a.N - 1;
                    if (a.anc && (i0 + c) < a.N) a.anc[rowoff + i0 + c] = (uint32_t)anc;
                    xin[k][c] = lds_x[anc];
                    lw_old[k][c] = 0.0;
                }
            }
        }

        // --- standard normals, fSamp / q1Samp, logGEv, tile max ---
        double lg[NK][2];
        double mx = -dinf();
        bool nan = false;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i0 = (k * NT + tid) * 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double xn = (t == 0) ? zn[k][c] * mc.a2 : model_prop<MODEL>(mc, xin[k][c], zn[k][c], zcov, lds_etab);
                const double l = lw_old[k][c] + model_logg<MODEL>(mc, y, xn, lds_etab);
                const bool valid = (i0 + c) < a.N;
                xcur[k][c] = valid ? xn : 0.0;
                lg[k][c] = valid ? l : -dinf();
                if (valid) { nan = nan || (l != l); mx = (l > mx) ? l : mx; }
            }
            lwcur[k][0] = lg[k][0]; lwcur[k][1] = lg[k][1];
        }
        const double mb = block_max_nanprop<NT>(mx, nan, lds_d2);    // barrier: every search / gather of this step is done

        // --- tile-local fixed-point weights, exact scan -> the next step's cdf (LDS) ---
        double q[NK][2], inc[NK][2], total;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i0 = (k * NT + tid) * 2;
            q[k][0] = (i0 < a.N) ? __builtin_rint(dexp_scaled_t(lg[k][0] - mb, kTileShift, lds_etab)) : 0.0;
            q[k][1] = (i0 + 1 < a.N) ? __builtin_rint(dexp_scaled_t(lg[k][1] - mb, kTileShift, lds_etab)) : 0.0;
        }
        block_scan_f64<NT, NK>(q, inc, total, lds_seg_c);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i0 = (k * NT + tid) * 2;
            *reinterpret_cast<double2*>(lds_cdf + i0) = make_double2(inc[k][0], inc[k][1]);
            *reinterpret_cast<double2*>(lds_x + i0) = make_double2(xcur[k][0], xcur[k][1]);
        }
        A_prev = total;
        mb_prev = mb;
        __syncthreads();                                              // cdf / x of step t visible to step t+1
    }

    // --- the last step's (m, S), then the accounting of the whole series (series_accounting) and the state hand-over ---
    {
        const double m_last = (mb_prev != mb_prev) ? dnan() : mb_prev;
        const double S_last = __builtin_rint(A_prev * dexp_scaled_t(mb_prev - m_last, a.rshift - kTileShift, lds_etab));
        if (tid == 0) { ms[2 * (T - 1)] = m_last; ms[2 * (T - 1) + 1] = S_last; }
        __syncthreads();
        series_accounting<NT>(a, r, T, ms, m_last, S_last, A_prev, mb_prev);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i0 = (k * NT + tid) * 2;
        *reinterpret_cast<double2*>(a.x_out + rowoff + i0) = make_double2(xcur[k][0], xcur[k][1]);
        *reinterpret_cast<double2*>(a.cdf_out + rowoff + i0) = *reinterpret_cast<const double2*>(lds_cdf + i0);
        if (a.logw) *reinterpret_cast<double2*>(a.logw + rowoff + i0) = make_double2(lwcur[k][0], lwcur[k][1]);
    }
    if (P < kTile) {
        // slots beyond P: x = 0, cdf flat at the tile sum, log-weight -inf -- what k_filter_step writes there
        for (int i = P + tid * 2; i < kTile; i += NT * 2) {
            *reinterpret_cast<double2*>(a.x_out + rowoff + i) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(a.cdf_out + rowoff + i) = make_double2(A_prev, A_prev);
            if (a.logw) *reinterpret_cast<double2*>(a.logw + rowoff + i) = make_double2(-dinf(), -dinf());
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// One particle per lane (N <= 512; measured slower than the pair kernel at 1024).  The pair kernel above runs one wave per SIMD, each carrying the ~1000-instruction
// chain of a particle PAIR, and is bound by dependent-instruction latency.  Here a lane owns ONE particle: the two lanes
// of a pair both evaluate the pair's Philox call and Box-Muller (lanes are free, instructions are not) and each keeps
// its own half, so a wave's chain per step is ~55 % of the pair kernel's and twice as many waves share a SIMD.
// The log conditional likelihoods leave the time loop altogether: a step only records (m_t, S_t) (thread 0, 16 bytes);
// after the loop the T logarithms are evaluated in parallel and one wave replays the sequential accounting
// (ll_t = lse_t - prev_t, loglik += ll_t) in the order of the loop, 64 steps per load.  Same arithmetic, same bits.
// grid = (R), block = NT = P particle slots (64 .. 512); a.small_ms: [R][Tcap][2] scratch.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_scan1_f64(double v, double& incl, double& total, double* lds_seg) {
    constexpr int NSEG = NT / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double inc = wave_incl_scan_f64(v);
    if constexpr (NSEG == 1) { incl = inc; total = readlane_f64(inc, 63); return; }
    if (lane == 63) lds_seg[wave] = inc;
    __syncthreads();
    double sv = (lane & 15) < NSEG ? lds_seg[lane & 15] : 0.0;
    sv = sv + dpp_f64_zero<0x111, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x112, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x114, 0xF>(sv);
    sv = sv + dpp_f64_zero<0x118, 0xF>(sv);
    total = readlane_f64(sv, 15);
    const double pre = wave ? readlane_f64(sv, wave - 1) : 0.0;
    incl = pre + inc;
}

// two independent block scans behind ONE barrier (their DPP steps interleave): v -> (incl, total), u -> (incl_u, total_u)
template <int NT>
__device__ __forceinline__ void block_scan2_f64(double v, double& incl, double& total, double* lds_seg, double u, double& incl_u,
                                                double& total_u, double* lds_seg_u) {
    constexpr int NSEG = NT / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double inc = wave_incl_scan_f64(v), incu = wave_incl_scan_f64(u);
    if constexpr (NSEG == 1) { incl = inc; total = readlane_f64(inc, 63); incl_u = incu; total_u = readlane_f64(incu, 63); return; }
    if (lane == 63) { lds_seg[wave] = inc; lds_seg_u[wave] = incu; }
    __syncthreads();
    double sv = (lane & 15) < NSEG ? lds_seg[lane & 15] : 0.0, su = (lane & 15) < NSEG ? lds_seg_u[lane & 15] : 0.0;
    sv = sv + dpp_f64_zero<0x111, 0xF>(sv); su = su + dpp_f64_zero<0x111, 0xF>(su);
    sv = sv + dpp_f64_zero<0x112, 0xF>(sv); su = su + dpp_f64_zero<0x112, 0xF>(su);
    sv = sv + dpp_f64_zero<0x114, 0xF>(sv); su = su + dpp_f64_zero<0x114, 0xF>(su);
    sv = sv + dpp_f64_zero<0x118, 0xF>(sv); su = su + dpp_f64_zero<0x118, 0xF>(su);
    total = readlane_f64(sv, 15); total_u = readlane_f64(su, 15);
    incl = (wave ? readlane_f64(sv, wave - 1) : 0.0) + inc;
    incl_u = (wave ? readlane_f64(su, wave - 1) : 0.0) + incu;
}

// min(#{ j < P : tile[j] < target }, P-1) by a radix-8 descent (see count_less_radix8_x2)
template <int P>
__device__ __forceinline__ int count_less_radix8(const double* tile, double t0) {
    int p0 = 0;
#pragma unroll
    for (int w = P; w > 1;) {
        const int radix = (w >= 8) ? 8 : w;
        const int s = w / radix;
        int c0 = 0;
#pragma unroll
        for (int k = 1; k < radix; ++k) c0 += (tile[p0 + k * s - 1] < t0) ? 1 : 0;
        p0 += c0 * s;
        w = s;
    }
    return p0;
}

template <int MODEL, int NT>
__global__ __launch_bounds__(NT) void k_filter_series_lane(const StepArgs a, const int T) {
    constexpr int P = NT;
    __shared__ double lds_x[P];
    __shared__ double lds_cdf[P];
    __shared__ double lds_seg_a[16];
    __shared__ double lds_seg_c[16];
    __shared__ double lds_d2[16];
    __shared__ __attribute__((aligned(16))) DrawTabs lds_dtab;
    __shared__ __attribute__((aligned(16))) ExpTabEntry lds_etab[SSME_EXP_TABLE_SIZE];

    const int tid = threadIdx.x;
    load_log_table<NT>(&lds_dtab);
    load_exp_table<NT>(lds_etab);
    __syncthreads();
    const int r = blockIdx.x;
    const uint32_t rep = a.first_filter + (uint32_t)r;
    const uint32_t key0 = a.keyp[0], key1 = a.keyp[1];
    const size_t rowoff = (size_t)r * a.Npad;
    const ModelConst mc = a.mc[r];
    const bool multinomial_kind = a.resampler == RESAMP_MULTINOMIAL;
    const bool valid = tid < a.N;
    const int c = tid & 1;                            // which half of the pair (tid >> 1) this lane keeps
    double* ms = a.small_ms + (size_t)r * a.Tcap * 2;

    double xcur = 0.0, lwcur = 0.0;
    double A_prev = 0.0, mb_prev = 0.0;
    double y_n = a.y[0], z_n = a.z ? a.z[0] : 0.0;
    double gam_n = 0.0, pgam_n = 0.0, G_n = 1.0;

    // The draws of a step do not depend on the filter's state, and at one or two waves per SIMD the step is a chain of dependent
    // instructions: the Philox call, the Box-Muller pair and the spacing of step t + 1 are therefore computed DURING step t, in the
    // block that waits for the search's LDS reads, and the spacings' block scan shares the barrier of the weights' scan
    // (block_scan2_f64).  Carried to the next iteration: the normal, the spacing's inclusive sum and the total.  Same arithmetic.
    double zn_n = 0.0, le_n = 0.0, se_n = 1.0, qe_n = 0.0;
    auto draw_for = [&](int tn) {
        // the pair's Philox call: words 0-1 -> Box-Muller (this lane keeps cos or sin), words 2 / 3 -> this lane's spacing
        const u32x4 o = pair_words((uint32_t)(tid >> 1), (uint32_t)tn, rep, key0, key1);
        double z0, z1;
        pair_normals(o.v0, o.v1, &lds_dtab, &z0, &z1);
        zn_n = c ? z1 : z0;
        const bool mult = multinomial_kind && tn > 0 && (tn % a.resamp_sched == 0);
        const double e = -dlog_u32(u01_mid32(c ? o.v3 : o.v2), lds_dtab.log);
        qe_n = (valid && mult) ? __builtin_rint(e * 34359738368.0 /* 2^35 */) : 0.0;
    };
    draw_for(0);

    for (int t = 0; t < T; ++t) {
        const double y = y_n, zcov = z_n;
        const double gam = gam_n, pgam = pgam_n, G = G_n;
        const double zn = zn_n, le = le_n, se = se_n;
        if (t + 1 < T) {
            y_n = a.y[t + 1];
            if (a.z) z_n = a.z[t + 1];
            if (multinomial_kind && ((t + 1) % a.resamp_sched == 0)) {
                const size_t gidx = ((size_t)(t + 1) * a.R + r) * a.B;
                gam_n = a.gam[gidx]; pgam_n = a.pgam[gidx]; G_n = a.gtot[(size_t)(t + 1) * a.R + r];
            }
        }
        const bool resampled = (t > 0) && (t % a.resamp_sched == 0);

        // --- level-2 with one tile; (m, S) of step t-1 go to the scratch, their logarithm is taken after the loop ---
        double S = 0.0, R0 = 0.0;
        if (t > 0) {
            const double m = (mb_prev != mb_prev) ? dnan() : mb_prev;
            const double dm = mb_prev - m;
            const int sh = a.rshift - kTileShift;
            const double Ap = (dm == 0.0 && sh == 0) ? __builtin_rint(A_prev) : __builtin_rint(A_prev * dexp_scaled_t(dm, sh, lds_etab));
            S = Ap;
            R0 = A_prev / Ap;
            if (tid == 0) { ms[2 * (t - 1)] = m; ms[2 * (t - 1) + 1] = S; }
        }

        double xin = 0.0, lw_old = 0.0;
        if (t == 0) {
            draw_for(t + 1);
        } else if (!resampled) {
            xin = xcur; lw_old = lwcur;
            draw_for(t + 1);
        } else {
            double tau;
            if (a.resampler == RESAMP_MULTINOMIAL) {
                const double t_scale = S / G;
                const double ratio = gam / se;
                const double t1 = ratio * le;
                const double t2 = pgam + t1;
                tau = __builtin_ceil(t2 * t_scale);
            } else if (a.resampler == RESAMP_SYSTEMATIC) {
                const double t_scale = S / (double)a.N;
                const u32x4 ox = philox4x32_10(0u, (uint32_t)t, rep, STREAM_RESAMP_EXTRA, key0, key1);
                const double u0 = u01_co(ox.v0, ox.v1);
                tau = __builtin_ceil(((double)tid + u0) * t_scale);
            } else {
                const double t_scale = S / (double)a.N;
                const u32x4 o = philox4x32_10((uint32_t)(tid >> 1), (uint32_t)t, rep, STREAM_RESAMP, key0, key1);
                const double v = c ? u01_co(o.v2, o.v3) : u01_co(o.v0, o.v1);
                tau = (a.resampler == RESAMP_STRATIFIED) ? __builtin_ceil(((double)tid + v) * t_scale) : __builtin_ceil(v * S);
            }
            const double tl = __builtin_ceil((tau - 0.0) * R0);
            const int jj = count_less_radix8<P>(lds_cdf, tl);
            draw_for(t + 1);                                   // fills the waits of the descent's LDS reads and of the gather
            const int anc = jj < a.N - 1 ? jj : a.N - 1;
            if (a.anc && valid) a.anc[rowoff + tid] = (uint32_t)anc;
            xin = lds_x[anc];
        }

        // --- fSamp / q1Samp, logGEv, tile max ---
        const double xn = (t == 0) ? zn * mc.a2 : model_prop<MODEL>(mc, xin, zn, zcov, lds_etab);
        const double l = lw_old + model_logg<MODEL>(mc, y, xn, lds_etab);
        xcur = valid ? xn : 0.0;
        const double lg = valid ? l : -dinf();
        lwcur = lg;
        const bool nan = valid && (l != l);
        const double mx = (valid && l > -dinf()) ? l : -dinf();
        const double mb = block_max_nanprop<NT>(mx, nan, lds_d2);    // barrier: every search / gather of this step is done

        // --- tile-local fixed-point weights, exact scan -> the next step's cdf (LDS); the next step's spacings ride along ---
        const double q = valid ? __builtin_rint(dexp_scaled_t(lg - mb, kTileShift, lds_etab)) : 0.0;
        double inc, total;
        block_scan2_f64<NT>(q, inc, total, lds_seg_c, qe_n, le_n, se_n, lds_seg_a);
        lds_cdf[tid] = inc;
        lds_x[tid] = xcur;
        A_prev = total;
        mb_prev = mb;
        __syncthreads();                                              // cdf / x of step t visible to step t+1
    }

    // --- the last step's (m, S); then every log conditional likelihood: logarithms in parallel, accounting in loop order ---
    const double m_last = (mb_prev != mb_prev) ? dnan() : mb_prev;
    const double S_last = __builtin_rint(A_prev * dexp_scaled_t(mb_prev - m_last, a.rshift - kTileShift, lds_etab));
    if (tid == 0) { ms[2 * (T - 1)] = m_last; ms[2 * (T - 1) + 1] = S_last; }
    __syncthreads();
    series_accounting<NT>(a, r, T, ms, m_last, S_last, A_prev, mb_prev);
    // --- state hand-over: what k_filter_step leaves (slots beyond N: x = 0, flat cdf, log-weight -inf) ---
    a.x_out[rowoff + tid] = xcur;
    a.cdf_out[rowoff + tid] = lds_cdf[tid];
    if (a.logw) a.logw[rowoff + tid] = lwcur;
    for (int i = P + tid; i < kTile; i += NT) {
        a.x_out[rowoff + i] = 0.0;
        a.cdf_out[rowoff + i] = A_prev;
        if (a.logw) a.logw[rowoff + i] = -dinf();
    }
}

}  // namespace ssme
