/* ssme_pf.h -- C ABI of the MI355X-native bootstrap-particle-filter core.
 *
 * Drop-in boundary for the hot path behind ssme's BSFilter<>::filter() step.  In the
 * reference this boundary is compile-time C++ inheritance from the external `pf` library
 * (no FFI exists): a model derives from pf::filters::BSFilter<nparts,dimx,dimy,resampT,
 * float_t> (example/univ_svol_bootstrap_filter.h:18) and callers use
 *     mod.filter(y_t);  logLike += mod.getLogCondLike();     example/estimate_univ_svol.h:124-125
 *     mods[i].filter(y_t, z_t, fs); getExpectations(); getLogCondLike();
 *                                                             include/ssme/pswarm_filter.h:86-92,380-388
 * Host virtual callbacks cannot run per particle on a GPU, so the model is selected by
 * enum and compiled into the kernels.  Plain pointers and sizes only; no C++/torch types.
 * Every function returns an int status (0 = ok); nothing throws across this boundary.
 * NaN / -inf log-likelihoods are VALUES, not errors (reference: ada_pmmh_mvn.h:349,357).
 *
 * Threading: re-entrant, no global mutable state.  One handle = one HIP stream; distinct
 * handles may be used concurrently from distinct host threads (as thread_pool.h:242-245
 * calls log_like_eval concurrently on distinct model objects); a single handle is not
 * thread-safe (same as one BSFilter object).
 */
#ifndef SSME_PF_H
#define SSME_PF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
enum {
    SSME_OK = 0,
    SSME_ERR_INVALID_ARG = 1,   /* reference: std::invalid_argument                     */
    SSME_ERR_LENGTH = 2,        /* reference: std::length_error (empty data, :113)      */
    SSME_ERR_UNSUPPORTED = 3,   /* valid request this build does not implement          */
    SSME_ERR_HIP = 4,           /* HIP runtime failure; text via ssme_pf_last_error     */
    SSME_ERR_STATE = 5          /* call order (e.g. step before set_params)             */
};

/* models: per-particle callbacks compiled into the kernels */
enum {
    SSME_MODEL_SVOL = 0,          /* svol_bs, example/univ_svol_bootstrap_filter.h:55-103;
                                     theta = (beta, phi, sigma)  [sigma = sqrt(ss), :58-60] */
    SSME_MODEL_SVOL_LEVERAGE = 1, /* svol_leverage, test/test_pswarm.cpp:80-134;
                                     theta = (phi, mu, sigma, rho); covariate z_t = y_{t-1} */
    SSME_MODEL_LIN_GAUSS = 2,     /* x' = phi x + sigma e, y ~ N(x, tau^2); theta = (phi, sigma, tau).
                                     Not in the reference: exact-Kalman correctness anchor.  */
    SSME_MODEL_USER0 = 3          /* the model compiled in from a user header (ssme_amd/csrc/model_api.h: the counterpart of
                                     deriving a class from BSFilter and overriding fSamp / logGEv / q1Samp,
                                     example/univ_svol_bootstrap_filter.h:37-41); SSME_ERR_UNSUPPORTED in a library
                                     built without one.  theta has ssme_pf_user_model_n_theta() entries.  */
};

/* resamplers (pf::resamplers::*, in-tree twin include/ssme/liu_west_filter.h:91-145) */
enum {
    SSME_RESAMP_MULTINOMIAL = 0,     /* multinomial by sorted uniforms (exponential spacings),
                                        the algorithm of liu_west_filter.h:105-139 / mn_resamp_fast1 */
    SSME_RESAMP_SYSTEMATIC = 1,
    SSME_RESAMP_STRATIFIED = 2,
    SSME_RESAMP_MULTINOMIAL_IID = 3  /* multinomial with unsorted iid uniforms (mn_resampler form) */
};

enum { SSME_F64 = 0, SSME_F32 = 1 };

/* built-in functionals h(x) for weighted expectations (std::function cannot run on device) */
enum { SSME_H_X = 0, SSME_H_X2 = 1, SSME_H_VOL = 2 /* exp(x/2) */, SSME_H_CONST42 = 3 };

typedef struct ssme_pf_s* ssme_pf_handle;

typedef struct ssme_pf_config {
    int32_t  model;            /* SSME_MODEL_*                                               */
    int32_t  n_particles;      /* N per filter (reference: template parameter nparts)        */
    int32_t  n_filters;        /* R independent filters/replicates held by this handle:
                                  thread_pool's num_pfilters (thread_pool.h:189-215) or the
                                  swarm's nparamparts (pswarm_filter.h:280-304)              */
    int32_t  dtype;            /* SSME_F64, or SSME_F32 = float at the boundary (the reference
                                  built with float_t = float, example/main.cpp:13): y, z and theta
                                  are rounded to float on entry and every returned log-likelihood,
                                  expectation, particle and weight is rounded to float; the
                                  arithmetic stays fp64 (plain fp32 VALU issues at the fp64 rate on
                                  gfx950, so float arithmetic would buy no time).  Not for sharded
                                  handles or the parity downloads (ssme_pf_download_state / _scalars
                                  hand out the raw fp64 state)                                 */
    int32_t  resampler;        /* SSME_RESAMP_*                                              */
    int32_t  resamp_sched;     /* resample every k-th step; 1 = reference default            */
    uint64_t seed;             /* Philox4x32-10 key (reference RNGs are clock-seeded)        */
    int32_t  device;           /* HIP device ordinal                                         */
    uint32_t first_filter_id;  /* global id of filter 0 of this handle; enters the Philox
                                  counter, so sharding R over GPUs keeps every stream        */
    int32_t  tile_particles;   /* particles per tile: 2048, 1024 or 512, or 0 = ssme_pf_default_tile(N, bank size), where
                                  the bank size is n_filters_total if > 0, else n_filters.
                                  Part of the arithmetic specification:
                                  weights are fixed point relative to their tile's maximum and
                                  the resampler draws one Gamma variate per tile */
    int32_t  n_filters_total;  /* 0, or the size of the WHOLE bank this handle holds a part of (replicates or swarm
                                  members dealt to several GPUs, or one handle per member): the default tile is then
                                  chosen from the bank, so that filter (seed, id, N) gives the same bits however the
                                  bank is split over handles.  Ignored when tile_particles != 0.             */
} ssme_pf_config;

/* Allocates device state for R filters of N particles.  Replaces construction of the
 * model object (estimate_univ_svol.h:119), but the handle is reusable across theta. */
int ssme_pf_create(const ssme_pf_config* cfg, ssme_pf_handle* out);
int ssme_pf_destroy(ssme_pf_handle h);
/* The tile size tile_particles = 0 stands for: 2048 for N <= 2048; else 512 while bank_filters * ceil(N / 512) <= 256
 * (one workgroup per CU), else 1024 while bank_filters * ceil(N / 1024) <= 512, else 2048 -- a mid-size bank then spreads
 * over the chip (profiles/r02_tile_sweep.txt).  It reads the BANK, not the handle: callers that split a bank over GPUs or
 * handles pass n_filters_total (or this value as tile_particles) so that results do not depend on the split. */
int ssme_pf_default_tile(int32_t n_particles, int32_t bank_filters);
/* Length of theta for SSME_MODEL_USER0, or 0 when this library was built without a user model. */
int ssme_pf_user_model_n_theta(void);
/* State and observation dimension of SSME_MODEL_USER0 (BSFilter<nparts, dimx, dimy, ...>: 1 .. 4 each; ssme_amd/csrc/model_api.h);
 * SSME_ERR_UNSUPPORTED and zeros without a user model.  With dim_y > 1 every y argument of this interface is dim_y values per time
 * step (run_series: y[t * dim_y + j]; step: dim_y values); with dim_x > 1 ssme_pf_download_state and ssme_pf_download_weights return
 * x[d * N + i] (dim_x planes; the weights are what a host-side functional h(x) of the whole state needs) and the device functionals
 * see component 0.  Vector models are unsharded, SSME_F64. */
int ssme_pf_user_model_dims(int32_t* dim_x, int32_t* dim_y);

/* UNTRANSFORMED parameters, as the reference's model ctors receive them from
 * pack::get_untrans_params (univ_svol_bootstrap_filter.h:55-61).  theta is
 * [n_rows][n_theta]; n_rows = 1 broadcasts one theta to all filters (PMMH replicates),
 * n_rows = n_filters gives each filter its own row (swarm).  Also resets time to 0. */
int ssme_pf_set_params(ssme_pf_handle h, const double* theta, int32_t n_theta, int32_t n_rows);

/* Back to t = 0 with the current parameters (a fresh model object in the reference). */
/* New random stream for the next evaluation; resets the filters.  The reference seeds every model object from the clock
 * (a fresh likelihood estimate per PMMH proposal, estimate_univ_svol.h:119); here the caller supplies the seed. */
int ssme_pf_set_seed(ssme_pf_handle h, uint64_t seed);
int ssme_pf_reset(ssme_pf_handle h);

/* One filter() call on every filter of the handle: BSFilter::filter(y_t) /
 * BSFilterWC::filter(y_t, z_t).  y: 1 value; z: 1 value or NULL.  logcondlike_out (R
 * values, nullable) receives getLogCondLike() of each filter.  With logcondlike_out = NULL the
 * step is only queued on the handle's stream (the swarm classes: ssme_pf_swarm_aggregate, which
 * follows, hands the data back); every later call on the handle is ordered behind it. */
int ssme_pf_step(ssme_pf_handle h, const double* y, const double* z, double* logcondlike_out);

/* The whole log_like_eval loop (estimate_univ_svol.h:121-127) for all R filters in one
 * call: reset, T steps, sum of log p(y_t | y_{1:t-1}).  loglik_out: R values. */
int ssme_pf_run_series(ssme_pf_handle h, const double* y, const double* z, int32_t T, double* loglik_out);

/* Per-step log conditional likelihoods of the last run_series: out[r*T + t]. */
int ssme_pf_get_per_step(ssme_pf_handle h, double* out, int32_t T);

/* Accumulated log-likelihood since the last reset: R values. */
int ssme_pf_get_loglik(ssme_pf_handle h, double* out);

/* E[h(x_t) | y_{1:t}] with the pre-resampling weights of the last step
 * (getExpectations(); twin liu_west_filter.h:1662-1683).  out: R values. */
int ssme_pf_get_expectations(ssme_pf_handle h, int32_t functional, double* out);
/* The same for n <= 4 built-in functionals at once (filter(y, z, fs) takes a VECTOR of functions, pswarm_filter.h:87):
 * one pass over the particles, one download.  out[i*R + r] = E[h_i] of filter r. */
int ssme_pf_get_expectations_multi(ssme_pf_handle h, const int32_t* functionals, int32_t n, double* out);
/* Swarm::update's aggregation over the R member filters of this handle (pswarm_filter.h:96-160,233-235: uniform
 * weights, i.e. plain means), reduced on the device: mean of the last step's log conditional likelihoods and mean of each
 * requested expectation (n may be 0).  One download of n + 1 doubles. */
int ssme_pf_swarm_aggregate(ssme_pf_handle h, const int32_t* functionals, int32_t n, double* mean_logcondlike /*1*/,
                            double* mean_expectations /*n*/);
/* The same as the reference computes it when its pool has num_threads workers: member i is dealt to thread i % num_threads
 * (thread_pool.h:443-447), every thread averages its members and the thread averages are averaged (pswarm_filter.h:96-160) --
 * the plain mean exactly when num_threads divides R, a slightly differently weighted mean otherwise (a thread with one member
 * fewer counts each of its members more).  num_threads <= 0 or >= R: ssme_pf_swarm_aggregate's plain mean / one member per thread. */
int ssme_pf_swarm_aggregate_threads(ssme_pf_handle h, const int32_t* functionals, int32_t n, int32_t num_threads,
                                    double* mean_logcondlike /*1*/, double* mean_expectations /*n*/);
/* Arbitrary host-side h (the reference's filt_func is a std::function, pswarm_filter.h:44,87-89): particles x (N,
 * nullable) and weights w (N) of one filter after the last step, w_j = exp(logw_j - max logw) in the 2^-41 fixed point
 * the resampler and ssme_pf_get_expectations use; the caller forms sum h(x_j) w_j / sum w_j.  Needs no debug mode. */
int ssme_pf_download_weights(ssme_pf_handle h, int32_t filter, double* x, double* w);

/* Replicate aggregation of thread_pool.h:263-268: log-mean-exp of the R log-likelihoods. */
int ssme_pf_log_mean_exp(ssme_pf_handle h, double* out);

/* Parity/debug: state of one filter after the last step.  Any pointer may be NULL.
 * x: N pre-resampling particles; logw: their log-weights (only kept in memory after
 * set_debug(flags & 2), or when resamp_sched > 1); cdf: N tile-local inclusive sums of the
 * fixed-point weights q_i = rne(exp(logw_i - max_tile) * 2^41) (exact integers < 2^53, carried in fp64 on the device);
 * ancestors: N indices used by the last step (requires set_debug(flags & 1)). */
int ssme_pf_download_state(ssme_pf_handle h, int32_t filter, double* x, double* logw, uint64_t* cdf,
                           uint32_t* ancestors);
/* The tile decomposition in use: particles per tile (2048, 1024 or 512) and tiles per filter. */
int ssme_pf_get_layout(ssme_pf_handle h, int32_t* tile_particles, int32_t* n_tiles);
/* max_logw: max log-weight of the last step; sum_q: exact integer sum of the rescaled tile sums;
 * tile_sums / tile_max: one integer weight sum and one max log-weight per tile;
 * rshift: the fixed-point exponent rg = 52 - ceil(log2(Npad)) of sum_q. */
int ssme_pf_download_scalars(ssme_pf_handle h, int32_t filter, double* max_logw, uint64_t* sum_q,
                             uint64_t* tile_sums, double* tile_max, int32_t* rshift);
/* flags: bit 0 = record ancestor indices, bit 1 = keep log-weights in memory (parity tests), bit 2 = compute level-2 (global
 * max, rescaled tile sums, their scan, every tile's source range) once per filter in its own launch instead of in every
 * workgroup; bit 3 = the opposite (in-kernel level-2, possible up to 2048 tiles).  Default: split above 1024 tiles
 * (N > 2^20), where it is faster; always split above 2048 tiles (N > 2^22; limit N <= 2^25).  Same results to the bit. */
int ssme_pf_set_debug(ssme_pf_handle h, int32_t flags);

/* Threads per 2048-particle tile of the step kernel: 256, 512 (default) or 1024.  Results do
 * not depend on it (the weight cdf is exact integer arithmetic). */
int ssme_pf_set_tuning(ssme_pf_handle h, int32_t threads_per_tile);

/* Filters of one tile (N <= 2048) run ssme_pf_run_series as ONE launch that loops over the series with the state in
 * LDS (default 1); 0 forces the tiled per-step kernel.  Results are bit-identical either way (parity tests). */
int ssme_pf_set_small_series(ssme_pf_handle h, int32_t enable);
/* Execution policy of run_series: 0 = eager launches, 1 = one hipGraph per series (default). */
int ssme_pf_set_graph_mode(ssme_pf_handle h, int32_t mode);

/* HIP-event time (ms) of the last run_series on the handle's stream (kernels only, the
 * 8*T-byte upload of y and the 8*R-byte download of the result excluded). */
int ssme_pf_last_elapsed_ms(ssme_pf_handle h, float* ms);

/* Measurement aid for bench.py: runs a T-step series eagerly (no graph) with a HIP event on the handle's
 * stream after every 32 launches of the step kernel (k_filter_step); returns the mean launch duration in
 * microseconds (mean_us_out[0]) and the launch count (launches_out[0]). */
int ssme_pf_profile_series(ssme_pf_handle h, const double* y, const double* z, int32_t T,
                           double* mean_us_out /*1*/, int32_t* launches_out /*1*/);

/* Device-side primitives exposed for bit-parity tests against the oracle. */
int ssme_pf_test_math(int32_t device, int32_t fn /*0 exp,1 log,2 sin2pi,3 cos2pi,4 sqrt,5 log (normal-only core),6 log of a uniform (table),7 exp (table form of the bootstrap filter),8 / 9 sin / cos of 2 pi k / 2^24 (in = k),10 log of a spacing uniform,11 sqrt of a positive normal*/,
                      const double* in, double* out, int64_t n);
int ssme_pf_test_philox(int32_t device, const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4);
/* q = rne(exp(in) * 2^shift) as uint64, n values */
int ssme_pf_test_quantize(int32_t device, const double* in, int32_t shift, uint64_t* out, int64_t n);
/* out = rint((double)tile_sum * exp(dm) * 2^shift): the cross-tile rescaling of tile sums */
int ssme_pf_test_rescale(int32_t device, const uint64_t* tile_sums, const double* dm, int32_t shift, uint64_t* out,
                         int64_t n);
/* exact inclusive scan of 2048 integers (< 2^53 in total) by one block of 256/512/1024 threads (fp64 DPP wave scans) */
int ssme_pf_test_block_scan(int32_t device, int32_t threads, const uint64_t* in2048, uint64_t* incl2048,
                            uint64_t* total);
/* measurement aid: `repeats` streaming copies of n doubles with 16-byte-per-lane accesses (counter calibration) */
int ssme_pf_test_copy(int32_t device, int64_t n_doubles, int32_t repeats);
/* n Gamma(shape) draws for tiles 0..n-1 at time t of filter `rep` */
int ssme_pf_test_gamma(int32_t device, uint64_t seed, uint32_t rep, int32_t t, double shape, int32_t n, double* out);

/* ============================================================================================
 * Particle-sharded filter: ONE filter of cfg->n_particles particles over `world` GPUs,
 * one process per GPU.  A tile = 2048 particles, B = ceil(n_particles / 2048) tiles, Bl = ceil(B / world): rank g owns tiles
 * [g Bl, min((g+1) Bl, B)) -- any n_particles up to 2^25 for which every rank owns at least one tile, (world-1) Bl < B (always
 * true from world (world-1) tiles on; SSME_ERR_UNSUPPORTED otherwise): the last rank may own fewer tiles and a ragged last
 * tile, and every per-rank layout (the gathered tile arrays: world x Bl entries of which the first B are tiles; buffers; halos)
 * has Bl rows per rank -- ssme_pf_shard_layout.  n_filters = 1; any resamp_sched: a step without a resampling draw exchanges
 * nothing but the tile sums and carries this rank's log-weights.  Per time step the host side (the C++ driver below, or
 * ssme_amd/sharded.py over torch.distributed) does:   all_gather of the tile sums / maxima  ->  ssme_pf_shard_plan (which source tiles each
 * rank's resampling touches)  ->  exchange of those tiles (cdf + particles)  ->  ssme_pf_shard_step.
 * The level-2 arithmetic, the RNG counters (global particle index) and the Gamma tables (global tile id) are those of
 * the unsharded filter, so a sharded run is bit-identical to ssme_pf_run_series with the same N and seed.
 * Buffers are the caller's device pointers; all launches go to the stream given to ssme_pf_set_stream.
 * Reference counterpart: none (the reference is single-process); the semantics reproduced are those of BSFilter::filter
 * driven by the log_like_eval loop (example/estimate_univ_svol.h:121-127), the decomposition is SURVEY.md section 8e's.
 * ============================================================================================ */
int ssme_pf_shard_create(const ssme_pf_config* cfg, int32_t rank, int32_t world, ssme_pf_handle* out);
/* out4 = { B (tiles of the filter), Bl (rows per rank in every layout), tiles this rank owns, particles this rank owns } */
int ssme_pf_shard_layout(ssme_pf_handle h, int32_t* out4);
/* Launch on the caller's HIP stream (hipStream_t as void*; NULL = the handle's own stream). */
int ssme_pf_set_stream(ssme_pf_handle h, void* hip_stream);
/* Uploads the series, draws the Gamma tables of all T steps (every rank holds all tiles' draws), resets the scalars. */
int ssme_pf_shard_prepare(ssme_pf_handle h, const double* y, const double* z, int32_t T);
/* Step t >= 1: from the gathered tile sums / maxima of step t-1 (device, world x Bl doubles each) the inclusive source-tile range
 * [lo, hi] of every rank: lo_hi_host[2*g], lo_hi_host[2*g+1].  Synchronises the stream. */
int ssme_pf_shard_plan(ssme_pf_handle h, const double* tsum_all, const double* tmax_all, int32_t t, int32_t* lo_hi_host);
/* One filter step on this rank's tiles.  x_win / cdf_win hold source tiles win_tile0 .. (at least) this rank's hi,
 * 2048 doubles per tile (ignored at t = 0); outputs: this rank's particles, integer cdf, tile sums and maxima.
 * anc_out (optional): global ancestor index of every output particle.  Accounts log p(y_{t-1} | .) on every rank. */
int ssme_pf_shard_step(ssme_pf_handle h, int32_t t, const double* x_win, const double* cdf_win, int32_t win_tile0,
                       const double* tsum_all, const double* tmax_all, double* x_out, double* cdf_out, double* tsum_out,
                       double* tmax_out, uint32_t* anc_out);
/* Accounts the log conditional likelihood of the last step t from its gathered tile arrays; then
 * ssme_pf_get_loglik / ssme_pf_get_per_step return the same values on every rank. */
int ssme_pf_shard_finalize(ssme_pf_handle h, int32_t t, const double* tsum_all, const double* tmax_all);

/* ---- the same loop in C++ over RCCL: a C or C++ caller of ssme needs no Python to use several GPUs ------------------------
 * RCCL is resolved at run time from what the process has loaded (dlsym; `librccl.so` from the loader path otherwise); this
 * library does not link against it.  comm = an ncclComm_t whose rank / size equal the handle's (pass your own, or make
 * one: rank 0 calls ssme_shard_comm_get_unique_id, ships the 128 bytes to the other ranks by any means, every rank calls
 * ssme_shard_comm_init).
 * ssme_pf_shard_run_series runs the whole series on the handle's stream.  mode 0: fixed-halo exchange with the two
 * neighbouring ranks and no host synchronisation inside the time loop; a device flag per rank records whether one of its
 * resampling windows ever left the halo, the flags are reduced over the ranks after the loop (ncclAllReduce, max), and if any is
 * set EVERY rank runs the series again on the exact path; mode 1: fixed halo only
 * (SSME_ERR_STATE if a window left it); mode 2: exact path (the plan is downloaded every step, exactly the planned tiles
 * travel between any two ranks).  Results are bit-identical to the unsharded filter on every path.  loglik_out: 1 value,
 * identical on every rank.  Buffers are owned by the handle. */
int ssme_shard_comm_get_unique_id(void* id128 /*128 bytes*/);
int ssme_shard_comm_init(const void* id128, int32_t rank, int32_t world, int32_t device, void** comm_out);
int ssme_shard_comm_destroy(void* comm);
int ssme_pf_shard_run_series(ssme_pf_handle h, void* nccl_comm, const double* y, const double* z, int32_t T, int32_t mode,
                             double* loglik_out);
/* after ssme_pf_shard_run_series: this rank's particles (ssme_pf_shard_layout's out4[3]) and integer cdf (nullable), the path the last series
 * took (1 fixed halo, 2 exact) and the number of tiles this rank received from other ranks */
int ssme_pf_shard_download(ssme_pf_handle h, double* x_local, uint64_t* cdf_local, int32_t* path, int64_t* exchanged_tiles);
/* after ssme_pf_shard_run_series, of its last FIXED-HALO pass: out4[0] = 1 if a resampling window left the halo on ANY rank (the
 * per-rank flags reduced by one ncclAllReduce(max) after the time loop -- every rank reads the same value, so every rank takes
 * the same fallback decision), out4[1] = this rank's own flag, out4[2] / out4[3] = widest reach left / right of the own tiles
 * where the split level-2 planned the exchange (0 otherwise) */
int ssme_pf_shard_stats(ssme_pf_handle h, int32_t* out4);

/* ============================================================================================
 * Liu-West filter: LWFilterWithCovs<nparts,1,1,1,4,float_t>::filter (include/ssme/liu_west_filter.h:971-1159)
 * with the model of svol_lw_1_par (test/test_liu_west.cpp:22-157): parameters (phi, mu, sigma, rho), one
 * transform per parameter (parameters.h:27 enum order: 0 null, 1 twice_fisher, 2 logit, 3 log), uniform priors,
 * shrinkage a = (3 delta - 1) / (2 delta), resampling of states and parameters every step (:91-145).
 * ============================================================================================ */
typedef struct ssme_lw_s* ssme_lw_handle;
typedef struct ssme_lw_config {
    int32_t  n_particles;       /* N per filter                                                        */
    int32_t  n_filters;         /* independent filters in this handle                                   */
    uint64_t seed;
    int32_t  device;
    uint32_t first_filter_id;
    double   delta;             /* discount factor, ctor argument `delta` (liu_west_filter.h:954-960)    */
    int32_t  transforms[4];     /* ctor argument `transforms`; svol_lw_1_par: logit, null, log, twice_fisher */
    double   prior_lo[4];       /* paramPriorSamp(): theta_d ~ U(lo_d, hi_d), test_liu_west.cpp:140-150  */
    double   prior_hi[4];
    int32_t  form;              /* 0: auxiliary-particle form, LWFilterWithCovs::filter (liu_west_filter.h:971-1159, model
                                   svol_lw_1_par); 1: SISR form, LWFilter2WithCovs::filter (:2191-2343, model svol_lw_2_par,
                                   test/test_liu_west.cpp:214-358)                                           */
    int32_t  resamp_sched;      /* m_rs / m_resampSched: resample when (t + 1) % m_rs == 0; 0 or 1 = every step (default) */
} ssme_lw_config;

int ssme_lw_create(const ssme_lw_config* cfg, ssme_lw_handle* out);
int ssme_lw_destroy(ssme_lw_handle h);
int ssme_lw_reset(ssme_lw_handle h);
/* filter(y_t, z_t): one step on every filter; logcondlike_out (R values) = getLogCondLike() */
int ssme_lw_step(ssme_lw_handle h, const double* y, const double* z, double* logcondlike_out);
/* T steps; loglik_out (R values) = sum of the log conditional likelihoods */
int ssme_lw_run_series(ssme_lw_handle h, const double* y, const double* z, int32_t T, double* loglik_out);
int ssme_lw_get_per_step(ssme_lw_handle h, double* out, int32_t T);
/* weighted means of the untransformed parameters under the last step's weights: out[r*4 + d] */
int ssme_lw_get_param_means(ssme_lw_handle h, double* out);
/* E[h | y_{1:t}] under the last step's (pre-resampling) weights for built-in functionals (the reference takes
 * std::function h(x, z, theta), liu_west_filter.h:1054-1075 / :2267-2290): ids 0-3 = SSME_H_X, SSME_H_X2, SSME_H_VOL,
 * SSME_H_CONST42 of the state; 4-7 = the untransformed parameters phi, mu, sigma, rho.  out[i*R + r]. */
int ssme_lw_get_expectations(ssme_lw_handle h, const int32_t* functionals, int32_t n, double* out);
/* Arbitrary host-side h: particles x (N), UNTRANSFORMED parameters theta[d*N + i] (4N) and weights w (N) of one filter
 * after the last step, w_i = exp(logw_i - max logw) in the resampler's fixed point; any pointer but w may be NULL. */
int ssme_lw_download_weights(ssme_lw_handle h, int32_t filter, double* x, double* theta_untrans, double* w);
/* parity/debug: particles, transformed parameters theta[d*N + i] (getParamSamples()), k indices and resampling
 * ancestors of the last step (after set_debug(1)), theta-bar and the Cholesky factor of (1 - a^2) V (row-major 4x4) */
int ssme_lw_download_state(ssme_lw_handle h, int32_t filter, double* x, double* theta, uint32_t* kidx, uint32_t* ancestors,
                           double* thetabar, double* chol);
int ssme_lw_set_debug(ssme_lw_handle h, int32_t flags);
int ssme_lw_last_elapsed_ms(ssme_lw_handle h, float* ms);
const char* ssme_lw_last_error(ssme_lw_handle h);

/* ---- particle-sharded Liu-West filter: ONE filter of cfg->n_particles particles over `world` GPUs (BASELINE.json configs[4]).
 * Rank g owns tiles [g Bl, min((g+1) Bl, B)), Bl = ceil(B / world), as ssme_pf_shard_create lays a filter out (any n_particles
 * up to 2^25 with (world-1) Bl < B; ssme_lw_shard_layout); n_filters = 1; any resampling schedule (a step
 * without a resampling draw, t % resamp_sched != 0: ssme_lw_shard_stage1 is given this rank's OWN rows, win_tile0 = its first tile,
 * and nothing is exchanged for it; the handle carries the second-stage log-weights); both forms -- the SISR form, form = 1, has no k draw, so its stage 2 reads this rank's own stage-1 outputs and a step
 * has ONE window exchange instead of two).  Per step the
 * host side (ssme_amd/sharded.py, ShardedLiuWest) gathers the tile sums / maxima of the second-stage weights, plans and
 * exchanges windows of (cdfB, x, theta) for the resampling draw (stage 1), gathers the first-stage tile sums / maxima and
 * the 14 moment partials per tile, runs ssme_lw_shard_mid on every rank (theta-bar, Cholesky factor: the moment sums are
 * added in tile order, so every rank gets the unsharded filter's bits), plans and exchanges windows of (cdfA, lw1, x, theta)
 * for the k draw (stage 2) -- call ssme_lw_shard_plan(which = 1) BEFORE ssme_lw_shard_mid: above 1024 tiles the plan also
 * provides the first-stage (m, S) that mid turns into the log-sum-exp.  theta buffers are 4 planes: [4][tiles * 2048].  Bit-identical to ssme_lw_run_series. */
int ssme_lw_shard_create(const ssme_lw_config* cfg, int32_t rank, int32_t world, ssme_lw_handle* out);
int ssme_lw_shard_layout(ssme_lw_handle h, int32_t* out4);      /* as ssme_pf_shard_layout */
int ssme_lw_set_stream(ssme_lw_handle h, void* hip_stream);
/* theta OUTPUT buffers are 4 planes of `tiles` rows each (default Bl; larger when the caller keeps halo rows around
 * its own tiles).  In the stage calls `win_tiles` is the same thing for the theta SOURCE window: rows per plane. */
int ssme_lw_shard_set_plane_tiles(ssme_lw_handle h, int32_t tiles);
int ssme_lw_shard_prepare(ssme_lw_handle h, const double* y, const double* z, int32_t T);
/* t = 0 on this rank's tiles: prior draws, q1Samp, first weights */
int ssme_lw_shard_init(ssme_lw_handle h, double* xB, double* thB, double* cdfB, double* tsumB, double* tmaxB);
/* which = 0: source-tile ranges of the resampling draw (from the gathered second-stage tile sums), 1: of the k draw
 * (from the gathered first-stage tile sums); lo_hi_host[2g], [2g+1] per rank.  Synchronises the stream. */
int ssme_lw_shard_plan(ssme_lw_handle h, int32_t which, int32_t t, const double* tsum_all, const double* tmax_all,
                       int32_t* lo_hi_host);
int ssme_lw_shard_stage1(ssme_lw_handle h, int32_t t, int32_t win_tile0, int32_t win_tiles, const double* w_xB, const double* w_thB,
                         const double* w_cdfB, const double* tsumB_all, const double* tmaxB_all, double* xr, double* thr, double* lw1,
                         double* cdfA, double* tsumA, double* tmaxA, double* mom /*[tiles][16]*/, uint32_t* anc);
int ssme_lw_shard_mid(ssme_lw_handle h, int32_t t, const double* tsumA_all, const double* tmaxA_all, const double* mom_all);
int ssme_lw_shard_stage2(ssme_lw_handle h, int32_t t, int32_t win_tile0, int32_t win_tiles, const double* w_xr, const double* w_thr,
                         const double* w_lw1, const double* w_cdfA, const double* tsumA_all, const double* tmaxA_all, double* xB,
                         double* thB, double* cdfB, double* tsumB, double* tmaxB, uint32_t* kidx);
/* accounts the last step t from its gathered second-stage tile sums; then ssme_lw_get_loglik / get_per_step */
int ssme_lw_shard_finalize(ssme_lw_handle h, int32_t t, const double* tsumB_all, const double* tmaxB_all);
/* The same loop in C++ over RCCL (see ssme_pf_shard_run_series; comm from ssme_shard_comm_init or the caller's own
 * ncclComm_t): fixed-halo exchange with the two neighbouring ranks, no host synchronisation inside the time loop.  The
 * stage kernels verify their own source tiles against the exchanged window; if one ever left it ON ANY RANK (flags reduced by
 * ncclAllReduce after the loop) the call returns SSME_ERR_STATE ON EVERY RANK and the caller runs the exact host-planned loop over the step-wise entry points above.  loglik_out: 1. */
int ssme_lw_shard_run_series(ssme_lw_handle h, void* nccl_comm, const double* y, const double* z, int32_t T, double* loglik_out);
/* after ssme_lw_shard_run_series: this rank's n particles (ssme_lw_shard_layout), transformed parameters theta[d * n + i], tiles received */
int ssme_lw_shard_download(ssme_lw_handle h, double* x_local, double* theta_local, int64_t* exchanged_tiles);
/* as ssme_pf_shard_stats: out4[0] the reduced flag (SSME_ERR_STATE was returned on EVERY rank if it is set), out4[1] this rank's own */
int ssme_lw_shard_stats(ssme_lw_handle h, int32_t* out4);
int ssme_lw_get_loglik(ssme_lw_handle h, double* out);


const char* ssme_pf_strerror(int status);
const char* ssme_pf_last_error(ssme_pf_handle h);
int ssme_pf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSME_PF_H */
