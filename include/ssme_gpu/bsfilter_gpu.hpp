// bsfilter_gpu.hpp -- header-only C++ adaptor: the caller-visible surface of ssme's bootstrap-filter
// models over the C ABI of include/ssme_pf.h.
//
// Reference surface reproduced (files under /root/reference):
//   svol_bs<nparts,dimx,dimy,resampT,float_t>               example/univ_svol_bootstrap_filter.h:17-61
//       ctor (phi, beta, sigma); ctor from param::pack<float_t,3> in the order beta, phi, ss (sigma = sqrt(ss))
//       void filter(const osv&);  float_t getLogCondLike() const;          example/estimate_univ_svol.h:124-125
//   svol_leverage<nparts,resampT,float_t> : BSFilterWC<...>                test/test_pswarm.cpp:32-76
//       default ctor + copy-assign (Swarm requires both, pswarm_filter.h:292), ctor (phi, mu, sigma, rho, dte)
//       void filter(const osv&, const cvsv&, const std::vector<func>&); getExpectations(); getLogCondLike()
//                                                                          include/ssme/pswarm_filter.h:380-388
//   typedefs float_type, dynamic_matrix, func                              include/ssme/pswarm_filter.h:29,41,44
// The classes are templated on the pack / vector / matrix types (duck typing), so they compile with the reference's
// Eigen-based param::pack, Eigen vectors and Eigen dynamic matrices as well as with plain stand-ins (tests/cpp).
// svol_leverage_gpu satisfies what the UNMODIFIED Swarm / SwarmWithCovs templates ask of ModType
// (pswarm_filter.h:29-60,86-92,272-304,380-388): float_type / dynamic_matrix / func typedefs with func a
// std::function returning a dynamic matrix, default ctor + copy-assign, filter(y, z, const std::vector<func>&),
// std::vector<dynamic_matrix> getExpectations(), and -- through the optional `Base` template parameter -- inheritance
// from pf::bases::pf_withcov_base<float_t,dimy,dimx,dimcov>, so that the static_assert at :352 holds as written.
// Host std::function callbacks cannot run on the device.  Each h in fs is therefore PROBED on the host at six state
// values: a function that is constant, x, x^2 or exp(x/2) there is served by the device functionals (SSME_H_*);
// anything else is evaluated on the host over the downloaded particles and weights (ssme_pf_download_weights) --
// correct for every h, fast for the common ones.  gpu_options::probe_functionals = false forces the host path.
// Every model object draws its own random stream (the reference clock-seeds each object): the filter id defaults to a
// process-wide counter.  Errors: std::invalid_argument / std::runtime_error, as the reference throws; NaN/-inf are values.
#ifndef SSME_GPU_BSFILTER_GPU_HPP
#define SSME_GPU_BSFILTER_GPU_HPP

#include <atomic>
#include <cmath>
#include <cstddef>
#include <functional>
#include <fstream>
#include <limits>
#include <sstream>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ssme_pf.h"

namespace ssme_gpu {

inline void check(int status, ssme_pf_handle h = nullptr) {
    if (status == SSME_OK) return;
    std::string msg = ssme_pf_strerror(status);
    if (h && status == SSME_ERR_HIP) msg += std::string(" (") + ssme_pf_last_error(h) + ")";
    if (status == SSME_ERR_INVALID_ARG || status == SSME_ERR_LENGTH) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// RAII owner of one C-ABI handle holding R filters.
class handle {
public:
    handle() = default;
    handle(int model, int nparts, int nfilters, std::uint64_t seed, int resampler, int rs, int device, unsigned first_id,
           int dtype = SSME_F64) {
        ssme_pf_config c{};
        c.model = model; c.n_particles = nparts; c.n_filters = nfilters; c.dtype = dtype; c.resampler = resampler;
        c.resamp_sched = rs; c.seed = seed; c.device = device; c.first_filter_id = first_id;
        ssme_pf_handle raw = nullptr;
        check(ssme_pf_create(&c, &raw));
        h_ = std::shared_ptr<ssme_pf_s>(raw, [](ssme_pf_handle p) { if (p) ssme_pf_destroy(p); });
    }
    ssme_pf_handle get() const { return h_.get(); }
    explicit operator bool() const { return (bool)h_; }
private:
    std::shared_ptr<ssme_pf_s> h_;   // copy-assignable like the reference's models; copies share device state
};

struct gpu_options {
    std::uint64_t seed = 0;          // the reference seeds from the clock; here the stream is reproducible
    int resampler = SSME_RESAMP_MULTINOMIAL;
    int resamp_sched = 1;
    int device = 0;
    bool probe_functionals = true;   // recognise constant / x / x^2 / exp(x/2) functionals and run them on the device
};

// Filter id of a model object constructed without one: a process-wide counter, so that every object has its own random
// stream under one seed (the reference seeds every sampler of every model object from the clock, liu_west_filter.h:75-76).
constexpr unsigned auto_filter_id = 0xffffffffu;
namespace detail {
struct no_base {};
// float_t = float (the shipped example, example/main.cpp:13) -> SSME_F32: float at the boundary (ssme_pf.h)
template <typename float_t> constexpr int dtype_of() { return sizeof(float_t) == sizeof(float) ? SSME_F32 : SSME_F64; }
inline unsigned resolve_filter_id(unsigned id) {
    static std::atomic<unsigned> next{0};
    return id == auto_filter_id ? next.fetch_add(1) : id;
}
}  // namespace detail

// ---- svol_bs ---------------------------------------------------------------------------------------------
template <std::size_t nparts, typename float_t = double>
class svol_bs_gpu {
public:
    using float_type = float_t;
    svol_bs_gpu(const float_t& phi, const float_t& beta, const float_t& sigma, gpu_options o = gpu_options(),
                unsigned filter_id = auto_filter_id)
        : h_(SSME_MODEL_SVOL, (int)nparts, 1, o.seed, o.resampler, o.resamp_sched, o.device, detail::resolve_filter_id(filter_id),
             detail::dtype_of<float_t>()) {
        const double th[3] = {(double)beta, (double)phi, (double)sigma};
        check(ssme_pf_set_params(h_.get(), th, 3, 1), h_.get());
    }
    // ctor from a param::pack: order beta, phi, ss (univ_svol_bootstrap_filter.h:55-61)
    template <typename Pack>
    explicit svol_bs_gpu(const Pack& pp, gpu_options o = gpu_options(), unsigned filter_id = auto_filter_id)
        : svol_bs_gpu((float_t)pp.get_untrans_params(1, 1)(0), (float_t)pp.get_untrans_params(0, 0)(0),
                      (float_t)std::sqrt((double)pp.get_untrans_params(2, 2)(0)), o, filter_id) {}

    template <typename Osv>
    void filter(const Osv& yt) {
        const double y = (double)yt(0);
        double out = 0.0;
        check(ssme_pf_step(h_.get(), &y, nullptr, &out), h_.get());
        last_ = (float_t)out;
    }
    float_t getLogCondLike() const { return last_; }
    ssme_pf_handle native() const { return h_.get(); }

private:
    handle h_;
    float_t last_ = 0;
};

// ---- expectations of host-side functionals ---------------------------------------------------------------------
// Shared by the swarm-member models below.  h: any callable Mat(const Ssv&) (the covariate, if any, already bound).
namespace detail {
template <typename float_t, typename Mat, typename Ssv>
struct functional_engine {
    // Which device functional reproduces h?  Probed at six states; exact agreement required (exp(x/2): 4 ulp).
    // Returns SSME_H_* (and the factor to apply to the device value) or -1 = evaluate on the host.
    template <typename H>
    static int classify(const H& h, double* scale) {
        static const double probes[6] = {-1.7, -0.3125, 0.0, 0.5625, 1.9, 3.25};
        bool is_const = true, is_x = true, is_x2 = true, is_vol = true;
        float_t c0 = 0;
        for (int p = 0; p < 6; ++p) {
            Ssv xv;
            xv(0) = (float_t)probes[p];
            const Mat m = h(xv);
            if (m.rows() != 1 || m.cols() != 1) return -1;
            const float_t v = m(0, 0), xx = xv(0);
            if (p == 0) c0 = v;
            is_const = is_const && (v == c0);
            is_x = is_x && (v == xx);
            is_x2 = is_x2 && (v == xx * xx);
            const float_t e = (float_t)std::exp((float_t)0.5 * xx);
            is_vol = is_vol && (std::fabs(v - e) <= 4 * std::numeric_limits<float_t>::epsilon() * e);
        }
        *scale = 1.0;
        if (is_const) { *scale = (double)c0 / 42.0; return SSME_H_CONST42; }   // E[c] = c E[42] / 42 (a NaN filter stays NaN)
        if (is_x) return SSME_H_X;
        if (is_x2) return SSME_H_X2;
        if (is_vol) return SSME_H_VOL;
        return -1;
    }
    // sum_i h(x_i) w_i / sum_i w_i on the host, any matrix shape (twin liu_west_filter.h:1662-1683)
    template <typename H>
    static Mat host_expectation(const H& h, const std::vector<double>& x, const std::vector<double>& w) {
        std::vector<double> acc;
        long rows = 0, cols = 0;
        double wsum = 0.0;
        for (std::size_t i = 0; i < x.size(); ++i) {
            Ssv xv;
            xv(0) = (float_t)x[i];
            const Mat hv = h(xv);
            if (i == 0) { rows = (long)hv.rows(); cols = (long)hv.cols(); acc.assign((std::size_t)(rows * cols), 0.0); }
            for (long r = 0; r < rows; ++r)
                for (long c = 0; c < cols; ++c) acc[(std::size_t)(r * cols + c)] += (double)hv(r, c) * w[i];
            wsum += w[i];
        }
        Mat m(rows, cols);
        for (long r = 0; r < rows; ++r)
            for (long c = 0; c < cols; ++c) m(r, c) = (float_t)(acc[(std::size_t)(r * cols + c)] / wsum);
        return m;
    }
    // E[h_i] for every h of `hs` after the last step of single-filter handle `hd`: device functionals in one pass,
    // everything else on the host over ONE download of (x, weights).
    template <typename H>
    static std::vector<Mat> expectations(ssme_pf_handle hd, const std::vector<H>& hs, std::size_t nparts, bool probe) {
        std::vector<Mat> out;
        if (hs.empty()) return out;
        std::vector<int> kind(hs.size(), -1);
        std::vector<double> scale(hs.size(), 1.0);
        std::vector<int32_t> ids;
        for (std::size_t i = 0; i < hs.size(); ++i) {
            if (probe) kind[i] = classify(hs[i], &scale[i]);
            if (kind[i] >= 0 && ids.size() == 4) kind[i] = -1;           // more than 4 device functionals: the rest on the host
            if (kind[i] >= 0) ids.push_back(kind[i]);
        }
        std::vector<double> dev(ids.size());
        if (!ids.empty()) check(ssme_pf_get_expectations_multi(hd, ids.data(), (int32_t)ids.size(), dev.data()), hd);
        std::vector<double> x, w;
        std::size_t d = 0;
        for (std::size_t i = 0; i < hs.size(); ++i) {
            if (kind[i] >= 0) {
                Mat m(1, 1);
                m(0, 0) = (float_t)(dev[d++] * scale[i]);
                out.push_back(m);
            } else {
                if (w.empty()) {
                    x.resize(nparts); w.resize(nparts);
                    check(ssme_pf_download_weights(hd, 0, x.data(), w.data()), hd);
                }
                out.push_back(host_expectation(hs[i], x, w));
            }
        }
        return out;
    }
};
}  // namespace detail

// ---- svol_leverage (BSFilterWC): a ModType for the unmodified SwarmWithCovs ------------------------------------------
// Template parameters: Mat = the model's dynamic_matrix (Eigen::Matrix<float_t,-1,-1> in the reference), Osv / Ssv / Cvsv =
// observation / state / covariate vectors (Eigen::Matrix<float_t,1,1>), Base = pf::bases::pf_withcov_base<float_t,1,1,1>
// when the pf headers are present (its pure virtuals filter / getLogCondLike are overridden by the members below
// [pf-recollection: pf_base.h]); detail::no_base otherwise.  Needs of the types: Mat(rows, cols), rows(), cols(),
// operator()(i, j); vectors: default ctor, operator()(i).
template <std::size_t nparts, typename float_t, typename Mat, typename Osv, typename Ssv, typename Cvsv,
          typename Base = detail::no_base>
class svol_leverage_gpu : public Base {
public:
    using float_type = float_t;
    using dynamic_matrix = Mat;
    using func = std::function<const Mat(const Ssv&, const Cvsv&)>;     // pf_withcov_base::func; what Swarm binds (:272-275)
    svol_leverage_gpu() = default;    // Swarm default-constructs its array of models (pswarm_filter.h:71,367)
    svol_leverage_gpu(const float_t& phi, const float_t& mu, const float_t& sigma, const float_t& rho, unsigned /*dte*/ = 0,
                      gpu_options o = gpu_options(), unsigned filter_id = auto_filter_id)
        : h_(SSME_MODEL_SVOL_LEVERAGE, (int)nparts, 1, o.seed, o.resampler, o.resamp_sched, o.device,
             detail::resolve_filter_id(filter_id), detail::dtype_of<float_t>()), probe_(o.probe_functionals) {
        const double th[4] = {(double)phi, (double)mu, (double)sigma, (double)rho};
        check(ssme_pf_set_params(h_.get(), th, 4, 1), h_.get());
    }
    // BSFilterWC::filter(y_t, z_t, fs) as SwarmWithCovs::comp_func calls it (pswarm_filter.h:383)
    void filter(const Osv& yt, const Cvsv& zt, const std::vector<func>& fs = std::vector<func>()) {
        if (!h_) throw std::runtime_error("model not constructed");
        const double y = (double)yt(0), z = (double)zt(0);
        double out = 0.0;
        check(ssme_pf_step(h_.get(), &y, &z, &out), h_.get());
        last_ = (float_t)out;
        std::vector<std::function<const Mat(const Ssv&)>> hs;
        for (const func& f : fs) hs.push_back([&f, &zt](const Ssv& xv) { return f(xv, zt); });
        expectations_ = detail::functional_engine<float_t, Mat, Ssv>::expectations(h_.get(), hs, nparts, probe_);
    }
    float_t getLogCondLike() const { return last_; }
    std::vector<Mat> getExpectations() const { return expectations_; }   // E[h(x_t) | y_{1:t}], pre-resampling weights
    ssme_pf_handle native() const { return h_.get(); }

private:
    handle h_;
    bool probe_ = true;
    float_t last_ = 0;
    std::vector<Mat> expectations_;
};

// ---- svol_bs as a ModType for the unmodified Swarm (no covariates; pswarm_filter.h:23-320, comp_func :86-92) -----------
// Base = pf::bases::pf_base<float_t,1,1> when pf is present.  Same model as svol_bs_gpu (ctor order phi, beta, sigma).
template <std::size_t nparts, typename float_t, typename Mat, typename Osv, typename Ssv, typename Base = detail::no_base>
class svol_bs_member_gpu : public Base {
public:
    using float_type = float_t;
    using dynamic_matrix = Mat;
    using func = std::function<const Mat(const Ssv&)>;                   // pf_base::func
    svol_bs_member_gpu() = default;
    svol_bs_member_gpu(const float_t& phi, const float_t& beta, const float_t& sigma, gpu_options o = gpu_options(),
                       unsigned filter_id = auto_filter_id)
        : h_(SSME_MODEL_SVOL, (int)nparts, 1, o.seed, o.resampler, o.resamp_sched, o.device, detail::resolve_filter_id(filter_id),
             detail::dtype_of<float_t>()),
          probe_(o.probe_functionals) {
        const double th[3] = {(double)beta, (double)phi, (double)sigma};
        check(ssme_pf_set_params(h_.get(), th, 3, 1), h_.get());
    }
    void filter(const Osv& yt, const std::vector<func>& fs = std::vector<func>()) {
        if (!h_) throw std::runtime_error("model not constructed");
        const double y = (double)yt(0);
        double out = 0.0;
        check(ssme_pf_step(h_.get(), &y, nullptr, &out), h_.get());
        last_ = (float_t)out;
        expectations_ = detail::functional_engine<float_t, Mat, Ssv>::expectations(h_.get(), fs, nparts, probe_);
    }
    float_t getLogCondLike() const { return last_; }
    std::vector<Mat> getExpectations() const { return expectations_; }
    ssme_pf_handle native() const { return h_.get(); }

private:
    handle h_;
    bool probe_ = true;
    float_t last_ = 0;
    std::vector<Mat> expectations_;
};

// ---- log_like_eval with replicate batching -----------------------------------------------------------------
// example/estimate_univ_svol.h:108-131 + thread_pool's num_pfilters replicates (thread_pool.h:189-215,263-268)
// in ONE call: R filters on the device, whole series, log-mean-exp.  `data` is any container of vectors with
// operator()(0) (std::vector<Eigen::Matrix<float_t,1,1>> in the reference).
template <typename Pack, typename Data>
double log_like_eval_gpu(const Pack& theta, const Data& data, int nparts, int num_pfilters, gpu_options o = gpu_options()) {
    if (data.empty()) throw std::length_error("can't read in data\n");   // estimate_univ_svol.h:112-113
    std::vector<double> y(data.size());
    for (std::size_t i = 0; i < data.size(); ++i) y[i] = (double)data[i](0);
    handle h(SSME_MODEL_SVOL, nparts, num_pfilters, o.seed, o.resampler, o.resamp_sched, o.device, 0);
    const double th[3] = {(double)theta.get_untrans_params(0, 0)(0), (double)theta.get_untrans_params(1, 1)(0),
                          std::sqrt((double)theta.get_untrans_params(2, 2)(0))};
    check(ssme_pf_set_params(h.get(), th, 3, 1), h.get());
    std::vector<double> ll((std::size_t)num_pfilters);
    check(ssme_pf_run_series(h.get(), y.data(), nullptr, (int)y.size(), ll.data()), h.get());
    double out = 0.0;
    check(ssme_pf_log_mean_exp(h.get(), &out), h.get());
    return out;
}

// ---- persistent evaluator: one handle, one captured graph, a fresh random stream per call --------------------------
// The PMMH caller evaluates log_like_eval once per MCMC iteration (ada_pmmh_mvn.h:344,363 through thread_pool::work).
// Constructing a model per evaluation (estimate_univ_svol.h:119) costs device allocations and a graph capture; this
// object keeps the handle, so an evaluation is: upload theta, new seed, replay the graph, log-mean-exp.
class svol_log_like_evaluator {
public:
    template <typename Data>
    svol_log_like_evaluator(const Data& data, int nparts, int num_pfilters, gpu_options o = gpu_options())
        : h_(SSME_MODEL_SVOL, nparts, num_pfilters, o.seed, o.resampler, o.resamp_sched, o.device, 0), ll_((std::size_t)num_pfilters) {
        if (data.empty()) throw std::length_error("can't read in data\n");
        y_.resize(data.size());
        for (std::size_t i = 0; i < data.size(); ++i) y_[i] = (double)data[i](0);
    }
    // theta in the reference's pack order: beta, phi, ss (sigma = sqrt(ss))
    template <typename Pack>
    double operator()(const Pack& theta, std::uint64_t seed) {
        const double th[3] = {(double)theta.get_untrans_params(0, 0)(0), (double)theta.get_untrans_params(1, 1)(0),
                              std::sqrt((double)theta.get_untrans_params(2, 2)(0))};
        check(ssme_pf_set_seed(h_.get(), seed), h_.get());
        check(ssme_pf_set_params(h_.get(), th, 3, 1), h_.get());
        check(ssme_pf_run_series(h_.get(), y_.data(), nullptr, (int)y_.size(), ll_.data()), h_.get());
        double out = 0.0;
        check(ssme_pf_log_mean_exp(h_.get(), &out), h_.get());
        return out;
    }
    float device_ms() const { float ms = 0; ssme_pf_last_elapsed_ms(h_.get(), &ms); return ms; }
    ssme_pf_handle native() const { return h_.get(); }

private:
    handle h_;
    std::vector<double> y_, ll_;
};

// ---- svol_lw_1_par / svol_lw_2_par (Liu-West filters) ------------------------------------------------------------
// test/test_liu_west.cpp:22-157: ctor (delta, phi_l, phi_u, mu_l, mu_u, sig_l, sig_u, rho_l, rho_u[, dte]);
// filter(y, z[, fs]), getLogCondLike(), getExpectations() (liu_west_filter.h:971-1159).  Transforms as svol_lw_1_par passes
// them to its base: logit, null, log, twice_fisher (test_liu_west.cpp:70).
// FORM 0 = auxiliary-particle form (LWFilterWithCovs, svol_lw_1_par); 1 = SISR form (LWFilter2WithCovs::filter,
// liu_west_filter.h:2191-2343, svol_lw_2_par of test/test_liu_west.cpp:214-358).  gpu_options::resamp_sched = m_rs.
// filter(y[, fs]) without a covariate is the call of the no-covariate LWFilter / LWFilter2 (:238, :1447): the same model
// with the covariate term at zero.
// Functionals: the reference's std::function<const Mat(const ssv&, const csv&, const psv&)> (no-covariate forms:
// (const ssv&, const psv&)) with the UNTRANSFORMED parameters (:1054-1075, :2267-2290).  Each h is probed on the host; one
// that is constant, x, x^2, exp(x/2) or one of the four parameters there runs on the device (ssme_lw_get_expectations),
// anything else is summed on the host over one download of (x, theta, weights).  Mat = the caller's dynamic matrix
// (Eigen::Matrix<float_t,-1,-1> in the reference); detail::small_matrix when none is given.
namespace detail {
template <typename T>
class small_matrix {                                  // the least a dynamic matrix must offer here
public:
    small_matrix() = default;
    small_matrix(long r, long c) : r_(r), c_(c), v_((std::size_t)(r * c), T(0)) {}
    long rows() const { return r_; }
    long cols() const { return c_; }
    T& operator()(long i, long j) { return v_[(std::size_t)(i * c_ + j)]; }
    const T& operator()(long i, long j) const { return v_[(std::size_t)(i * c_ + j)]; }
private:
    long r_ = 0, c_ = 0;
    std::vector<T> v_;
};
// argument types of the two functional signatures
template <typename F> struct lw_func_traits;
template <typename M, typename S, typename C, typename P>
struct lw_func_traits<std::function<const M(const S&, const C&, const P&)>> {
    using Mat = M; using Ssv = S; using Psv = P;
    static M call(const std::function<const M(const S&, const C&, const P&)>& f, const S& x, double z, const P& p) {
        C zv;
        zv(0) = z;
        return f(x, zv, p);
    }
};
template <typename M, typename S, typename P>
struct lw_func_traits<std::function<const M(const S&, const P&)>> {
    using Mat = M; using Ssv = S; using Psv = P;
    static M call(const std::function<const M(const S&, const P&)>& f, const S& x, double, const P& p) { return f(x, p); }
};
}  // namespace detail

template <std::size_t nparts, typename float_t = double, int FORM = 0, typename Mat = detail::small_matrix<float_t>>
class svol_lw_1_par_gpu {
public:
    using float_type = float_t;
    using dynamic_matrix = Mat;
    svol_lw_1_par_gpu(const float_t& delta, const float_t& phi_l, const float_t& phi_u, const float_t& mu_l, const float_t& mu_u,
                      const float_t& sig_l, const float_t& sig_u, const float_t& rho_l, const float_t& rho_u, unsigned /*dte*/ = 0,
                      gpu_options o = gpu_options(), unsigned filter_id = auto_filter_id) : probe_(o.probe_functionals) {
        ssme_lw_config c{};
        c.n_particles = (int)nparts; c.n_filters = 1; c.seed = o.seed; c.device = o.device;
        c.first_filter_id = detail::resolve_filter_id(filter_id);
        c.form = FORM; c.resamp_sched = o.resamp_sched;
        c.delta = (double)delta;
        const int tr[4] = {2, 0, 3, 1};
        const double lo[4] = {(double)phi_l, (double)mu_l, (double)sig_l, (double)rho_l};
        const double hi[4] = {(double)phi_u, (double)mu_u, (double)sig_u, (double)rho_u};
        for (int d = 0; d < 4; ++d) { c.transforms[d] = tr[d]; c.prior_lo[d] = lo[d]; c.prior_hi[d] = hi[d]; }
        ssme_lw_handle raw = nullptr;
        check(ssme_lw_create(&c, &raw));
        h_ = std::shared_ptr<ssme_lw_s>(raw, [](ssme_lw_handle p) { if (p) ssme_lw_destroy(p); });
    }
    // LWFilterWithCovs::filter(obs, cov) / LWFilter2WithCovs::filter (:840, :2050)
    template <typename Osv, typename Cvsv>
    void filter(const Osv& yt, const Cvsv& zt) { step((double)yt(0), (double)zt(0)); expectations_.clear(); }
    // ... with functionals
    template <typename Osv, typename Cvsv, typename F>
    void filter(const Osv& yt, const Cvsv& zt, const std::vector<F>& fs) {
        step((double)yt(0), (double)zt(0));
        compute_expectations(fs, (double)zt(0));
    }
    // LWFilter::filter(data[, fs]) / LWFilter2::filter (:238, :1447): no covariate
    template <typename Osv>
    void filter(const Osv& yt) { step((double)yt(0), 0.0); expectations_.clear(); }
    template <typename Osv, typename F>
    void filter(const Osv& yt, const std::vector<F>& fs) {
        step((double)yt(0), 0.0);
        compute_expectations(fs, 0.0);
    }
    float_t getLogCondLike() const { return last_; }
    // the reference's getExpectations(): one matrix per functional of the last filter() call (:1177-1180)
    std::vector<Mat> getExpectations() const { return expectations_; }
    // weighted posterior means of (phi, mu, sigma, rho) under the last step's weights
    std::vector<double> getParamMeans() const {
        std::vector<double> m(4);
        check(ssme_lw_get_param_means(h_.get(), m.data()));
        return m;
    }
    // built-in functionals by id: 0-3 = SSME_H_* of the state, 4-7 = phi, mu, sigma, rho
    std::vector<double> getExpectations(const std::vector<int32_t>& ids) const {
        std::vector<double> e(ids.size());
        if (!ids.empty()) check(ssme_lw_get_expectations(h_.get(), ids.data(), (int32_t)ids.size(), e.data()));
        return e;
    }
    ssme_lw_handle native() const { return h_.get(); }

private:
    void step(double y, double z) {
        double out = 0.0;
        const int rc = ssme_lw_step(h_.get(), &y, &z, &out);
        if (rc != SSME_OK) throw std::runtime_error(std::string(ssme_pf_strerror(rc)) + " (" + ssme_lw_last_error(h_.get()) + ")");
        last_ = (float_t)out;
    }
    // Which built-in id reproduces h?  Six probes with distinct states and parameters; -1 = evaluate on the host.
    template <typename F>
    static int classify(const F& h, double z, double* scale) {
        using T = detail::lw_func_traits<F>;
        static const double px[6] = {-1.7, -0.3125, 0.0, 0.5625, 1.9, 3.25};
        static const double pp[6][4] = {{0.91, -0.07, 0.021, -0.31}, {0.83, 0.02, 0.034, -0.12}, {0.95, 0.09, 0.077, -0.45},
                                        {0.88, -0.03, 0.055, -0.02}, {0.97, 0.05, 0.012, -0.27}, {0.81, -0.09, 0.093, -0.38}};
        bool is_const = true, is_x = true, is_x2 = true, is_vol = true, is_p[4] = {true, true, true, true};
        float_t c0 = 0;
        for (int q = 0; q < 6; ++q) {
            typename T::Ssv xv; typename T::Psv pv;
            xv(0) = (float_t)px[q];
            for (int d = 0; d < 4; ++d) pv(d) = (float_t)pp[q][d];
            const Mat m = T::call(h, xv, z, pv);
            if (m.rows() != 1 || m.cols() != 1) return -1;
            const float_t v = m(0, 0), xx = xv(0);
            if (q == 0) c0 = v;
            is_const = is_const && (v == c0);
            is_x = is_x && (v == xx);
            is_x2 = is_x2 && (v == xx * xx);
            const float_t e = (float_t)std::exp((float_t)0.5 * xx);
            is_vol = is_vol && (std::fabs(v - e) <= 4 * std::numeric_limits<float_t>::epsilon() * e);
            for (int d = 0; d < 4; ++d) is_p[d] = is_p[d] && (v == pv(d));
        }
        *scale = 1.0;
        if (is_const) { *scale = (double)c0 / 42.0; return SSME_H_CONST42; }
        if (is_x) return SSME_H_X;
        if (is_x2) return SSME_H_X2;
        if (is_vol) return SSME_H_VOL;
        for (int d = 0; d < 4; ++d) if (is_p[d]) return 4 + d;
        return -1;
    }
    template <typename F>
    void compute_expectations(const std::vector<F>& fs, double z) {
        using T = detail::lw_func_traits<F>;
        expectations_.clear();
        if (fs.empty()) return;
        std::vector<int> kind(fs.size(), -1);
        std::vector<double> scale(fs.size(), 1.0);
        std::vector<int32_t> ids;
        for (std::size_t i = 0; i < fs.size(); ++i) {
            if (probe_) kind[i] = classify(fs[i], z, &scale[i]);
            if (kind[i] >= 0) ids.push_back(kind[i]);
        }
        const std::vector<double> dev = getExpectations(ids);
        std::vector<double> x, th, w;
        std::size_t d = 0;
        for (std::size_t i = 0; i < fs.size(); ++i) {
            if (kind[i] >= 0) {
                Mat m(1, 1);
                m(0, 0) = (float_t)(dev[d++] * scale[i]);
                expectations_.push_back(m);
                continue;
            }
            if (w.empty()) {       // one download of (x, untransformed theta, weights) serves every host functional
                x.resize(nparts); th.resize(4 * nparts); w.resize(nparts);
                check(ssme_lw_download_weights(h_.get(), 0, x.data(), th.data(), w.data()));
            }
            std::vector<double> acc;
            long rows = 0, cols = 0;
            double wsum = 0.0;
            for (std::size_t p = 0; p < nparts; ++p) {
                typename T::Ssv xv; typename T::Psv pv;
                xv(0) = (float_t)x[p];
                for (int q = 0; q < 4; ++q) pv(q) = (float_t)th[(std::size_t)q * nparts + p];
                const Mat hv = T::call(fs[i], xv, z, pv);
                if (p == 0) { rows = (long)hv.rows(); cols = (long)hv.cols(); acc.assign((std::size_t)(rows * cols), 0.0); }
                for (long r = 0; r < rows; ++r)
                    for (long c = 0; c < cols; ++c) acc[(std::size_t)(r * cols + c)] += (double)hv(r, c) * w[p];
                wsum += w[p];
            }
            Mat m(rows, cols);
            for (long r = 0; r < rows; ++r)
                for (long c = 0; c < cols; ++c) m(r, c) = (float_t)(acc[(std::size_t)(r * cols + c)] / wsum);
            expectations_.push_back(m);
        }
    }

    std::shared_ptr<ssme_lw_s> h_;
    float_t last_ = 0;
    bool probe_ = true;
    std::vector<Mat> expectations_;
};

template <std::size_t nparts, typename float_t = double, typename Mat = detail::small_matrix<float_t>>
using svol_lw_2_par_gpu = svol_lw_1_par_gpu<nparts, float_t, 1, Mat>;

// ---- SwarmWithCovs over SVOL-leverage members (include/ssme/pswarm_filter.h:325-560; test/test_pswarm.cpp:146-208) ----
// All nparamparts member filters live in ONE handle (n_filters = nparamparts, one theta row each): update(y, z) is one
// launch; the swarm's log conditional likelihood and expectations are the plain averages over the members, as the
// reference's intra/inter_agg_func compute them (:392-460).  samp_untrans_params() is the reference's pure virtual.
template <std::size_t n_state_parts, std::size_t n_param_parts, typename float_t = double>
class swarm_with_covs_gpu {
public:
    using float_type = float_t;
    using func = int;                                            // SSME_H_* functional id
    explicit swarm_with_covs_gpu(const std::vector<func>& fs, gpu_options o = gpu_options()) : fs_(fs), opt_(o) {}
    virtual ~swarm_with_covs_gpu() = default;
    virtual std::vector<float_t> samp_untrans_params() = 0;      // order phi, mu, sigma, rho

    template <typename Osv, typename Csv>
    void update(const Osv& yt, const Csv& zt) {
        if (!h_) finish_construction();
        const double y = (double)yt(0), z = (double)zt(0);
        check(ssme_pf_step(h_.get(), &y, &z, nullptr), h_.get());
        aggregate();
        ++num_obs_;
    }
    float_t getLogCondLike() const { return log_cond_like_; }
    std::vector<double> getExpectations() const { return expectations_; }
    const std::vector<double>& params() const { return theta_; }         // [n_param_parts][4]

private:
    // intra/inter_agg_func (pswarm_filter.h:96-160): plain means over the members, reduced on the device, one download
    void aggregate() {
        if (fs_.size() > 4) throw std::invalid_argument("at most 4 device functionals per swarm");
        std::vector<int32_t> ids(fs_.begin(), fs_.end());
        double lcl = 0.0;
        expectations_.assign(fs_.size(), 0.0);
        check(ssme_pf_swarm_aggregate(h_.get(), ids.data(), (int32_t)ids.size(), &lcl, expectations_.data()), h_.get());
        log_cond_like_ = (float_t)lcl;
    }
    void finish_construction() {                                  // pswarm_filter.h:280-304
        theta_.resize(n_param_parts * 4);
        for (std::size_t i = 0; i < n_param_parts; ++i) {
            const std::vector<float_t> p = samp_untrans_params();
            if (p.size() != 4) throw std::invalid_argument("samp_untrans_params must return phi, mu, sigma, rho");
            for (int d = 0; d < 4; ++d) theta_[i * 4 + d] = (double)p[d];
        }
        h_ = handle(SSME_MODEL_SVOL_LEVERAGE, (int)n_state_parts, (int)n_param_parts, opt_.seed, opt_.resampler, opt_.resamp_sched,
                    opt_.device, 0, detail::dtype_of<float_t>());
        check(ssme_pf_set_params(h_.get(), theta_.data(), 4, (int)n_param_parts), h_.get());
    }
    std::vector<func> fs_;
    gpu_options opt_;
    handle h_;
    std::vector<double> theta_, expectations_;
    float_t log_cond_like_ = 0;
    unsigned num_obs_ = 0;
};

// ---- Swarm over univariate-SVOL members (include/ssme/pswarm_filter.h:23-320: the variant without covariates) ---------
// update(y) = filter(y, fs) on every member, plain averages over members; parameters in the model's ctor order of
// svol_bs: samp_untrans_params() returns (phi, beta, sigma) (univ_svol_bootstrap_filter.h:34).
template <std::size_t n_state_parts, std::size_t n_param_parts, typename float_t = double>
class swarm_gpu {
public:
    using float_type = float_t;
    using func = int;
    explicit swarm_gpu(const std::vector<func>& fs, gpu_options o = gpu_options()) : fs_(fs), opt_(o) {}
    virtual ~swarm_gpu() = default;
    virtual std::vector<float_t> samp_untrans_params() = 0;      // phi, beta, sigma

    template <typename Osv>
    void update(const Osv& yt) {
        if (!h_) finish_construction();
        const double y = (double)yt(0);
        check(ssme_pf_step(h_.get(), &y, nullptr, nullptr), h_.get());
        aggregate();
    }
    float_t getLogCondLike() const { return log_cond_like_; }
    std::vector<double> getExpectations() const { return expectations_; }

private:
    // intra/inter_agg_func (pswarm_filter.h:96-160): plain means over the members, reduced on the device, one download
    void aggregate() {
        if (fs_.size() > 4) throw std::invalid_argument("at most 4 device functionals per swarm");
        std::vector<int32_t> ids(fs_.begin(), fs_.end());
        double lcl = 0.0;
        expectations_.assign(fs_.size(), 0.0);
        check(ssme_pf_swarm_aggregate(h_.get(), ids.data(), (int32_t)ids.size(), &lcl, expectations_.data()), h_.get());
        log_cond_like_ = (float_t)lcl;
    }
    void finish_construction() {
        std::vector<double> theta(n_param_parts * 3);
        for (std::size_t i = 0; i < n_param_parts; ++i) {
            const std::vector<float_t> p = samp_untrans_params();
            if (p.size() != 3) throw std::invalid_argument("samp_untrans_params must return phi, beta, sigma");
            theta[i * 3 + 0] = (double)p[1]; theta[i * 3 + 1] = (double)p[0]; theta[i * 3 + 2] = (double)p[2];   // C ABI order: beta, phi, sigma
        }
        h_ = handle(SSME_MODEL_SVOL, (int)n_state_parts, (int)n_param_parts, opt_.seed, opt_.resampler, opt_.resamp_sched, opt_.device, 0,
                    detail::dtype_of<float_t>());
        check(ssme_pf_set_params(h_.get(), theta.data(), 3, (int)n_param_parts), h_.get());
    }
    std::vector<func> fs_;
    gpu_options opt_;
    handle h_;
    std::vector<double> expectations_;
    float_t log_cond_like_ = 0;
};

// ---- headerless CSV -> rows of doubles (utils::read_data, include/ssme/utils.h:25-64) -------------------------------
// Same tolerance as the reference: rows that fail to parse are skipped; an unreadable file yields an empty vector
// (callers then throw length_error, estimate_univ_svol.h:112-113).
struct csv_row {
    std::vector<double> v;
    double operator()(std::size_t i) const { return v[i]; }
};
inline std::vector<csv_row> read_data(const std::string& file_loc, std::size_t ncols = 1) {
    std::vector<csv_row> rows;
    std::ifstream in(file_loc);
    std::string line, cell;
    while (std::getline(in, line)) {
        csv_row r;
        std::istringstream ls(line);
        bool ok = true;
        while (ok && std::getline(ls, cell, ',')) {
            try { r.v.push_back(std::stod(cell)); } catch (const std::exception&) { ok = false; }
        }
        if (ok && r.v.size() >= ncols) { r.v.resize(ncols); rows.push_back(r); }
    }
    return rows;
}

}  // namespace ssme_gpu
#endif
