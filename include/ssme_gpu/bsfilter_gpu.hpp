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
// The classes are templated on the pack / vector types only through duck typing, so they compile with the
// reference's Eigen-based param::pack and Eigen vectors as well as with plain stand-ins (tests/cpp).
// Host std::function callbacks cannot run on the device: fs entries are SSME_H_* enums (include/ssme_pf.h).
// Errors: std::invalid_argument / std::runtime_error, as the reference throws; NaN/-inf are values.
#ifndef SSME_GPU_BSFILTER_GPU_HPP
#define SSME_GPU_BSFILTER_GPU_HPP

#include <cmath>
#include <cstddef>
#include <fstream>
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
    handle(int model, int nparts, int nfilters, std::uint64_t seed, int resampler, int rs, int device, unsigned first_id) {
        ssme_pf_config c{};
        c.model = model; c.n_particles = nparts; c.n_filters = nfilters; c.dtype = SSME_F64; c.resampler = resampler;
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
};

// ---- svol_bs ---------------------------------------------------------------------------------------------
template <std::size_t nparts, typename float_t = double>
class svol_bs_gpu {
public:
    using float_type = float_t;
    svol_bs_gpu(const float_t& phi, const float_t& beta, const float_t& sigma, gpu_options o = gpu_options())
        : h_(SSME_MODEL_SVOL, (int)nparts, 1, o.seed, o.resampler, o.resamp_sched, o.device, 0) {
        const double th[3] = {(double)beta, (double)phi, (double)sigma};
        check(ssme_pf_set_params(h_.get(), th, 3, 1), h_.get());
    }
    // ctor from a param::pack: order beta, phi, ss (univ_svol_bootstrap_filter.h:55-61)
    template <typename Pack>
    explicit svol_bs_gpu(const Pack& pp, gpu_options o = gpu_options())
        : svol_bs_gpu((float_t)pp.get_untrans_params(1, 1)(0), (float_t)pp.get_untrans_params(0, 0)(0),
                      (float_t)std::sqrt((double)pp.get_untrans_params(2, 2)(0)), o) {}

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

// ---- svol_leverage (BSFilterWC) ------------------------------------------------------------------------------
template <std::size_t nparts, typename float_t = double, typename Mat = std::vector<std::vector<float_t>>>
class svol_leverage_gpu {
public:
    using float_type = float_t;
    using dynamic_matrix = Mat;
    using func = int;                 // SSME_H_* functional id instead of std::function
    svol_leverage_gpu() = default;    // Swarm default-constructs its array of models (pswarm_filter.h:71)
    svol_leverage_gpu(const float_t& phi, const float_t& mu, const float_t& sigma, const float_t& rho, unsigned /*dte*/ = 0,
                      gpu_options o = gpu_options(), unsigned filter_id = 0)
        : h_(SSME_MODEL_SVOL_LEVERAGE, (int)nparts, 1, o.seed, o.resampler, o.resamp_sched, o.device, filter_id) {
        const double th[4] = {(double)phi, (double)mu, (double)sigma, (double)rho};
        check(ssme_pf_set_params(h_.get(), th, 4, 1), h_.get());
    }
    template <typename Osv, typename Cvsv>
    void filter(const Osv& yt, const Cvsv& zt, const std::vector<func>& fs = std::vector<func>()) {
        if (!h_) throw std::runtime_error("model not constructed");
        const double y = (double)yt(0), z = (double)zt(0);
        double out = 0.0;
        check(ssme_pf_step(h_.get(), &y, &z, &out), h_.get());
        last_ = (float_t)out;
        expectations_.assign(fs.size(), 0.0);
        for (std::size_t i = 0; i < fs.size(); ++i) check(ssme_pf_get_expectations(h_.get(), fs[i], &expectations_[i]), h_.get());
    }
    float_t getLogCondLike() const { return last_; }
    std::vector<double> getExpectations() const { return expectations_; }
    ssme_pf_handle native() const { return h_.get(); }

private:
    handle h_;
    float_t last_ = 0;
    std::vector<double> expectations_;
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

// ---- svol_lw_1_par (LWFilterWithCovs) ---------------------------------------------------------------------------
// test/test_liu_west.cpp:22-157: ctor (delta, phi_l, phi_u, mu_l, mu_u, sig_l, sig_u, rho_l, rho_u[, dte]);
// filter(y, z), getLogCondLike() (liu_west_filter.h:971-1159).  Transforms as svol_lw_1_par passes them to its base:
// logit, null, log, twice_fisher (test_liu_west.cpp:70).
template <std::size_t nparts, typename float_t = double>
class svol_lw_1_par_gpu {
public:
    svol_lw_1_par_gpu(const float_t& delta, const float_t& phi_l, const float_t& phi_u, const float_t& mu_l, const float_t& mu_u,
                      const float_t& sig_l, const float_t& sig_u, const float_t& rho_l, const float_t& rho_u, unsigned /*dte*/ = 0,
                      gpu_options o = gpu_options()) {
        ssme_lw_config c{};
        c.n_particles = (int)nparts; c.n_filters = 1; c.seed = o.seed; c.device = o.device; c.first_filter_id = 0;
        c.delta = (double)delta;
        const int tr[4] = {2, 0, 3, 1};
        const double lo[4] = {(double)phi_l, (double)mu_l, (double)sig_l, (double)rho_l};
        const double hi[4] = {(double)phi_u, (double)mu_u, (double)sig_u, (double)rho_u};
        for (int d = 0; d < 4; ++d) { c.transforms[d] = tr[d]; c.prior_lo[d] = lo[d]; c.prior_hi[d] = hi[d]; }
        ssme_lw_handle raw = nullptr;
        check(ssme_lw_create(&c, &raw));
        h_ = std::shared_ptr<ssme_lw_s>(raw, [](ssme_lw_handle p) { if (p) ssme_lw_destroy(p); });
    }
    template <typename Osv, typename Cvsv>
    void filter(const Osv& yt, const Cvsv& zt) {
        const double y = (double)yt(0), z = (double)zt(0);
        double out = 0.0;
        const int rc = ssme_lw_step(h_.get(), &y, &z, &out);
        if (rc != SSME_OK) throw std::runtime_error(std::string(ssme_pf_strerror(rc)) + " (" + ssme_lw_last_error(h_.get()) + ")");
        last_ = (float_t)out;
    }
    float_t getLogCondLike() const { return last_; }
    // weighted posterior means of (phi, mu, sigma, rho) under the last step's weights
    std::vector<double> getParamMeans() const {
        std::vector<double> m(4);
        check(ssme_lw_get_param_means(h_.get(), m.data()));
        return m;
    }
    ssme_lw_handle native() const { return h_.get(); }

private:
    std::shared_ptr<ssme_lw_s> h_;
    float_t last_ = 0;
};

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
        std::vector<double> ll(n_param_parts), e(n_param_parts);
        check(ssme_pf_step(h_.get(), &y, &z, ll.data()), h_.get());
        double s = 0.0;
        for (double v : ll) s += v;
        log_cond_like_ = (float_t)(s / (double)n_param_parts);
        expectations_.assign(fs_.size(), 0.0);
        for (std::size_t i = 0; i < fs_.size(); ++i) {
            check(ssme_pf_get_expectations(h_.get(), fs_[i], e.data()), h_.get());
            double se = 0.0;
            for (double v : e) se += v;
            expectations_[i] = se / (double)n_param_parts;
        }
        ++num_obs_;
    }
    float_t getLogCondLike() const { return log_cond_like_; }
    std::vector<double> getExpectations() const { return expectations_; }
    const std::vector<double>& params() const { return theta_; }         // [n_param_parts][4]

private:
    void finish_construction() {                                  // pswarm_filter.h:280-304
        theta_.resize(n_param_parts * 4);
        for (std::size_t i = 0; i < n_param_parts; ++i) {
            const std::vector<float_t> p = samp_untrans_params();
            if (p.size() != 4) throw std::invalid_argument("samp_untrans_params must return phi, mu, sigma, rho");
            for (int d = 0; d < 4; ++d) theta_[i * 4 + d] = (double)p[d];
        }
        h_ = handle(SSME_MODEL_SVOL_LEVERAGE, (int)n_state_parts, (int)n_param_parts, opt_.seed, opt_.resampler, opt_.resamp_sched,
                    opt_.device, 0);
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
        std::vector<double> ll(n_param_parts), e(n_param_parts);
        check(ssme_pf_step(h_.get(), &y, nullptr, ll.data()), h_.get());
        double s = 0.0;
        for (double v : ll) s += v;
        log_cond_like_ = (float_t)(s / (double)n_param_parts);
        expectations_.assign(fs_.size(), 0.0);
        for (std::size_t i = 0; i < fs_.size(); ++i) {
            check(ssme_pf_get_expectations(h_.get(), fs_[i], e.data()), h_.get());
            double se = 0.0;
            for (double v : e) se += v;
            expectations_[i] = se / (double)n_param_parts;
        }
    }
    float_t getLogCondLike() const { return log_cond_like_; }
    std::vector<double> getExpectations() const { return expectations_; }

private:
    void finish_construction() {
        std::vector<double> theta(n_param_parts * 3);
        for (std::size_t i = 0; i < n_param_parts; ++i) {
            const std::vector<float_t> p = samp_untrans_params();
            if (p.size() != 3) throw std::invalid_argument("samp_untrans_params must return phi, beta, sigma");
            theta[i * 3 + 0] = (double)p[1]; theta[i * 3 + 1] = (double)p[0]; theta[i * 3 + 2] = (double)p[2];   // C ABI order: beta, phi, sigma
        }
        h_ = handle(SSME_MODEL_SVOL, (int)n_state_parts, (int)n_param_parts, opt_.seed, opt_.resampler, opt_.resamp_sched, opt_.device, 0);
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
