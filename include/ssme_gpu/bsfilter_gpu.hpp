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

}  // namespace ssme_gpu
#endif
