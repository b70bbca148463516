// swarm_shape.hpp -- TEST STAND-IN.  The reference's Swarm / SwarmWithCovs templates cannot be compiled here (Eigen and
// pf are absent), so this header restates, in its own words, exactly what those templates REQUIRE of their ModType
// argument -- the type relationships and the statements a ModType must support -- so that the claim "svol_leverage_gpu /
// svol_bs_member_gpu drop into the unmodified swarm templates" is checked by a compiler:
//   include/ssme/pswarm_filter.h:29-60    typedefs pulled from ModType, pair/vector member types, static_assert on a pf base
//   include/ssme/pswarm_filter.h:71,367   std::array of (model, functions) pairs => ModType default-constructible
//   include/ssme/pswarm_filter.h:86-92,380-388   filter(y[, z], fs); std::vector<DynMat> r = getExpectations(); getLogCondLike()
//   include/ssme/pswarm_filter.h:96-160   running means: DynMat + DynMat / scalar, 0x0 matrices mean "nothing yet"
//   include/ssme/pswarm_filter.h:272-275,572-575   filt_func = std::bind(proto, _1[, _2], params)
//   include/ssme/pswarm_filter.h:280-304  models[i].first = instantiate_mod(theta) => copy/move-assignable
// It runs the members in a plain loop (the reference hands them to split_data_thread_pool; its averaging over members is the
// same arithmetic when nparamparts is a multiple of the thread count).  Not product code.
#pragma once
#include <array>
#include <cstddef>
#include <functional>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

namespace shape {

// ---- minimal linear-algebra stand-ins with the operations the swarm templates apply to Eigen types ----
template <typename F>
struct vec1 {                                  // Eigen::Matrix<F,1,1>
    F v{};
    F operator()(int) const { return v; }
    F& operator()(int) { return v; }
};
template <typename F>
struct vec4 {                                  // Eigen::Matrix<F,4,1>
    F v[4]{};
    F operator()(int i) const { return v[i]; }
    F& operator()(int i) { return v[i]; }
};
template <typename F>
class dynmat {                                 // Eigen::Matrix<F,Dynamic,Dynamic>
public:
    dynmat() = default;
    dynmat(long r, long c) : r_(r), c_(c), d_((std::size_t)(r * c)) {}
    dynmat(const vec1<F>& x) : r_(1), c_(1), d_(1, x.v) {}          // a fixed vector converts to a dynamic matrix, as in Eigen
    long rows() const { return r_; }
    long cols() const { return c_; }
    F operator()(long i, long j) const { return d_[(std::size_t)(i * c_ + j)]; }
    F& operator()(long i, long j) { return d_[(std::size_t)(i * c_ + j)]; }
    friend dynmat operator+(const dynmat& a, const dynmat& b) {
        if (a.r_ != b.r_ || a.c_ != b.c_) throw std::invalid_argument("dynmat size mismatch");
        dynmat o(a.r_, a.c_);
        for (std::size_t i = 0; i < o.d_.size(); ++i) o.d_[i] = a.d_[i] + b.d_[i];
        return o;
    }
    friend dynmat operator/(const dynmat& a, F s) {
        dynmat o(a.r_, a.c_);
        for (std::size_t i = 0; i < o.d_.size(); ++i) o.d_[i] = a.d_[i] / s;
        return o;
    }
private:
    long r_ = 0, c_ = 0;
    std::vector<F> d_;
};

// ---- stand-ins for pf::bases::pf_base / pf_withcov_base [pf-recollection: abstract bases with these two virtuals] ----
template <typename F>
struct pf_base_like {
    using osv = vec1<F>;
    using func = std::function<const dynmat<F>(const vec1<F>&)>;
    virtual void filter(const osv& y, const std::vector<func>& fs) = 0;
    virtual F getLogCondLike() const = 0;
    virtual ~pf_base_like() = default;
};
template <typename F>
struct pf_withcov_base_like {
    using osv = vec1<F>;
    using func = std::function<const dynmat<F>(const vec1<F>&, const vec1<F>&)>;
    virtual void filter(const osv& y, const vec1<F>& z, const std::vector<func>& fs) = 0;
    virtual F getLogCondLike() const = 0;
    virtual ~pf_withcov_base_like() = default;
};

// ---- what SwarmWithCovs asks of ModType ----
template <typename ModType, std::size_t n_filt_funcs, std::size_t nparamparts>
class swarm_with_covs_shape {
public:
    using float_type = typename ModType::float_type;
    using osv = vec1<float_type>;
    using csv = vec1<float_type>;
    using ssv = vec1<float_type>;
    using psv = vec4<float_type>;
    using DynMat = typename ModType::dynamic_matrix;
    using filt_func = typename ModType::func;
    using state_cov_parm_func = std::function<const DynMat(const ssv&, const csv&, const psv&)>;
    using mod_funcs_pair = std::pair<ModType, std::vector<filt_func>>;
    using mats_and_loglike = std::pair<std::vector<DynMat>, float_type>;
    static_assert(std::is_base_of<pf_withcov_base_like<float_type>, ModType>::value, "ModType must inherit from the particle filter base");

    explicit swarm_with_covs_shape(const std::vector<state_cov_parm_func>& fs) : fresh_(true), n_obs_(0) {
        if (fs.size() != n_filt_funcs) throw std::invalid_argument("wrong number of filtering functions");
        protos_ = fs;
        expectations_.resize(n_filt_funcs);
    }
    virtual ~swarm_with_covs_shape() = default;
    virtual psv samp_untrans_params() = 0;
    virtual ModType instantiate_mod(const psv& untrans_params) = 0;

    void update(const osv& yt, const csv& zt) {
        if (fresh_) finish_construction();
        mats_and_loglike agg;
        agg.first.resize(n_filt_funcs);
        agg.second = 0;
        for (std::size_t i = 0; i < nparamparts; ++i) agg = running_mean(agg, one_member(yt, zt, members_[i]));
        expectations_ = agg.first;
        log_cond_like_ = agg.second;
        ++n_obs_;
    }
    float_type getLogCondLike() const { return log_cond_like_; }
    std::vector<DynMat> getExpectations() const { return expectations_; }

private:
    static mats_and_loglike one_member(const osv& yt, const csv& zt, mod_funcs_pair& pf_funcs) {
        pf_funcs.first.filter(yt, zt, pf_funcs.second);
        mats_and_loglike r;
        r.first = pf_funcs.first.getExpectations();
        r.second = pf_funcs.first.getLogCondLike();
        return r;
    }
    static mats_and_loglike running_mean(const mats_and_loglike& agg, const mats_and_loglike& term) {
        mats_and_loglike res = agg;
        res.second += term.second / static_cast<float_type>(nparamparts);
        bool started = false;
        for (std::size_t i = 0; i < n_filt_funcs; ++i) started = started || agg.first[i].rows() > 0 || agg.first[i].cols() > 0;
        for (std::size_t i = 0; i < n_filt_funcs; ++i)
            res.first[i] = started ? res.first[i] + term.first[i] / static_cast<float_type>(nparamparts)
                                   : term.first[i] / static_cast<float_type>(nparamparts);
        return res;
    }
    filt_func bind_params(const state_cov_parm_func& in_f, const psv& params) {
        filt_func out_f = std::bind(in_f, std::placeholders::_1, std::placeholders::_2, params);
        return out_f;
    }
    void finish_construction() {
        if (!fresh_) throw std::runtime_error("models sampled twice");
        psv untrans_params;
        for (std::size_t i = 0; i < nparamparts; ++i) {
            untrans_params = samp_untrans_params();
            members_[i].first = instantiate_mod(untrans_params);
            std::vector<filt_func> funcs;
            for (std::size_t j = 0; j < n_filt_funcs; ++j) funcs.push_back(bind_params(protos_[j], untrans_params));
            members_[i].second = funcs;
        }
        fresh_ = false;
    }
    bool fresh_;
    std::vector<state_cov_parm_func> protos_;
    std::array<mod_funcs_pair, nparamparts> members_;
    float_type log_cond_like_{};
    std::vector<DynMat> expectations_;
    unsigned n_obs_;
};

// ---- what Swarm (no covariates) asks of ModType ----
template <typename ModType, std::size_t n_filt_funcs, std::size_t nparamparts>
class swarm_shape {
public:
    using float_type = typename ModType::float_type;
    using osv = vec1<float_type>;
    using ssv = vec1<float_type>;
    using psv = vec4<float_type>;
    using DynMat = typename ModType::dynamic_matrix;
    using filt_func = typename ModType::func;
    using state_parm_func = std::function<const DynMat(const ssv&, const psv&)>;
    using mod_funcs_pair = std::pair<ModType, std::vector<filt_func>>;
    static_assert(std::is_base_of<pf_base_like<float_type>, ModType>::value, "ModType must inherit from the particle filter base");

    explicit swarm_shape(const std::vector<state_parm_func>& fs) : protos_(fs) { expectations_.resize(n_filt_funcs); }
    virtual ~swarm_shape() = default;
    virtual psv samp_untrans_params() = 0;
    virtual ModType instantiate_mod(const psv& untrans_params) = 0;
    void update(const osv& yt) {
        if (fresh_) {
            for (std::size_t i = 0; i < nparamparts; ++i) {
                const psv p = samp_untrans_params();
                members_[i].first = instantiate_mod(p);
                for (std::size_t j = 0; j < n_filt_funcs; ++j) {
                    filt_func f = std::bind(protos_[j], std::placeholders::_1, p);
                    members_[i].second.push_back(f);
                }
            }
            fresh_ = false;
        }
        float_type lcl = 0;
        std::vector<DynMat> ex(n_filt_funcs);
        for (std::size_t i = 0; i < nparamparts; ++i) {
            members_[i].first.filter(yt, members_[i].second);
            const std::vector<DynMat> e = members_[i].first.getExpectations();
            lcl += members_[i].first.getLogCondLike() / static_cast<float_type>(nparamparts);
            for (std::size_t j = 0; j < n_filt_funcs; ++j)
                ex[j] = i ? ex[j] + e[j] / static_cast<float_type>(nparamparts) : e[j] / static_cast<float_type>(nparamparts);
        }
        log_cond_like_ = lcl;
        expectations_ = ex;
    }
    float_type getLogCondLike() const { return log_cond_like_; }
    std::vector<DynMat> getExpectations() const { return expectations_; }

private:
    bool fresh_ = true;
    std::vector<state_parm_func> protos_;
    std::array<mod_funcs_pair, nparamparts> members_;
    float_type log_cond_like_{};
    std::vector<DynMat> expectations_;
};

}  // namespace shape
