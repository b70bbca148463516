// svol_student_t.h -- TEST MODEL for the extension point of ssme_amd/csrc/model_api.h: the univariate
// stochastic-volatility model of example/univ_svol_bootstrap_filter.h with Student-t observations,
//     x_t = phi x_{t-1} + sigma e_t,      y_t = beta exp(x_t / 2) t_nu,      x_0 ~ N(0, sigma^2 / (1 - phi^2)),
// theta = (beta, phi, sigma, nu).  Not in the reference: it exists to show that a fourth model is one header, no kernel edit.
//     logG(y | x) = lgamma((nu+1)/2) - lgamma(nu/2) - log(nu pi)/2 - log(beta) - x/2 - (nu+1)/2 log(1 + y^2 e^-x / (nu beta^2))
// Operation order below is the specification; tests/test_parity_gpu.py restates it on the CPU through the oracle's
// callback-driven model and compares bit for bit.
#pragma once
#include <cmath>

struct ssme_user_model0 {
    static constexpr int n_theta = 4;
    static ssme::ModelConst derive(const double* th) {            // host only: called once per filter by ssme_pf_set_params
        const double beta = th[0], phi = th[1], sigma = th[2], nu = th[3];
        ssme::ModelConst c{};
        c.a0 = phi;
        c.a1 = sigma;
        c.a2 = sigma / ssme::dsqrt(1.0 - phi * phi);                       // sd of the stationary t = 0 draw
        c.a3 = ((std::lgamma(0.5 * (nu + 1.0)) - std::lgamma(0.5 * nu)) - 0.5 * ssme::dlog(nu * 3.14159265358979311600)) - ssme::dlog(beta);
        c.a4 = 1.0 / (nu * (beta * beta));
        c.a5 = 0.5 * (nu + 1.0);
        c.bad = !(beta > 0.0) || !(nu > 0.0);
        return c;
    }
    static __device__ __forceinline__ double prop(const ssme::ModelConst& c, double x, double zn, double, const ssme::ExpTabEntry*) {
        return c.a0 * x + zn * c.a1;
    }
    static __device__ __forceinline__ double logg(const ssme::ModelConst& c, double y, double x, const ssme::ExpTabEntry* etab) {
        const double e = ssme::dexp_scaled_t(-x, 0, etab);
        const double w = 1.0 + ((y * y) * c.a4) * e;
        return (c.a3 - 0.5 * x) - c.a5 * ssme::dlog(w);
    }
};
