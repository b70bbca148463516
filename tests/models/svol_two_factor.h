// svol_two_factor.h -- TEST MODEL for the VECTOR form of the extension point (ssme_amd/csrc/model_api.h): dim_x = 2, dim_y = 2,
// i.e. what pf::filters::BSFilter<nparts, 2, 2, ...> would hold.  Two log-volatility factors with correlated innovations,
//     x1' = phi1 x1 + sigma1 e1,      x2' = phi2 x2 + sigma2 (rho e1 + sqrt(1 - rho^2) e2),
// two observed series
//     y1 ~ N(0, beta^2 exp(x1 + x2)),      y2 ~ N(0, beta^2 exp(x2)),
// x_0 = (sigma1 z1, sigma2 sqrt(1 - rho^2) z2).  theta = (beta, phi1, phi2, sigma1, sigma2, rho).  Not in the reference: it exists
// to prove the vector path -- gathers of every component at the ancestor's index, one more Philox call per pair for the second
// normal, vector observations -- bit for bit against the oracle's callback-driven restatement (tests/test_parity_gpu.py).
#pragma once

struct ssme_user_model0 {
    static constexpr int n_theta = 6;
    static constexpr int dim_x = 2, dim_y = 2;
    static ssme::ModelConst derive(const double* th) {            // host only
        const double beta = th[0], phi1 = th[1], phi2 = th[2], s1 = th[3], s2 = th[4], rho = th[5];
        ssme::ModelConst c{};
        c.a0 = phi1;
        c.a1 = phi2;
        c.a2 = s1;
        c.a3 = s2 * rho;
        c.a4 = s2 * ssme::dsqrt(1.0 - rho * rho);
        c.a5 = ssme::dlog(beta);
        c.a6 = 1.0 / (beta * beta);
        c.bad = !(beta > 0.0);
        return c;
    }
    static __device__ __forceinline__ void init_vec(const ssme::ModelConst& c, const double* zn, double* x0) {
        x0[0] = zn[0] * c.a2;
        x0[1] = zn[1] * c.a4;
    }
    static __device__ __forceinline__ void prop_vec(const ssme::ModelConst& c, const double* x, const double* zn, double, double* xn,
                                                    const ssme::ExpTabEntry*) {
        xn[0] = c.a0 * x[0] + zn[0] * c.a2;
        xn[1] = (c.a1 * x[1] + zn[0] * c.a3) + zn[1] * c.a4;
    }
    static __device__ __forceinline__ double logg_vec(const ssme::ModelConst& c, const double* y, const double* x, const ssme::ExpTabEntry* etab) {
        const double s1 = x[0] + x[1], s2 = x[1];
        const double l1 = (-(c.a5 + 0.5 * s1) - 0.91893853320467274178) - 0.5 * (((y[0] * y[0]) * c.a6) * ssme::dexp_scaled_t(-s1, 0, etab));
        const double l2 = (-(c.a5 + 0.5 * s2) - 0.91893853320467274178) - 0.5 * (((y[1] * y[1]) * c.a6) * ssme::dexp_scaled_t(-s2, 0, etab));
        return l1 + l2;
    }
};
