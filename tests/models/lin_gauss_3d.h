// lin_gauss_3d.h -- TEST MODEL for the VECTOR form of the extension point (ssme_amd/csrc/model_api.h) with an odd shape: dim_x = 3,
// dim_y = 1.  Three independent AR(1) components observed through their sum,
//     x_d' = phi x_d + sigma_d e_d  (d = 1, 2, 3),      y = x_1 + x_2 + x_3 + tau v,      x_d(0) ~ N(0, sigma_d^2 / (1 - phi^2)),
// theta = (phi, sigma_1, sigma_2, sigma_3, tau).  Linear and Gaussian, so the exact log-likelihood is the Kalman filter's: the test
// compares the device with the oracle's restatement bit for bit AND with that exact value within Monte-Carlo error -- an anchor that
// does not share a line with either implementation (independent normals per component, gathers of all three planes).
#pragma once

struct ssme_user_model0 {
    static constexpr int n_theta = 5;
    static constexpr int dim_x = 3, dim_y = 1;
    static ssme::ModelConst derive(const double* th) {            // host only
        const double phi = th[0], tau = th[4];
        ssme::ModelConst c{};
        c.a0 = phi;
        c.a1 = th[1];
        c.a2 = th[2];
        c.a3 = th[3];
        c.a4 = 1.0 / ssme::dsqrt(1.0 - phi * phi);                // stationary sd of component d: sigma_d * a4
        c.a5 = ssme::dlog(tau);
        c.a6 = 1.0 / tau;
        c.bad = !(tau > 0.0);
        return c;
    }
    static __device__ __forceinline__ void init_vec(const ssme::ModelConst& c, const double* zn, double* x0) {
        x0[0] = zn[0] * (c.a1 * c.a4);
        x0[1] = zn[1] * (c.a2 * c.a4);
        x0[2] = zn[2] * (c.a3 * c.a4);
    }
    static __device__ __forceinline__ void prop_vec(const ssme::ModelConst& c, const double* x, const double* zn, double, double* xn,
                                                    const ssme::ExpTabEntry*) {
        xn[0] = c.a0 * x[0] + zn[0] * c.a1;
        xn[1] = c.a0 * x[1] + zn[1] * c.a2;
        xn[2] = c.a0 * x[2] + zn[2] * c.a3;
    }
    static __device__ __forceinline__ double logg_vec(const ssme::ModelConst& c, const double* y, const double* x, const ssme::ExpTabEntry*) {
        const double d = (y[0] - ((x[0] + x[1]) + x[2])) * c.a6;
        return (-c.a5 - 0.91893853320467274178) - 0.5 * (d * d);
    }
};
