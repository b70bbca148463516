// model_api.h -- the model extension point of the bootstrap-filter kernels.
//
// The reference's models are five virtual callbacks on a class derived from pf::filters::BSFilter
// (example/univ_svol_bootstrap_filter.h:37-41: logQ1Ev, logMuEv, logGEv, fSamp, q1Samp).  Host virtuals cannot run per
// particle on a GPU, so a model here is a POLICY: a struct of static functions compiled into the step kernels.  The three
// built-in models (svol_bs, svol_leverage, linear Gaussian) are written against the same interface below, and ONE more
// model can be compiled in without touching a kernel:
//
//     python -c "from ssme_amd import build; print(build.build_user_model('path/to/my_model.h', 'mymodel'))"
//         -> build/user/libssme_pf_mymodel.so          (hipcc ... -DSSME_USER_MODEL_HEADER="path/to/my_model.h")
//     SSME_PF_LIB=build/user/libssme_pf_mymodel.so ...      ssme_pf_config::model = SSME_MODEL_USER0
//
// The header defines `struct ssme_user_model0` with
//     static constexpr int n_theta = ...;                         // length of the untransformed parameter vector (<= 8)
//     static ssme::ModelConst derive(const double* theta);           // per-filter constants a0..a6, computed ONCE on the host:
//                                                                  // a2 is the standard deviation of the t = 0 draw
//                                                                  // x_0 = a2 * z (q1Samp with logMuEv == logQ1Ev, as in
//                                                                  // both reference models); bad != 0 => logG = -inf
//     static __device__ double prop(const ssme::ModelConst& c, double x, double zn, double zcov, const ssme::ExpTabEntry* etab);
//                                                                  // fSamp: the new state from x, a standard normal zn
//                                                                  // and the covariate
//     static __device__ double logg(const ssme::ModelConst& c, double y, double x, const ssme::ExpTabEntry* etab);
//                                                                  // logGEv
// using only + - * fma and the functions of ssme_math.h (dexp_scaled_t(x, 0, etab), dlog, dsqrt, ...): fixed IEEE operation
// sequences, so that a CPU restatement of the same sequence reproduces the filter bit for bit (the parity test of
// tests/models/svol_student_t.h does exactly that against the oracle's callback-driven model).  d_x = d_y = 1.
#pragma once
#include "ssme_math.h"

namespace ssme {

// Derived per-filter constants (the host computes them with the same ssme_math functions).
struct ModelConst {
    double a0, a1, a2, a3, a4, a5, a6;
    int32_t bad;
    int32_t pad;
};

enum { MODEL_SVOL = 0, MODEL_SVOL_LEVERAGE = 1, MODEL_LIN_GAUSS = 2, MODEL_USER0 = 3 };

}  // namespace ssme

#ifdef SSME_USER_MODEL_HEADER
#include SSME_USER_MODEL_HEADER
#define SSME_HAS_USER_MODEL 1
#else
#define SSME_HAS_USER_MODEL 0
#endif
