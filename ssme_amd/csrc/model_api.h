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
// tests/models/svol_student_t.h does exactly that against the oracle's callback-driven model).
//
// VECTOR state / observation (BSFilter<nparts, dimx, dimy, ...> with dimx, dimy > 1; up to 4 each).  The header adds
//     static constexpr int dim_x = 2, dim_y = 1;
//     static __device__ void init_vec(const ssme::ModelConst& c, const double* zn, double* x0);
//                                                                  // q1Samp: x_0 from dim_x standard normals
//     static __device__ void prop_vec(const ssme::ModelConst& c, const double* x, const double* zn, double zcov, double* xn,
//                                     const ssme::ExpTabEntry* etab);      // fSamp: dim_x normals in, dim_x components out
//     static __device__ double logg_vec(const ssme::ModelConst& c, const double* y, const double* x, const ssme::ExpTabEntry* etab);
// INSTEAD of prop / logg.  Particles are stored as dim_x planes ([dim_x][n_filters][N]; ssme_pf_download_state returns them in that
// order), a series is T rows of dim_y values, component 0 of the normals is the pair's Box-Muller draw of the scalar kernels and
// component d >= 1 comes from one more Philox call per particle pair (counter stream STREAM_XDIM + d).  Such a model runs on the
// tiled step kernel at every N (no whole-series kernel), unsharded; device functionals see component 0.
// tests/models/svol_two_factor.h is the test model (two volatility factors, two observed series).
#pragma once
#include <type_traits>
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

namespace ssme {
constexpr int kMaxDim = 4;                 // largest dim_x / dim_y of a user model
enum { STREAM_XDIM = 96 };                 // counter stream of state component d >= 1: STREAM_XDIM + d
template <class M, class = void> struct user_dims { static constexpr int dx = 1, dy = 1; };
template <class M> struct user_dims<M, std::void_t<decltype(M::dim_x), decltype(M::dim_y)>> {
    static constexpr int dx = M::dim_x, dy = M::dim_y;
    static_assert(dx >= 1 && dx <= kMaxDim && dy >= 1 && dy <= kMaxDim, "dim_x and dim_y of a user model: 1 .. 4");
};
// what the kernels call: a scalar model through its prop / logg, a vector model through its *_vec functions (the other set is a
// stub that no launch reaches: vector models never run the scalar kernels and the other way round)
template <class M, bool VEC = (user_dims<M>::dx > 1 || user_dims<M>::dy > 1)> struct user_calls;
template <class M> struct user_calls<M, false> {
    static __device__ __forceinline__ double prop(const ModelConst& c, double x, double zn, double zcov, const ExpTabEntry* etab) { return M::prop(c, x, zn, zcov, etab); }
    static __device__ __forceinline__ double logg(const ModelConst& c, double y, double x, const ExpTabEntry* etab) { return M::logg(c, y, x, etab); }
    static __device__ __forceinline__ void init_vec(const ModelConst& c, const double* zn, double* x0) { x0[0] = zn[0] * c.a2; }
    static __device__ __forceinline__ void prop_vec(const ModelConst& c, const double* x, const double* zn, double zcov, double* xn, const ExpTabEntry* etab) { xn[0] = M::prop(c, x[0], zn[0], zcov, etab); }
    static __device__ __forceinline__ double logg_vec(const ModelConst& c, const double* y, const double* x, const ExpTabEntry* etab) { return M::logg(c, y[0], x[0], etab); }
};
template <class M> struct user_calls<M, true> {
    static __device__ __forceinline__ double prop(const ModelConst&, double, double, double, const ExpTabEntry*) { return 0.0; }
    static __device__ __forceinline__ double logg(const ModelConst&, double, double, const ExpTabEntry*) { return 0.0; }
    static __device__ __forceinline__ void init_vec(const ModelConst& c, const double* zn, double* x0) { M::init_vec(c, zn, x0); }
    static __device__ __forceinline__ void prop_vec(const ModelConst& c, const double* x, const double* zn, double zcov, double* xn, const ExpTabEntry* etab) { M::prop_vec(c, x, zn, zcov, xn, etab); }
    static __device__ __forceinline__ double logg_vec(const ModelConst& c, const double* y, const double* x, const ExpTabEntry* etab) { return M::logg_vec(c, y, x, etab); }
};
#if SSME_HAS_USER_MODEL
template <int MODEL> constexpr int model_dx() { return MODEL == MODEL_USER0 ? user_dims<ssme_user_model0>::dx : 1; }
template <int MODEL> constexpr int model_dy() { return MODEL == MODEL_USER0 ? user_dims<ssme_user_model0>::dy : 1; }
#else
template <int MODEL> constexpr int model_dx() { return 1; }
template <int MODEL> constexpr int model_dy() { return 1; }
#endif
}  // namespace ssme
