// step_latency.cpp -- latency of the unchanged-caller path, mod.filter(y); ll += mod.getLogCondLike()
// (example/estimate_univ_svol.h:121-127), through the C++ adaptor, without any Python in the loop.
//   g++ -std=c++17 -O2 tools/step_latency.cpp -o tools/step_latency ssme_amd/libssme_pf.so -Wl,-rpath,$PWD/ssme_amd
//   ./tools/step_latency tests/golden/spy_returns.csv
#include <chrono>
#include <cstdio>
#include <fstream>
#include <vector>

#include "../include/ssme_gpu/bsfilter_gpu.hpp"
#include "../tests/cpp/swarm_shape.hpp"      // the reference's SwarmWithCovs template, restated structurally (Eigen / pf are absent)

struct vec1 { double v; double operator()(int) const { return v; } };

template <std::size_t N>
static void run(const std::vector<vec1>& data, const char* label) {
    ssme_gpu::gpu_options o;
    o.seed = 1;
    ssme_gpu::svol_bs_gpu<N, double> mod(0.95, 1.0, 0.25, o, 0);
    double ll = 0.0;
    for (int t = 0; t < 64; ++t) { mod.filter(data[t]); ll += mod.getLogCondLike(); }
    const int K = 1000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < K; ++t) { mod.filter(data[64 + t]); ll += mod.getLogCondLike(); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    std::printf("filter() N=%s: %.1f us per call (loglik %.6f)\n", label, us, ll);
}

template <std::size_t N>
static void run_lw(const std::vector<vec1>& data, const char* label) {
    ssme_gpu::gpu_options o;
    o.seed = 1;
    ssme_gpu::svol_lw_1_par_gpu<N> mod(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 0, o, 0);   // test/test_liu_west.cpp:160-200
    double ll = 0.0;
    for (int t = 0; t < 64; ++t) { mod.filter(data[t], vec1{t ? data[t - 1].v : 0.0}); ll += mod.getLogCondLike(); }
    const int K = 500;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < K; ++t) { mod.filter(data[64 + t], data[63 + t]); ll += mod.getLogCondLike(); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    std::printf("Liu-West filter() N=%s: %.1f us per call (loglik %.6f)\n", label, us, ll);
}

// SwarmWithCovs::update (pswarm_filter.h:380-460) with all members in one handle: one step launch + the aggregation
template <std::size_t NS, std::size_t NP>
struct lat_swarm : ssme_gpu::swarm_with_covs_gpu<NS, NP, double> {
    using ssme_gpu::swarm_with_covs_gpu<NS, NP, double>::swarm_with_covs_gpu;
    int k = 0;
    std::vector<double> samp_untrans_params() override {
        const double u = (0.5 + (k++ % 97)) / 97.0;
        return {0.8 + 0.19 * u, -0.1 + 0.2 * u, 0.01 + 0.09 * u, -0.5 + 0.49 * u};
    }
};
template <std::size_t NS, std::size_t NP>
static void run_swarm(const std::vector<vec1>& data, const char* label) {
    ssme_gpu::gpu_options o;
    o.seed = 1;
    lat_swarm<NS, NP> sw({SSME_H_CONST42, SSME_H_X}, o);
    double ll = 0.0;
    for (int t = 0; t < 64; ++t) { sw.update(data[t], vec1{t ? data[t - 1].v : 0.0}); ll += sw.getLogCondLike(); }
    const int K = 500;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < K; ++t) { sw.update(data[64 + t], data[63 + t]); ll += sw.getLogCondLike(); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    std::printf("swarm update() %s: %.1f us per call (loglik %.6f, E[x] %.6f)\n", label, us, ll, sw.getExpectations()[1]);
}

// The UNMODIFIED swarm template (pswarm_filter.h:325-560 through its structural stand-in) over svol_leverage_gpu members that
// share a swarm_context: filter() -> getExpectations() -> getLogCondLike() per member, ONE launch per observation.
template <std::size_t NS, std::size_t NP>
struct ctx_swarm : shape::swarm_with_covs_shape<ssme_gpu::svol_leverage_gpu<NS, double, shape::dynmat<double>, shape::vec1<double>, shape::vec1<double>,
                                                                          shape::vec1<double>, shape::pf_withcov_base_like<double>>, 2, NP> {
    using mod = ssme_gpu::svol_leverage_gpu<NS, double, shape::dynmat<double>, shape::vec1<double>, shape::vec1<double>, shape::vec1<double>,
                                            shape::pf_withcov_base_like<double>>;
    using base = shape::swarm_with_covs_shape<mod, 2, NP>;
    using base::base;
    using psv = typename base::psv;
    int k = 0;
    std::shared_ptr<typename mod::context> ctx;
    psv samp_untrans_params() override {
        const double u = (0.5 + (k++ % 97)) / 97.0;
        psv p;
        p(0) = 0.8 + 0.19 * u; p(1) = -0.1 + 0.2 * u; p(2) = 0.01 + 0.09 * u; p(3) = -0.5 + 0.49 * u;
        return p;
    }
    mod instantiate_mod(const psv& th) override { return mod(th(0), th(1), th(2), th(3), ctx); }
};
template <std::size_t NS, std::size_t NP>
static void run_ctx_swarm(const std::vector<vec1>& data, const char* label) {
    using sw_t = ctx_swarm<NS, NP>;
    using sv = shape::vec1<double>;
    using Mat = shape::dynmat<double>;
    std::vector<typename sw_t::state_cov_parm_func> fs;
    fs.push_back([](const sv&, const sv&, const shape::vec4<double>&) -> const Mat { sv a; a(0) = 42.0; return a; });
    fs.push_back([](const sv& x, const sv&, const shape::vec4<double>&) -> const Mat { sv a; a(0) = x(0); return a; });
    sw_t sw(fs);
    ssme_gpu::gpu_options o;
    o.seed = 1;
    o.declared_functionals = {SSME_H_CONST42, SSME_H_X};
    sw.ctx = std::make_shared<typename sw_t::mod::context>(SSME_MODEL_SVOL_LEVERAGE, NS, NP, o);
    double ll = 0.0;
    for (int t = 0; t < 64; ++t) { sw.update(sv{data[t].v}, sv{t ? data[t - 1].v : 0.0}); ll += sw.getLogCondLike(); }
    const int K = 300;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < K; ++t) { sw.update(sv{data[64 + t].v}, sv{data[63 + t].v}); ll += sw.getLogCondLike(); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    std::printf("UNMODIFIED swarm template + swarm_context, update() %s: %.1f us per call (loglik %.6f, E[x] %.6f)\n", label, us, ll,
                sw.getExpectations()[1](0, 0));
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<vec1> data;
    std::ifstream f(argv[1]);
    double v;
    while (f >> v) data.push_back(vec1{v});
    run<500>(data, "500");
    run<4096>(data, "4096");
    run<65536>(data, "2^16");
    run<1048576>(data, "2^20");
    run_swarm<500, 100>(data, "100 members x 500 particles, 2 functionals");
    run_swarm<16384, 64>(data, "64 members x 2^14 particles, 2 functionals");
    run_swarm<16384, 512>(data, "512 members x 2^14 particles, 2 functionals (BASELINE.json configs[3] per GPU)");
    run_ctx_swarm<500, 100>(data, "100 members x 500 particles, 2 functionals");
    run_ctx_swarm<16384, 64>(data, "64 members x 2^14 particles, 2 functionals");
    run_ctx_swarm<16384, 512>(data, "512 members x 2^14 particles, 2 functionals");
    run_lw<500>(data, "500");
    run_lw<65536>(data, "2^16");
    return 0;
}
