// Compiles the header-only adaptor against stand-ins for the reference's Eigen-based types (Eigen is absent
// here) and drives it like the reference's callers: estimate_univ_svol.h:119-127, pswarm_filter.h:380-388.
// Prints the log-likelihoods; tests/test_cpp_adaptor.py compares them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../include/ssme_gpu/bsfilter_gpu.hpp"

struct vec1 {                          // stand-in for Eigen::Matrix<double,1,1>
    double v;
    double operator()(int) const { return v; }
};
struct pack3 {                         // stand-in for param::pack<double,3>::get_untrans_params(i,i)
    double p[3];
    vec1 get_untrans_params(unsigned a, unsigned) const { return vec1{p[a]}; }
};

// svol_swarm_1 of test/test_pswarm.cpp:146-208 with a deterministic stand-in for its uniform prior samplers
struct test_swarm : ssme_gpu::swarm_with_covs_gpu<600, 5, double> {
    using ssme_gpu::swarm_with_covs_gpu<600, 5, double>::swarm_with_covs_gpu;
    int k = 0;
    std::vector<double> samp_untrans_params() override {
        const double u = 0.1 + 0.2 * k++;
        return {0.8 + 0.19 * u, -0.1 + 0.2 * u, 0.01 + 0.09 * u, -0.5 + 0.49 * u};
    }
};

struct test_swarm_nocov : ssme_gpu::swarm_gpu<400, 3, double> {
    using ssme_gpu::swarm_gpu<400, 3, double>::swarm_gpu;
    int k = 0;
    std::vector<double> samp_untrans_params() override { const double u = 0.2 + 0.3 * k++; return {0.9 + 0.05 * u, 0.8 + 0.4 * u, 0.2 + 0.1 * u}; }
};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<vec1> data;
    std::ifstream f(argv[1]);
    double v;
    while (f >> v && data.size() < 64) data.push_back(vec1{v});
    ssme_gpu::gpu_options o;
    o.seed = 77;
    // (1) the log_like_eval loop with the model object (estimate_univ_svol.h:119-127)
    pack3 theta{{1.0, 0.95, 0.0625}};                    // beta, phi, ss
    ssme_gpu::svol_bs_gpu<500, double> mod(theta, o);
    double logLike = 0.0;
    for (size_t row = 0; row < data.size(); ++row) {
        mod.filter(data[row]);
        logLike += mod.getLogCondLike();
    }
    std::printf("svol_bs %.17g\n", logLike);
    // (2) replicate-batched evaluation
    std::printf("log_like_eval_gpu %.17g\n", ssme_gpu::log_like_eval_gpu(theta, data, 500, 4, o));
    // (3) covariate model as Swarm::comp_func calls it (pswarm_filter.h:380-388)
    ssme_gpu::svol_leverage_gpu<1000> lev(0.9, 0.0, 1.0, -0.1, 0, o, 3), lev2;
    lev2 = lev;                                          // copy-assignable, default-constructible
    double ll = 0.0;
    for (size_t row = 0; row < 8; ++row) {
        lev2.filter(data[row], vec1{row ? data[row - 1].v : 0.0}, {SSME_H_CONST42, SSME_H_X});
        ll += lev2.getLogCondLike();
    }
    std::printf("svol_leverage %.17g\n", ll);
    std::printf("expect42 %.17g\n", lev2.getExpectations()[0]);
    // (5) persistent evaluator (one handle, fresh stream per call) == a fresh model with that seed
    ssme_gpu::svol_log_like_evaluator ev(data, 500, 4, o);
    ev(theta, 5);                                        // some other stream first
    std::printf("evaluator %.17g\n", ev(theta, 77));
    // (6) Liu-West model as test/test_liu_west.cpp:160-200 drives it
    ssme_gpu::svol_lw_1_par_gpu<800> lwmod(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 0, o);
    double lwll = 0.0;
    for (size_t row = 0; row < 6; ++row) {
        lwmod.filter(data[row], vec1{row ? data[row - 1].v : 0.0});
        lwll += lwmod.getLogCondLike();
    }
    std::printf("liu_west %.17g\n", lwll);
    // (7) utils::read_data stand-in
    const auto rows = ssme_gpu::read_data(argv[1], 1);
    std::printf("read_data %zu %.17g\n", rows.size(), rows.empty() ? 0.0 : rows[0](0));
    // (8) particle swarm as test/test_pswarm.cpp:236-252 drives it
    test_swarm sw({SSME_H_CONST42, SSME_H_X}, o);
    double swll = 0.0;
    for (size_t row = 0; row < 5; ++row) {
        sw.update(data[row], vec1{row ? data[row - 1].v : 0.0});
        swll += sw.getLogCondLike();
    }
    std::printf("swarm %.17g\n", swll);
    std::printf("swarm42 %.17g\n", sw.getExpectations()[0]);
    std::printf("swarmx %.17g\n", sw.getExpectations()[1]);
    // (9) swarm without covariates (pswarm_filter.h:23-320)
    test_swarm_nocov sn({SSME_H_X}, o);
    double snll = 0.0;
    for (size_t row = 0; row < 4; ++row) { sn.update(data[row]); snll += sn.getLogCondLike(); }
    std::printf("swarm_nocov %.17g\n", snll);
    // (4) error mapping
    try { std::vector<vec1> empty; ssme_gpu::log_like_eval_gpu(theta, empty, 100, 1, o); std::printf("no-throw\n"); }
    catch (const std::length_error&) { std::printf("length_error ok\n"); }
    return 0;
}
