// estimate_univ_svol_gpu -- the shipped example (example/main.cpp + example/estimate_univ_svol.h) with the particle
// filter on the MI355X: adaptive particle-marginal Metropolis-Hastings for the univariate SVOL model.
//
// This is a HARNESS for end-to-end timing (BASELINE.json configs[2]); the product is
// libssme_pf.so behind include/ssme_gpu/bsfilter_gpu.hpp.  The reference's own estimator (ada_pmmh_mvn.h) needs Eigen
// and pf, which are absent here, so the outer loop is restated compactly on plain arrays, following:
//   CLI                       example/main.cpp:15-45 (+ optional 6th/7th argument: particles per filter, seed)
//   start point, C0, t0, t1   example/estimate_univ_svol.h:150-160   (theta = beta, phi, ss; transforms null, twice_fisher, log)
//   prior                     example/estimate_univ_svol.h:84-104    (beta ~ N(1,1), phi ~ U(0,1), ss ~ InvGamma(.001,.001))
//   adaptation                include/ssme/ada_pmmh_mvn.h:214-250    (running mean/covariance, Ct = sd (Sigma + eps I) in (t0, t1))
//   accept/reject + outputs   include/ssme/ada_pmmh_mvn.h:266-377    (samples: untransformed csv rows; messages: 8 columns)
//   likelihood                one call per iteration: num_pfilters replicate filters on the device, whole series,
//                             log-mean-exp (thread_pool.h:189-215,263-268)
//
// build: g++ -O2 -std=c++17 -Iinclude examples/estimate_univ_svol_gpu.cpp ssme_amd/libssme_pf.so -Wl,-rpath,$PWD/ssme_amd -o estimate_univ_svol_gpu
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <random>
#include <string>

#include <ssme_gpu/bsfilter_gpu.hpp>

namespace {

constexpr int P = 3;
using vec = std::array<double, P>;
using mat = std::array<std::array<double, P>, P>;

struct scalar1 { double v; double operator()(int) const { return v; } };

// theta on the unconstrained scale with the example's transforms (parameters.h:317-457)
struct svol_pack {
    vec t;                                                      // beta (null), phi (twice_fisher), ss (log)
    vec untrans() const {
        const double e = t[1];
        const double phi = (e > 0) ? 2.0 / (1.0 + std::exp(-e)) - 1.0 : 1.0 - 2.0 / (1.0 + std::exp(e));
        return {t[0], phi, std::exp(t[2])};
    }
    double log_jacobian() const { return (std::log(2.0) + t[1] - 2.0 * std::log1p(std::exp(t[1]))) + t[2]; }
    scalar1 get_untrans_params(unsigned a, unsigned) const { return scalar1{untrans()[a]}; }
};

double log_prior(const svol_pack& p) {
    const vec u = p.untrans();
    const double beta = u[0], phi = u[1], ss = u[2];
    double lp = -0.5 * std::log(2.0 * M_PI) - 0.5 * (beta - 1.0) * (beta - 1.0);                 // N(1, 1)
    lp += (phi >= 0.0 && phi <= 1.0) ? 0.0 : -INFINITY;                                          // U(0, 1)
    const double a = 0.001, b = 0.001;                                                           // InvGamma(a, b)
    lp += (ss > 0.0) ? a * std::log(b) - std::lgamma(a) - (a + 1.0) * std::log(ss) - b / ss : -INFINITY;
    return lp;
}

bool cholesky(const mat& A, mat& L) {
    L = mat{};
    for (int i = 0; i < P; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) { if (!(s > 0.0)) return false; L[i][i] = std::sqrt(s); }
            else L[i][j] = s / L[j][j];
        }
    return true;
}

std::string with_time(const std::string& base) {
    char buf[80];
    const std::time_t now = std::time(nullptr);
    std::strftime(buf, sizeof(buf), "%Y-%m-%d.%H-%M-%S", std::localtime(&now));
    return base + "_" + buf;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 6) {
        std::cerr << "Please enter:\n1.) datafile location, \n2.) samples_base_name, \n3.) messages file base name, \n"
                     "4.) number of mcmc iterations. \n5.) number of pfilters. \n[6.) particles per filter (default 500)] [7.) seed]\n";
        return 0;
    }
    const std::string data_loc = argv[1];
    const unsigned iters = (unsigned)std::atoi(argv[4]);
    const int num_pfilters = std::atoi(argv[5]);
    const int nparts = argc > 6 ? std::atoi(argv[6]) : 500;
    const std::uint64_t seed = argc > 7 ? std::strtoull(argv[7], nullptr, 10) : (std::uint64_t)std::time(nullptr);

    const auto data = ssme_gpu::read_data(data_loc, 1);
    if (data.empty()) { std::cerr << "can't read in data\n"; return 1; }
    std::cerr << "first row of data: \n" << data[0](0) << "\n";
    std::ofstream samples(with_time(argv[2])), messages(with_time(argv[3]));

    ssme_gpu::gpu_options opt;
    opt.seed = seed;
    ssme_gpu::svol_log_like_evaluator loglike(data, nparts, num_pfilters, opt);
    std::mt19937_64 gen(seed ^ 0x9E3779B97F4A7C15ull);
    std::normal_distribution<double> rnorm(0.0, 1.0);
    std::uniform_real_distribution<double> runif(0.0, 1.0);

    svol_pack cur{{1.0, std::log(1.5) - std::log(0.5), std::log(2.0e-4)}};       // estimate_univ_svol.h:153-155
    const unsigned t0 = 150, t1 = 1000;
    const double sd = 2.4 * 2.4 / P, eps = 0.01;
    mat Ct{}; for (int i = 0; i < P; ++i) Ct[i][i] = 0.15;
    vec mean{}; mat sigma{};
    double old_ll = 0, new_ll = 0, old_lp = 0, new_lp = 0, log_accept = -INFINITY, ma_rate = 0.0;
    bool accepted = false;
    double dev_ms = 0.0;
    const auto wall0 = std::chrono::steady_clock::now();

    for (unsigned it = 0; it < iters; ++it) {
        if (it == 0) {
            old_ll = loglike(cur, seed + it);
            old_lp = log_prior(cur) + cur.log_jacobian();
        } else {
            // running moments of the transformed chain and the proposal covariance (ada_pmmh_mvn.h:214-250)
            const vec& x = cur.t;
            if (it == 1) {
                for (int i = 0; i < P; ++i) mean[i] += x[i];
            } else if (it == 2) {
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j) sigma[i][j] = 0.5 * (mean[i] * mean[j] + x[i] * x[j] - mean[i] * x[j] - x[i] * mean[j]);
                for (int i = 0; i < P; ++i) mean[i] = 0.5 * mean[i] + 0.5 * x[i];
            } else {
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j)
                        sigma[i][j] = sigma[i][j] * (it - 2.0) / (it - 1.0) + (x[i] - mean[i]) * (x[j] - mean[j]) / it;
                for (int i = 0; i < P; ++i) mean[i] = ((it - 1.0) * mean[i] + x[i]) / it;
            }
            if (t1 > it && it > t0)
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j) Ct[i][j] = sd * (sigma[i][j] + (i == j ? eps : 0.0));
            mat L;
            if (!cholesky(Ct, L)) { std::cerr << "proposal covariance not positive definite\n"; return 1; }
            vec z{rnorm(gen), rnorm(gen), rnorm(gen)};
            svol_pack prop = cur;
            for (int i = 0; i < P; ++i)
                for (int j = 0; j <= i; ++j) prop.t[i] += L[i][j] * z[j];
            new_lp = log_prior(prop) + prop.log_jacobian();
            new_ll = loglike(prop, seed + it);
            log_accept = new_lp + new_ll - old_lp - old_ll;
            accepted = std::log(runif(gen)) < log_accept;              // false for NaN, as in the reference
            ma_rate = (accepted ? 1.0 : 0.0) / (it + 1.0) + it * ma_rate / (it + 1.0);
            if (accepted) { cur = prop; old_lp = new_lp; old_ll = new_ll; }
            else if (std::isnan(log_accept)) std::cerr << "accept proability had a nan in it\n";
        }
        dev_ms += loglike.device_ms();
        const vec u = cur.untrans();
        samples << u[0] << "," << u[1] << "," << u[2] << "\n";
        if (it == 0) messages << "iter number, accept rate, old_ll, new_ll, old_lprior, new_lprior, accept prob, outcome\n";
        messages << it + 1 << ", " << ma_rate << ", " << old_ll << ", " << new_ll << ", " << old_lp << ", " << new_lp << ", "
                 << log_accept << ", " << accepted << "\n";
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    const double psteps = (double)iters * num_pfilters * (double)nparts * (double)data.size();
    std::fprintf(stderr,
                 "{\"harness\": \"ada-PMMH univ-SVOL\", \"iters\": %u, \"num_pfilters\": %d, \"nparts\": %d, \"T\": %zu, "
                 "\"wall_s\": %.3f, \"device_s\": %.3f, \"iters_per_s\": %.3f, \"particle_steps_per_s\": %.4g, "
                 "\"accept_rate\": %.4f, \"last_theta\": [%.6g, %.6g, %.6g]}\n",
                 iters, num_pfilters, nparts, data.size(), wall, dev_ms * 1e-3, iters / wall, psteps / wall, ma_rate,
                 cur.untrans()[0], cur.untrans()[1], cur.untrans()[2]);
    return 0;
}
