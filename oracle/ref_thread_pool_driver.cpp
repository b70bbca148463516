// ref_thread_pool_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref): a driver around the REFERENCE's own
// include/ssme/thread_pool.h (the one header of the hot path that needs nothing but the C++ standard library; pf,
// Eigen3 and Catch2 -- required by every other reference header -- are absent).
// The header is compiled where it lies under /root/reference (oracle/Makefile, target _ref); nothing of it is copied.
//
// thread_pool<dyn, static, out>::work(param) calls f(param, data) num_comps times and returns the log-mean-exp of the
// results (thread_pool.h:189-215, finalisation :263-268).  The driver feeds it a list of values through f and returns
// what the reference returns: the pin for ssme_pf_log_mean_exp / ssme_amd.parallel.log_mean_exp / orc_log_mean_exp.
// thread_pool.h uses std::exp, std::log, max_element and std::function without including their headers (its includers
// in the reference pull them in first); the standard headers are included here ahead of it for the same effect.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <vector>
using std::max_element;

#include <ssme/thread_pool.h>

extern "C" double ref_thread_pool_log_mean_exp(const double* vals, int n) {
    if (!vals || n < 1) return 0.0 / 0.0;
    std::atomic<unsigned> next{0};
    using pool_t = thread_pool<int, std::vector<double>, double>;
    pool_t::F f = [&next](int, std::vector<double> data) -> double { return data[next++ % data.size()]; };
    pool_t pool(f, (unsigned)n, /*mt=*/false);               // one worker: values are consumed in order
    pool.add_observed_data(std::vector<double>(vals, vals + n));
    return pool.work(0);
}
