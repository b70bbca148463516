// The C++ surface of a USER model (include/ssme_gpu/bsfilter_gpu.hpp: user_bs_gpu) driven as the reference's callers drive a
// BSFilter<nparts, dimx, dimy, ...> object: mod.filter(y); mod.getLogCondLike(); expectations of functions of the whole state.
// Linked against the library built with tests/models/svol_two_factor.h (dim_x = dim_y = 2).  Prints "name value" lines that
// tests/test_cpp_adaptor.py compares with the oracle's callback-driven model.
#include <array>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/ssme_gpu/bsfilter_gpu.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<double> spy;
    {
        std::ifstream f(argv[1]);
        std::string line;
        while (std::getline(f, line)) if (!line.empty()) spy.push_back(std::stod(line));
    }
    constexpr std::size_t N = 3000;
    ssme_gpu::gpu_options o;
    o.seed = 21;
    ssme_gpu::user_bs_gpu<N, 2, 2> mod({1.1, 0.95, 0.9, 0.2, 0.15, -0.4}, o, /*filter_id=*/1);
    using sv = ssme_gpu::user_bs_gpu<N, 2, 2>::state_vector;
    std::vector<ssme_gpu::user_bs_gpu<N, 2, 2>::func> fs = {
        [](const sv& x) { return x[0] + x[1]; },            // the log-variance of the first series
        [](const sv& x) { return x[0] * x[1]; },
        [](const sv&) { return 42.0; }};
    double ll = 0.0;
    const int T = 8;
    for (int t = 0; t < T; ++t) {
        const std::array<double, 2> y = {spy[t], spy[100 + t]};
        if (t == T - 1) mod.filter(y, fs); else mod.filter(y);
        ll += mod.getLogCondLike();
    }
    std::printf("user_vec_ll %.17g\n", ll);
    std::printf("user_vec_last %.17g\n", (double)mod.getLogCondLike());
    std::printf("user_vec_sum %.17g\n", mod.getExpectations()[0]);
    std::printf("user_vec_prod %.17g\n", mod.getExpectations()[1]);
    std::printf("user_vec_42 %.17g\n", mod.getExpectations()[2]);
    // the wrong dimensions are refused
    try { ssme_gpu::user_bs_gpu<N, 1, 1> bad({1.0}, o); std::printf("dims_check missing\n"); }
    catch (const std::invalid_argument&) { std::printf("dims_check ok\n"); }
    return 0;
}
