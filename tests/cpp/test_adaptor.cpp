// Compiles the header-only adaptor against stand-ins for the reference's Eigen-based types (Eigen is absent
// here) and drives it like the reference's callers: estimate_univ_svol.h:119-127, pswarm_filter.h:380-388.
// Prints the log-likelihoods; tests/test_cpp_adaptor.py compares them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "../../include/ssme_gpu/bsfilter_gpu.hpp"
#include "swarm_shape.hpp"

using vec1 = shape::vec1<double>;      // stand-in for Eigen::Matrix<double,1,1>
using Mat = shape::dynmat<double>;     // stand-in for Eigen::Matrix<double,Dynamic,Dynamic>
struct pack3 {                         // stand-in for param::pack<double,3>::get_untrans_params(i,i)
    double p[3];
    vec1 get_untrans_params(unsigned a, unsigned) const { return vec1{p[a]}; }
};

// ---- the model types handed to the (structurally restated) UNMODIFIED swarm templates ----
using lev_mod = ssme_gpu::svol_leverage_gpu<600, double, Mat, vec1, vec1, vec1, shape::pf_withcov_base_like<double>>;
using bs_mod = ssme_gpu::svol_bs_member_gpu<400, double, Mat, vec1, vec1, shape::pf_base_like<double>>;

// svol_swarm_1 of test/test_pswarm.cpp:146-208 on the swarm template's own terms: samp_untrans_params +
// instantiate_mod overrides, filter functions given as std::function (lambda) objects
struct unmodified_swarm : shape::swarm_with_covs_shape<lev_mod, 3, 5> {
    using base = shape::swarm_with_covs_shape<lev_mod, 3, 5>;
    using base::base;
    int k = 0;
    ssme_gpu::gpu_options opt;
    psv samp_untrans_params() override {
        const double u = 0.1 + 0.2 * k;
        psv p;
        p(0) = 0.8 + 0.19 * u; p(1) = -0.1 + 0.2 * u; p(2) = 0.01 + 0.09 * u; p(3) = -0.5 + 0.49 * u;
        return p;
    }
    lev_mod instantiate_mod(const psv& th) override { return lev_mod(th(0), th(1), th(2), th(3), 10, opt, (unsigned)k++); }
};
// the same swarm with its members in ONE handle behind the same unmodified template: instantiate_mod hands every model the
// swarm's context
struct unmodified_swarm_ctx : shape::swarm_with_covs_shape<lev_mod, 3, 5> {
    using base = shape::swarm_with_covs_shape<lev_mod, 3, 5>;
    using base::base;
    int k = 0;
    std::shared_ptr<lev_mod::context> ctx;
    psv samp_untrans_params() override {
        const double u = 0.1 + 0.2 * k++;
        psv p;
        p(0) = 0.8 + 0.19 * u; p(1) = -0.1 + 0.2 * u; p(2) = 0.01 + 0.09 * u; p(3) = -0.5 + 0.49 * u;
        return p;
    }
    lev_mod instantiate_mod(const psv& th) override { return lev_mod(th(0), th(1), th(2), th(3), ctx); }
};
struct unmodified_swarm_nocov_ctx : shape::swarm_shape<bs_mod, 1, 3> {
    using base = shape::swarm_shape<bs_mod, 1, 3>;
    using base::base;
    int k = 0;
    std::shared_ptr<bs_mod::context> ctx;
    psv samp_untrans_params() override { const double u = 0.2 + 0.3 * k++; psv p; p(0) = 0.9 + 0.05 * u; p(1) = 0.8 + 0.4 * u; p(2) = 0.2 + 0.1 * u; return p; }
    bs_mod instantiate_mod(const psv& th) override { return bs_mod(th(0), th(1), th(2), ctx); }
};
struct unmodified_swarm_nocov : shape::swarm_shape<bs_mod, 1, 3> {
    using base = shape::swarm_shape<bs_mod, 1, 3>;
    using base::base;
    int k = 0;
    ssme_gpu::gpu_options opt;
    psv samp_untrans_params() override { const double u = 0.2 + 0.3 * k; psv p; p(0) = 0.9 + 0.05 * u; p(1) = 0.8 + 0.4 * u; p(2) = 0.2 + 0.1 * u; return p; }
    bs_mod instantiate_mod(const psv& th) override { return bs_mod(th(0), th(1), th(2), opt, (unsigned)k++); }
};

// the batched fast path: all members in one handle
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
    ssme_gpu::svol_bs_gpu<500, double> mod(theta, o, 0);
    double logLike = 0.0;
    for (size_t row = 0; row < data.size(); ++row) {
        mod.filter(data[row]);
        logLike += mod.getLogCondLike();
    }
    std::printf("svol_bs %.17g\n", logLike);
    // (2) replicate-batched evaluation
    std::printf("log_like_eval_gpu %.17g\n", ssme_gpu::log_like_eval_gpu(theta, data, 500, 4, o));
    // (3) covariate model as Swarm::comp_func calls it (pswarm_filter.h:380-388)
    using lev1000 = ssme_gpu::svol_leverage_gpu<1000, double, Mat, vec1, vec1, vec1>;
    ssme_gpu::gpu_options od = o;                       // the first two functionals below ARE device functionals: say so
    od.declared_functionals = {SSME_H_CONST42, SSME_H_X};
    lev1000 lev(0.9, 0.0, 1.0, -0.1, 0, od, 3), lev2;
    lev2 = lev;                                          // copy-assignable, default-constructible
    // the reference test's constant lambda (test/test_pswarm.cpp:239-243), the state itself, and a function no device
    // functional covers (2x2 matrix-valued, uses the covariate): host path over the downloaded (x, weights)
    std::vector<lev1000::func> hs;
    hs.push_back([](const vec1&, const vec1&) -> const Mat { vec1 ans; ans(0) = 42.0; return ans; });
    hs.push_back([](const vec1& xt, const vec1&) -> const Mat { Mat m(1, 1); m(0, 0) = xt(0); return m; });
    hs.push_back([](const vec1& xt, const vec1& zt) -> const Mat {
        Mat m(2, 2); m(0, 0) = xt(0); m(0, 1) = xt(0) * xt(0); m(1, 0) = zt(0); m(1, 1) = std::sin(xt(0)); return m; });
    double ll = 0.0;
    for (size_t row = 0; row < 8; ++row) {
        lev2.filter(data[row], vec1{row ? data[row - 1].v : 0.0}, hs);
        ll += lev2.getLogCondLike();
    }
    std::printf("svol_leverage %.17g\n", ll);
    {
        const std::vector<Mat> e = lev2.getExpectations();
        std::printf("expect42 %.17g\n", e[0](0, 0));
        std::printf("expectx %.17g\n", e[1](0, 0));
        std::printf("host_x %.17g\nhost_x2 %.17g\nhost_z %.17g\nhost_sin %.17g\n", e[2](0, 0), e[2](0, 1), e[2](1, 0), e[2](1, 1));
        double x2dev = 0.0;
        ssme_gpu::check(ssme_pf_get_expectations(lev2.native(), SSME_H_X2, &x2dev));
        std::printf("dev_x2 %.17g\n", x2dev);
        // the same model with nothing declared (the default: every functional on the host) and with the opt-in probe
        lev1000 levh(0.9, 0.0, 1.0, -0.1, 0, o, 3);
        ssme_gpu::gpu_options op = o;
        op.probe_functionals = true;
        lev1000 levp(0.9, 0.0, 1.0, -0.1, 0, op, 3);
        // a clamp: equal to x on every probe point, different where the particles of this model never are -- and a tail
        // indicator that the fixed probes cannot see (the probe must not be the default)
        std::vector<lev1000::func> tricky = hs;
        tricky.push_back([](const vec1& xt, const vec1&) -> const Mat { Mat m(1, 1); m(0, 0) = xt(0) > 5.0 ? 5.0 : xt(0); return m; });
        tricky.push_back([](const vec1& xt, const vec1&) -> const Mat { Mat m(1, 1); m(0, 0) = xt(0) < -3.0 ? 1.0 : 0.0; return m; });
        for (size_t row = 0; row < 8; ++row) {
            levh.filter(data[row], vec1{row ? data[row - 1].v : 0.0}, tricky);
            levp.filter(data[row], vec1{row ? data[row - 1].v : 0.0}, tricky);
        }
        std::printf("hostdefault_42 %.17g\nhostdefault_x %.17g\n", levh.getExpectations()[0](0, 0), levh.getExpectations()[1](0, 0));
        std::printf("probe_x %.17g\n", levp.getExpectations()[1](0, 0));
        std::printf("tail_host %.17g\ntail_probe %.17g\n", levh.getExpectations()[4](0, 0), levp.getExpectations()[4](0, 0));
    }
    // (5) persistent evaluator (one handle, fresh stream per call) == a fresh model with that seed
    ssme_gpu::svol_log_like_evaluator ev(data, 500, 4, o);
    ev(theta, 5);                                        // some other stream first
    std::printf("evaluator %.17g\n", ev(theta, 77));
    // (6) Liu-West model as test/test_liu_west.cpp:160-200 drives it
    ssme_gpu::svol_lw_1_par_gpu<800> lwmod(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 0, o, 0);
    double lwll = 0.0;
    for (size_t row = 0; row < 6; ++row) {
        lwmod.filter(data[row], vec1{row ? data[row - 1].v : 0.0});
        lwll += lwmod.getLogCondLike();
    }
    std::printf("liu_west %.17g\n", lwll);
    // (6b) the alternative (SISR) Liu-West filter as test/test_liu_west.cpp:365-406 drives it, E[42] = 42
    ssme_gpu::svol_lw_2_par_gpu<800> lw2(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 10, o, 0);
    double lw2ll = 0.0;
    for (size_t row = 0; row < 6; ++row) {
        lw2.filter(data[row], vec1{row ? data[row - 1].v : 0.0});
        lw2ll += lw2.getLogCondLike();
    }
    std::printf("liu_west2 %.17g\n", lw2ll);
    std::printf("liu_west2_42 %.17g\n", lw2.getExpectations({SSME_H_CONST42})[0]);
    // (6c) functionals as test/test_liu_west.cpp:176-200 / :381-403 pass them: std::function (state, covariate, untransformed
    //      parameters) -> matrix; 42 and phi are recognised and run on the device, the 2x2 one is summed on the host
    {
        using psv4 = shape::vec4<double>;
        using lwfunc = std::function<const Mat(const vec1&, const vec1&, const psv4&)>;
        ssme_gpu::svol_lw_1_par_gpu<800, double, 0, Mat> lwf(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 10, o, 0);
        std::vector<lwfunc> lfs;
        lfs.push_back([](const vec1&, const vec1&, const psv4&) -> const Mat { vec1 ans; ans(0) = 42.0; return ans; });
        lfs.push_back([](const vec1&, const vec1&, const psv4& pt) -> const Mat { vec1 ans; ans(0) = pt(0); return ans; });
        lfs.push_back([](const vec1& xt, const vec1& zt, const psv4& pt) -> const Mat {
            Mat m(2, 2); m(0, 0) = xt(0); m(0, 1) = xt(0) * pt(2); m(1, 0) = zt(0); m(1, 1) = pt(3); return m; });
        for (size_t row = 0; row < 6; ++row) lwf.filter(data[row], vec1{row ? data[row - 1].v : 0.0}, lfs);
        const std::vector<Mat> e = lwf.getExpectations();
        const std::vector<double> dev = lwf.getExpectations(std::vector<int32_t>{SSME_H_X, 7});
        std::printf("lwf_ll %.17g\nlwf_42 %.17g\nlwf_phi %.17g\n", (double)lwf.getLogCondLike(), e[0](0, 0), e[1](0, 0));
        std::printf("lwf_host_x %.17g\nlwf_host_xsig %.17g\nlwf_host_z %.17g\nlwf_host_rho %.17g\n", e[2](0, 0), e[2](0, 1), e[2](1, 0), e[2](1, 1));
        std::printf("lwf_dev_x %.17g\nlwf_dev_rho %.17g\n", dev[0], dev[1]);
        // the no-covariate call (LWFilter::filter(data, fs), :238) with its two-argument functionals
        using lwfunc2 = std::function<const Mat(const vec1&, const psv4&)>;
        ssme_gpu::svol_lw_2_par_gpu<800, double, Mat> lwn(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 10, o, 0);
        std::vector<lwfunc2> nfs;
        nfs.push_back([](const vec1& xt, const psv4&) -> const Mat { vec1 ans; ans(0) = xt(0) * xt(0); return ans; });
        double nll = 0.0;
        for (size_t row = 0; row < 4; ++row) { lwn.filter(data[row], nfs); nll += lwn.getLogCondLike(); }
        std::printf("lwn_ll %.17g\nlwn_x2 %.17g\n", nll, lwn.getExpectations()[0](0, 0));
    }
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
    // (10) the UNMODIFIED swarm templates' requirements on ModType (swarm_shape.hpp restates pswarm_filter.h:29-60,86-92,
    //      272-304,380-388): typedefs, static_assert on the pf base, std::bind into filt_func, vector<DynMat> assignment
    {
        using sw_t = unmodified_swarm;
        std::vector<sw_t::state_cov_parm_func> fs;
        fs.push_back([](const vec1&, const vec1&, const shape::vec4<double>&) -> const Mat { vec1 ans; ans(0) = 42.0; return ans; });
        fs.push_back([](const vec1& xt, const vec1&, const shape::vec4<double>&) -> const Mat { vec1 ans; ans(0) = xt(0); return ans; });
        fs.push_back([](const vec1& xt, const vec1&, const shape::vec4<double>& pt) -> const Mat {   // uses the member's parameters
            vec1 ans; ans(0) = pt(1) + xt(0); return ans; });
        sw_t usw(fs);
        usw.opt = o;
        usw.opt.declared_functionals = {SSME_H_CONST42, SSME_H_X};      // the third uses the member's parameters: host
        double ull = 0.0;
        for (size_t row = 0; row < 5; ++row) {
            usw.update(data[row], vec1{row ? data[row - 1].v : 0.0});
            ull += usw.getLogCondLike();
        }
        std::printf("uswarm %.17g\n", ull);
        std::printf("uswarm42 %.17g\n", usw.getExpectations()[0](0, 0));
        std::printf("uswarmx %.17g\n", usw.getExpectations()[1](0, 0));
        std::printf("uswarmmux %.17g\n", usw.getExpectations()[2](0, 0));
        // (10b) the same swarm, members in ONE handle through a swarm_context: one launch per update, same numbers
        {
            unmodified_swarm_ctx csw(fs);
            ssme_gpu::gpu_options oc = o;
            oc.declared_functionals = {SSME_H_CONST42, SSME_H_X};
            csw.ctx = std::make_shared<lev_mod::context>(SSME_MODEL_SVOL_LEVERAGE, 600, 5, oc);
            double cll = 0.0;
            for (size_t row = 0; row < 5; ++row) {
                csw.update(data[row], vec1{row ? data[row - 1].v : 0.0});
                cll += csw.getLogCondLike();
            }
            std::printf("cswarm %.17g\n", cll);
            std::printf("cswarm42 %.17g\n", csw.getExpectations()[0](0, 0));
            std::printf("cswarmx %.17g\n", csw.getExpectations()[1](0, 0));
            std::printf("cswarmmux %.17g\n", csw.getExpectations()[2](0, 0));
            // a member given another observation than its swarm is an error, not a silent read of the cached row
            // members cannot join a running swarm; a member handed another observation than its swarm is an error
            try { lev_mod stray(0.9, 0.0, 0.05, -0.1, csw.ctx); std::printf("late_member no-throw\n"); }
            catch (const std::runtime_error&) { std::printf("late_member rejected\n"); }
        }
        // (10c) the reference hands the members to split_data_thread_pool workers (thread_pool.h:542-554): concurrent filter()
        //       calls on the members of one context, each thread its own subset -- same numbers as the serial loop
        {
            using ctx_t = lev_mod::context;
            ssme_gpu::gpu_options oc = o;
            oc.declared_functionals = {SSME_H_CONST42, SSME_H_X};
            auto ctx = std::make_shared<ctx_t>(SSME_MODEL_SVOL_LEVERAGE, 600, 5, oc);
            std::vector<lev_mod> mods;
            for (int k = 0; k < 5; ++k) {
                const double u = 0.1 + 0.2 * k;
                mods.push_back(lev_mod(0.8 + 0.19 * u, -0.1 + 0.2 * u, 0.01 + 0.09 * u, -0.5 + 0.49 * u, ctx));
            }
            std::vector<lev_mod::func> two;
            two.push_back([](const vec1&, const vec1&) -> const Mat { vec1 a; a(0) = 42.0; return a; });
            two.push_back([](const vec1& xt, const vec1&) -> const Mat { vec1 a; a(0) = xt(0); return a; });
            double tll = 0.0, tx = 0.0;
            for (size_t row = 0; row < 5; ++row) {
                const vec1 zt{row ? data[row - 1].v : 0.0};
                std::vector<std::thread> workers;
                for (int w = 0; w < 3; ++w)
                    workers.emplace_back([&, w] { for (int k = w; k < 5; k += 3) mods[(size_t)k].filter(data[row], zt, two); });
                for (auto& t : workers) t.join();
                tx = 0.0;
                for (int k = 0; k < 5; ++k) { tll += mods[(size_t)k].getLogCondLike() / 5.0; tx += mods[(size_t)k].getExpectations()[1](0, 0) / 5.0; }
            }
            std::printf("tswarm %.17g\ntswarmx %.17g\n", tll, tx);
            // a member given another observation than the rest of its swarm
            try {
                mods[0].filter(data[5], vec1{data[4].v}, two);
                mods[1].filter(data[6], vec1{data[4].v}, two);
                std::printf("mismatch no-throw\n");
            } catch (const std::invalid_argument&) { std::printf("mismatch rejected\n"); }
        }
        std::vector<unmodified_swarm_nocov::state_parm_func> gs;
        gs.push_back([](const vec1& xt, const shape::vec4<double>&) -> const Mat { vec1 ans; ans(0) = xt(0); return ans; });
        unmodified_swarm_nocov un(gs);
        un.opt = o;
        un.opt.declared_functionals = {SSME_H_X};
        double unll = 0.0;
        for (size_t row = 0; row < 4; ++row) { un.update(data[row]); unll += un.getLogCondLike(); }
        std::printf("uswarm_nocov %.17g\n", unll);
        unmodified_swarm_nocov_ctx cn(gs);
        ssme_gpu::gpu_options oc = o;
        oc.declared_functionals = {SSME_H_X};
        cn.ctx = std::make_shared<bs_mod::context>(SSME_MODEL_SVOL, 400, 3, oc);
        double cnll = 0.0;
        for (size_t row = 0; row < 4; ++row) { cn.update(data[row]); cnll += cn.getLogCondLike(); }
        std::printf("cswarm_nocov %.17g\ncswarm_nocov_x %.17g\nuswarm_nocov_x %.17g\n", cnll, cn.getExpectations()[0](0, 0), un.getExpectations()[0](0, 0));
    }
    // (4) error mapping
    try { std::vector<vec1> empty; ssme_gpu::log_like_eval_gpu(theta, empty, 100, 1, o); std::printf("no-throw\n"); }
    catch (const std::length_error&) { std::printf("length_error ok\n"); }
    return 0;
}
