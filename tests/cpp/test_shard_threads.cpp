// test_shard_threads.cpp -- the C++ shard drivers with SEVERAL ranks on one GPU: every rank is a host thread with its own
// handle, the "RCCL" underneath is tests/cpp/mock_rccl.cpp (linked before anything else, so dlsym finds it).  Prints, per
// configuration, what every rank got and what the unsharded filter gives; tests/test_sharded_gpu.py compares.
//   usage: test_shard_threads CSV WORLD N T MODEL RESAMPLER MODE SEED [TAU [YSCALE [RESAMP_SCHED [LW_FORM]]]]     (MODEL -1: Liu-West, RESAMPLER = delta x 1000)
//   TAU (linear-Gaussian model only): observation noise; a tiny value puts all the weight of a step on the one particle next to
//   y_t, so every rank's next resampling window is that particle's tile -- far ranks leave their halo, near ranks do not.
//   YSCALE: the observations are multiplied by it (outliers: the stochastic-volatility weights then degenerate the same way).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "../../include/ssme_pf.h"

static void die(const char* what, int rc, const char* msg) { std::fprintf(stderr, "%s: status %d (%s)\n", what, rc, msg ? msg : ""); std::exit(3); }

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int world = std::atoi(argv[2]), N = std::atoi(argv[3]), T = std::atoi(argv[4]), model = std::atoi(argv[5]), rs = std::atoi(argv[6]),
              mode = std::atoi(argv[7]);
    const unsigned long long seed = std::strtoull(argv[8], nullptr, 10);
    const int sched = argc > 11 ? std::atoi(argv[11]) : 1;
    const int lw_form = argc > 12 ? std::atoi(argv[12]) : 0;
    std::vector<double> y, z;
    { std::ifstream f(argv[1]); double v; while (f >> v && (int)y.size() < T) y.push_back(v); }
    if (argc > 10) for (double& v : y) v *= std::atof(argv[10]);
    z.assign(y.size(), 0.0);
    for (size_t t = 1; t < y.size(); ++t) z[t] = y[t - 1];
    const double th_svol[3] = {1.0, 0.95, 0.25}, th_lev[4] = {0.9, 0.0, 1.0, -0.1}, th_lg[3] = {0.9, 0.5, argc > 9 ? std::atof(argv[9]) : 0.7};
    char id[128];
    ssme_shard_comm_get_unique_id(id);
    std::vector<double> ll(world, 0.0);
    std::vector<int> path(world, 0), own_flag(world, 0), any_flag(world, 0);
    std::vector<long long> exch(world, 0);
    std::vector<std::vector<double>> xs(world);
    std::vector<size_t> first(world, 0);
    std::vector<std::thread> ranks;
    for (int r = 0; r < world; ++r) ranks.emplace_back([&, r] {
        void* comm = nullptr;
        int rc = ssme_shard_comm_init(id, r, world, 0, &comm);
        if (rc) die("comm_init", rc, "");
        int32_t lay[4] = {0, 0, 0, 0};                 // {B, Bl, tiles this rank owns, particles this rank owns}
        if (model >= 0) {
            ssme_pf_config c{};
            c.model = model; c.n_particles = N; c.n_filters = 1; c.dtype = SSME_F64; c.resampler = rs; c.resamp_sched = sched; c.seed = seed; c.device = 0;
            ssme_pf_handle h = nullptr;
            rc = ssme_pf_shard_create(&c, r, world, &h);
            if (rc) die("shard_create", rc, "");
            if (ssme_pf_shard_layout(h, lay)) die("shard_layout", 1, "");
            xs[r].resize((size_t)lay[3]); first[r] = (size_t)r * lay[1] * 2048;
            rc = ssme_pf_set_params(h, model == 0 ? th_svol : (model == 1 ? th_lev : th_lg), model == 1 ? 4 : 3, 1);
            if (rc) die("set_params", rc, ssme_pf_last_error(h));
            rc = ssme_pf_shard_run_series(h, comm, y.data(), model == 1 ? z.data() : nullptr, T, mode, &ll[r]);
            if (rc) die("shard_run_series", rc, ssme_pf_last_error(h));
            int32_t p = 0; int64_t e = 0;
            ssme_pf_shard_download(h, xs[r].data(), nullptr, &p, &e);
            path[r] = p; exch[r] = e;
            int32_t st[4] = {0, 0, 0, 0};
            ssme_pf_shard_stats(h, st);
            any_flag[r] = st[0]; own_flag[r] = st[1];
            ssme_pf_destroy(h);
        } else {
            ssme_lw_config c{};
            c.n_particles = N; c.n_filters = 1; c.seed = seed; c.device = 0; c.delta = rs / 1000.0; c.form = lw_form; c.resamp_sched = sched;
            const int tr[4] = {2, 0, 3, 1};
            const double lo[4] = {0.8, -0.1, 0.01, -0.5}, hi[4] = {0.99, 0.1, 0.1, -0.01};
            for (int d = 0; d < 4; ++d) { c.transforms[d] = tr[d]; c.prior_lo[d] = lo[d]; c.prior_hi[d] = hi[d]; }
            ssme_lw_handle h = nullptr;
            rc = ssme_lw_shard_create(&c, r, world, &h);
            if (rc) die("lw_shard_create", rc, "");
            if (ssme_lw_shard_layout(h, lay)) die("lw_shard_layout", 1, "");
            xs[r].resize((size_t)lay[3]); first[r] = (size_t)r * lay[1] * 2048;
            rc = ssme_lw_shard_run_series(h, comm, y.data(), z.data(), T, &ll[r]);
            path[r] = rc == SSME_ERR_STATE ? 2 : 1;                        // 2: a window left the halo (caller falls back)
            if (rc && rc != SSME_ERR_STATE) die("lw_shard_run_series", rc, ssme_lw_last_error(h));
            int64_t e = 0;
            if (!rc) ssme_lw_shard_download(h, xs[r].data(), nullptr, &e);
            exch[r] = e;
            int32_t st[4] = {0, 0, 0, 0};
            ssme_lw_shard_stats(h, st);
            any_flag[r] = st[0]; own_flag[r] = st[1];
            ssme_lw_destroy(h);
        }
        ssme_shard_comm_destroy(comm);
    });
    for (auto& t : ranks) t.join();
    // the unsharded filter with the same N and seed
    double ll_ref = 0.0;
    std::vector<double> xref(N);
    if (model >= 0) {
        ssme_pf_config c{};
        c.model = model; c.n_particles = N; c.n_filters = 1; c.dtype = SSME_F64; c.resampler = rs; c.resamp_sched = sched; c.seed = seed; c.device = 0;
        c.tile_particles = 2048;
        ssme_pf_handle h = nullptr;
        if (ssme_pf_create(&c, &h)) die("create", 1, "");
        ssme_pf_set_params(h, model == 0 ? th_svol : (model == 1 ? th_lev : th_lg), model == 1 ? 4 : 3, 1);
        ssme_pf_run_series(h, y.data(), model == 1 ? z.data() : nullptr, T, &ll_ref);
        ssme_pf_download_state(h, 0, xref.data(), nullptr, nullptr, nullptr);
        ssme_pf_destroy(h);
    } else {
        ssme_lw_config c{};
        c.n_particles = N; c.n_filters = 1; c.seed = seed; c.device = 0; c.delta = rs / 1000.0; c.form = lw_form; c.resamp_sched = sched;
        const int tr[4] = {2, 0, 3, 1};
        const double lo[4] = {0.8, -0.1, 0.01, -0.5}, hi[4] = {0.99, 0.1, 0.1, -0.01};
        for (int d = 0; d < 4; ++d) { c.transforms[d] = tr[d]; c.prior_lo[d] = lo[d]; c.prior_hi[d] = hi[d]; }
        ssme_lw_handle h = nullptr;
        if (ssme_lw_create(&c, &h)) die("lw_create", 1, "");
        ssme_lw_run_series(h, y.data(), z.data(), T, &ll_ref);
        ssme_lw_download_state(h, 0, xref.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
        ssme_lw_destroy(h);
    }
    size_t mism = 0, held = 0;
    for (int r = 0; r < world; ++r) {
        held += xs[r].size();
        if (path[r] == 1 || model >= 0)
            for (size_t i = 0; i < xs[r].size(); ++i) mism += xs[r][i] != xref[first[r] + i];
    }
    if (held != (size_t)N) mism += 1;                  // the ranks' shares add up to the filter
    std::printf("ref %.17g\n", ll_ref);
    // any_left_halo: the reduced flag of the last fixed-halo pass (identical on every rank); own_left_halo: what this rank's own workgroups saw
    for (int r = 0; r < world; ++r) std::printf("rank %d ll %.17g path %d exchanged %lld any_left_halo %d own_left_halo %d\n", r, ll[r], path[r], exch[r], any_flag[r], own_flag[r]);
    std::printf("particle_mismatches %zu\n", mism);
    return 0;
}
