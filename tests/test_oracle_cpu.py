"""CPU suite: pins the oracle (oracle/) against every known-answer value the reference's own
files hold for this path, against independent anchors, and against the committed golden vectors."""
import os

import numpy as np
import pytest


# ---- Philox4x32-10: Random123 known-answer vectors -------------------------------------------------
@pytest.mark.parametrize("ctr,key,exp", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_kat(oracle, ctr, key, exp):
    assert [int(v) for v in oracle.philox(ctr, key)] == exp


# ---- libm-free math vs glibc -------------------------------------------------------------------------
def _ulp(a, b):
    return np.max(np.abs(a - b) / np.spacing(np.abs(b)))


def test_math_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = rng.uniform(-700, 700, 100000)
    assert _ulp(oracle.exp(x), np.exp(x)) <= 1.0
    x = rng.uniform(-3, 3, 100000)
    assert _ulp(oracle.exp(x), np.exp(x)) <= 1.0
    for lo, hi in ((-700, 700), (-3, 3), (-45, 5)):            # the bootstrap filter's table form
        x = rng.uniform(lo, hi, 100000)
        assert _ulp(oracle.exp_t(x), np.exp(x)) <= 1.0
    u = rng.uniform(0, 1, 100000)
    assert _ulp(oracle.log(u), np.log(u)) <= 1.0
    x = np.exp(rng.uniform(-700, 700, 100000))
    assert _ulp(oracle.log(x), np.log(x)) <= 1.0
    s, c = oracle.sincos2pi(u)
    # libm sin(2*pi*u) carries the rounding of 2*pi*u; compare absolutely
    assert np.max(np.abs(s - np.sin(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(c - np.cos(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(s * s + c * c - 1)) < 5e-16
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.dps = 40
    for uv, sv, cv in zip(u[:1500], s[:1500], c[:1500]):
        a = 2 * mpmath.pi * mpmath.mpf(float(uv))
        ts, tc = mpmath.sin(a), mpmath.cos(a)
        assert abs(ts - mpmath.mpf(float(sv))) <= mpmath.mpf(float(np.spacing(abs(float(ts)))))
        assert abs(tc - mpmath.mpf(float(cv))) <= mpmath.mpf(float(np.spacing(abs(float(tc)))))


def test_log_of_uniform_accuracy(oracle):
    """The table-based log of the hot-loop draws: strictly negative on (0,1), absolute error < 2^-51 where |log| < 1
    and <= 2 ulp of the result elsewhere (its relative error next to x = 1 is NOT bounded: never needed, u <= 1 - 2^-41)."""
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.uniform(0, 1, 400000), (rng.integers(0, 2 ** 32, 100000) + 0.5) * 2.0 ** -32,
                        1 - (rng.integers(0, 2 ** 20, 100000) + 0.5) * 2.0 ** -40, np.exp(rng.uniform(-40, 0, 100000))])
    got, ref = oracle.log_u(u), np.log(u)
    assert np.all(got < 0)
    err = np.abs(got - ref)
    assert np.all(err <= np.maximum(2.0 ** -51, 2 * np.spacing(np.abs(ref))))


def test_round3_draw_functions_accuracy(oracle):
    """The spec-r3 forms of the hot-loop draws, checked independently of the device (mpmath, 40 digits):
    * sin / cos of the 24-bit Box-Muller angle by table: EVERY one of the 2^24 angles against libm (|error| < 4e-16 with libm's own
      argument rounding), a sample against mpmath (|error| <= 3e-16), sin^2 + cos^2 = 1 to 5e-16;
    * the spacing log: absolute error < 2^-43 (the spacings are quantised to 2^-35), strictly negative on the 32-bit grid."""
    k = np.arange(0, 1 << 24, dtype=np.float64)
    s, c = oracle.sincos_k24(k)
    a = 2 * np.pi * k / 2.0 ** 24
    assert np.max(np.abs(s - np.sin(a))) < 2e-15 and np.max(np.abs(c - np.cos(a))) < 2e-15
    assert np.max(np.abs(s * s + c * c - 1)) < 5e-16
    assert s[0] == 0.0 and c[0] == 1.0 and c[1 << 23] == -1.0 and s[1 << 22] == 1.0       # table angles come out exactly
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.dps = 40
    rng = np.random.default_rng(5)
    for kv in np.concatenate([rng.integers(0, 1 << 24, 1500), [131071, 131072, 131073, (1 << 24) - 1]]):
        ang = 2 * mpmath.pi * int(kv) / 2 ** 24
        sv, cv = oracle.sincos_k24(np.array([float(kv)]))
        assert abs(mpmath.sin(ang) - mpmath.mpf(float(sv[0]))) <= 3e-16 and abs(mpmath.cos(ang) - mpmath.mpf(float(cv[0]))) <= 3e-16
    u = np.concatenate([(rng.integers(0, 2 ** 32, 400000) + 0.5) * 2.0 ** -32, (np.arange(0, 4096) + 0.5) * 2.0 ** -32,
                        1 - (np.arange(0, 4096) + 0.5) * 2.0 ** -32])
    got = oracle.log_u32(u)
    assert np.all(got < 0)
    assert np.max(np.abs(got - np.log(u))) < 2.0 ** -43
    # the quantised spacing differs from the one the full-precision log would give by at most one unit of 2^-35, rarely
    q32, q = np.rint(-got * 2.0 ** 35), np.rint(-oracle.log_u(u) * 2.0 ** 35)
    assert np.max(np.abs(q32 - q)) <= 1 and np.mean(q32 != q) < 0.01


def test_math_special_values(oracle):
    with np.errstate(all="ignore"):
        x = np.array([-745.2, -745.0, -720.0, 709.7, 709.9, 0.0, np.inf, -np.inf])
        np.testing.assert_array_equal(oracle.exp(x), np.exp(x))
        x = np.array([0.0, 1.0, 5e-324, 1e-310, np.inf])
        np.testing.assert_allclose(oracle.log(x), np.log(x), rtol=2e-16)
        assert np.isnan(oracle.log(np.array([-1.0, np.nan]))).all()
    assert oracle.exp(np.array([np.nan]))[0] == 0.0      # the clamp squashes NaN
    s, c = oracle.sincos2pi(np.array([0.0, 0.25, 0.5, 0.75]))
    np.testing.assert_array_equal(s, [0.0, 1.0, -0.0, -1.0])
    np.testing.assert_array_equal(c, [1.0, -0.0, -1.0, 0.0])


def test_normals_moments(oracle):
    z = oracle.normals(7, 0, 3, 400000)
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1) < 1e-2
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 0.05


# ---- reference KATs ------------------------------------------------------------------------------------
def test_pack_transform_kats(oracle):
    """test/test_parameters.cpp:114-120,141-145: trans {1.0,-1.3,9.5,.89} x {null,log,logit,twice_fisher}."""
    NULL, TF, LOGIT, LOG = 0, 1, 2, 3
    got = [oracle.inv_transform(NULL, 1.0), oracle.inv_transform(LOG, -1.3), oracle.inv_transform(LOGIT, 9.5),
           oracle.inv_transform(TF, 0.89)]
    np.testing.assert_allclose(got, [1.0, 0.2725318, 0.9999252, 0.4177803], atol=1e-4)
    lj = (oracle.log_jacobian(NULL, 1.0) + oracle.log_jacobian(LOG, -1.3) + oracle.log_jacobian(LOGIT, 9.5)
          + oracle.log_jacobian(TF, 0.89))
    assert abs(lj - (-11.6851)) < 1e-4


def test_log_mean_exp_kat(oracle):
    """test/test_thread_pool.cpp:39-46: 10^4 results of 3.0 -> 3.0 +- 1e-3."""
    assert abs(oracle.log_mean_exp(np.full(10000, 3.0)) - 3.0) < 1e-3


def test_chain_start_point(oracle):
    """estimate_univ_svol.h:153-155: theta_trans = (1, twiceFisher(.5), log 2e-4) -> beta 1, phi .5, ss 2e-4."""
    tf = np.log(1.5) - np.log(0.5)
    assert abs(oracle.inv_transform(1, tf) - 0.5) < 1e-12
    assert abs(oracle.inv_transform(3, np.log(2e-4)) - 2e-4) < 1e-15


def test_constant_functional_is_42(oracle, spy):
    """test/test_pswarm.cpp:252: expectation of the constant 42 is 42."""
    f = oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, 10, [0.9, 0.0, 1.0, -0.1], 1)
    f.step(spy[0], 0.0)
    assert abs(f.expectation(3) - 42.0) < 1e-4
    assert f.loglik ** 2 > 0       # :251


# ---- exact fixed-point cdf and Gamma draws -------------------------------------------------------------
def test_quantize_and_exact_cdf(oracle, spy):
    """q = rne(exp(x) 2^r): integer, <= 2^r, monotone in x; the cdf is the exact integer prefix sum."""
    rng = np.random.default_rng(3)
    x = -np.sort(rng.exponential(5.0, 4000))
    x[0] = 0.0
    for r in (30, 38, 41):
        q = oracle.quantize(x, r)
        assert q.dtype == np.uint64 and q[0] == (1 << r) and (np.diff(q.astype(np.int64)) <= 0).all()
        np.testing.assert_allclose(q.astype(np.float64) * 2.0 ** -r, np.exp(x), atol=2.0 ** -r)
    assert oracle.quantize(np.array([-800.0, np.nan, -np.inf]), 41).tolist() == [0, 0, 0]
    f = oracle.Filter(oracle.MODEL_SVOL, 5000, [1.0, 0.95, 0.25], 3)
    for t in range(4):
        f.step(spy[t])
    st = f.state()
    # per-tile scales: q = rne(exp(logw - m_tile) 2^41), exact tile-local prefix sums
    tiles, sums = [], []
    assert f.tile == 512                                   # 2048 < N <= 2^18: 512-particle tiles
    for b, i in enumerate(range(0, 5000, f.tile)):
        lw = st["logw"][i:i + f.tile]
        assert st["mb"][b] == lw.max()
        c = np.cumsum(oracle.quantize(lw - st["mb"][b], 41), dtype=np.uint64)
        tiles.append(c)
        sums.append(c[-1])
    np.testing.assert_array_equal(st["cdf"], np.concatenate(tiles))
    np.testing.assert_array_equal(st["A"], sums)
    assert st["m"] == st["mb"].max() and st["rshift"] == 52 - 13           # Npad = 10 tiles x 512 = 5120 <= 2^13
    Ap = oracle.rescale(st["A"], st["mb"] - st["m"], st["rshift"] - 41)
    assert st["S"] == int(sum(int(v) for v in Ap))
    # log-sum-exp from the integers agrees with the floating-point one to ~1e-13
    lse_fp = st["m"] + np.log(np.exp(st["logw"] - st["m"]).sum())
    assert abs(st["m"] + np.log(st["S"] * 2.0 ** -st["rshift"]) - lse_fp) < 1e-12
    # ancestors of the (sorted) multinomial resampler are sorted and in range
    a = st["anc"].astype(np.int64)
    assert (np.diff(a) >= 0).all() and a.max() < 5000


def test_gamma_draws_moments(oracle):
    for shape in (1.0, 7.0, 2048.0):
        g = oracle.gamma_draws(11, 2, 5, shape, 40000)
        assert abs(g.mean() - shape) < 5 * np.sqrt(shape / 40000)
        assert abs(g.var() / shape - 1) < 0.05
    # counter based: same (seed, filter, t, tile) -> same draw; another t -> another draw
    np.testing.assert_array_equal(oracle.gamma_draws(11, 2, 5, 2048.0, 8), oracle.gamma_draws(11, 2, 5, 2048.0, 8))
    assert (oracle.gamma_draws(11, 2, 5, 2048.0, 8) != oracle.gamma_draws(11, 2, 6, 2048.0, 8)).all()


def test_multinomial_sorted_uniforms_are_uniform(oracle):
    """Gamma-per-tile exponential spacings give sorted U(0,1) order statistics: with unit weights the
    ancestor counts are multinomial(N; 1/N) -- mean 1, variance (1 - 1/N)."""
    n = 6000
    f = oracle.Filter(oracle.MODEL_LIN_GAUSS, n, [0.0, 1.0, 1e6], 4)      # tau huge -> equal weights
    counts = []
    for t in range(12):
        f.step(0.0)
        if t:
            counts.append(np.bincount(f.state()["anc"], minlength=n))
    c = np.concatenate(counts).astype(np.float64)
    assert abs(c.mean() - 1.0) < 1e-12 and abs(c.var() - 1.0) < 0.03
    # P(count = 0) = (1 - 1/N)^N ~ e^-1 for multinomial (systematic would give 0)
    assert abs((c == 0).mean() - np.exp(-1)) < 0.01


# ---- estimator anchors -----------------------------------------------------------------------------------
def _lg_data(T=150, phi=0.9, sig=0.5, tau=0.7, seed=5):
    rng = np.random.default_rng(seed)
    x = np.zeros(T)
    x[0] = rng.normal() * sig / np.sqrt(1 - phi ** 2)
    for t in range(1, T):
        x[t] = phi * x[t - 1] + sig * rng.normal()
    return x + tau * rng.normal(size=T), (phi, sig, tau)


@pytest.mark.parametrize("resampler", [0, 1, 2, 3])
def test_kalman_anchor(oracle, resampler):
    """The particle estimate of the log-likelihood of a linear-Gaussian model is within MC error of Kalman."""
    y, th = _lg_data()
    exact, _ = oracle.kalman_loglik(*th, y)
    lls = np.array([oracle.Filter(oracle.MODEL_LIN_GAUSS, 4096, th, s, resampler=resampler).run_series(y)[0]
                    for s in range(8)])
    se = lls.std(ddof=1) / np.sqrt(len(lls))
    assert abs(lls.mean() - exact) < 4 * se + 0.15


def test_mode_a_vs_mode_b(oracle, spy):
    """Reference-faithful mt19937 mode (A) and kernel-matched Philox mode (B) have the same mean log-likelihood:
    >= 200 seeds each, 3 SE, no additive slack, svol_bs at the shipped N = 500 on the first 300
    rows of spy_returns.csv; both resamplers of the reference (discrete_distribution and the sorted-uniform one)."""
    import stat_anchor as sa
    th = [1.0, 0.95, 0.25]
    y = spy[:300]
    a = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 500, y)
    b = np.array(sa.pmap(lambda s: oracle.Filter(oracle.MODEL_SVOL, 500, th, 1000 + s).run_series(y)[0], range(sa.SEEDS)))
    sa.assert_same_mean(a, b, "svol_bs mode A vs mode B")
    af = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 500, y, seed0=5000, fast_resampler=True)
    sa.assert_same_mean(a, af, "svol_bs mn_resampler vs mn_resamp_fast1")


def test_mode_a_vs_mode_b_leverage(oracle):
    """The same anchor for svol_leverage (test/test_pswarm.cpp:80-134) with the covariate z_t = y_{t-1}."""
    import stat_anchor as sa
    th = [0.95, 0.0, 0.2, -0.3]
    y, z = sa.sim_leverage(300, *th, seed=11)
    a = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL_LEVERAGE, th, 500, y, z)
    b = np.array(sa.pmap(lambda s: oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, 500, th, 1000 + s).run_series(y, z)[0],
                         range(sa.SEEDS)))
    sa.assert_same_mean(a, b, "svol_leverage mode A vs mode B")


def test_mode_a_vs_mode_b_liu_west(oracle):
    """The same anchor for the Liu-West filter: log-likelihood and the posterior means of all four parameters."""
    import stat_anchor as sa
    y, z = sa.sim_leverage(100, 0.95, 0.0, 0.05, -0.3, seed=9)
    a = sa.mode_a_liu_west(oracle, 2000, y, z)

    def one(s):
        o = oracle.LWFilter(2000, 1000 + s)
        for t in range(y.size):
            o.step(y[t], z[t])
        return (o.loglik,) + tuple(o.param_means())
    b = np.array(sa.pmap(one, range(sa.SEEDS)))
    sa.assert_same_mean(a, b, "Liu-West mode A vs mode B (loglik, phi, mu, sigma, rho)")


@pytest.mark.parametrize("form,rs", [(1, 1), (0, 3), (1, 3)])
def test_mode_a_vs_mode_b_liu_west_forms(oracle, form, rs):
    """SISR form (LWFilter2WithCovs) and resampling schedules: the kernel-matched restatement has the law of the
    reference-faithful one (200 seeds, 3 SE; log-likelihood and the four posterior means)."""
    import stat_anchor as sa
    y, z = sa.sim_leverage(100, 0.95, 0.0, 0.05, -0.3, seed=9)

    def ma(s):
        l, _, m = oracle.lw_ref_run(2000, y, z, seed=1 + s, form=form, resamp_sched=rs)
        return (l,) + tuple(m)

    def mb(s):
        o = oracle.LWFilter(2000, 1000 + s, form=form, resamp_sched=rs)
        for t in range(y.size):
            o.step(y[t], z[t])
        return (o.loglik,) + tuple(o.param_means())
    sa.assert_same_mean(np.array(sa.pmap(ma, range(sa.SEEDS))), np.array(sa.pmap(mb, range(sa.SEEDS))), f"Liu-West form {form} rs {rs}")


def test_variance_scales_with_n(oracle, spy):
    th = [1.0, 0.95, 0.25]
    y = spy[:200]
    v = []
    for n in (128, 2048):
        v.append(np.var([oracle.Filter(oracle.MODEL_SVOL, n, th, s).run_series(y)[0] for s in range(16)], ddof=1))
    assert v[1] < v[0]


def test_resample_schedule_and_replicates(oracle, spy):
    th = [1.0, 0.95, 0.25]
    f = oracle.Filter(oracle.MODEL_SVOL, 300, th, 1, resamp_sched=3)
    ll, per = f.run_series(spy[:50])
    assert np.isfinite(ll) and abs(per.sum() - ll) < 1e-9
    a = oracle.Filter(oracle.MODEL_SVOL, 300, th, 1, rep=0).run_series(spy[:50])[0]
    b = oracle.Filter(oracle.MODEL_SVOL, 300, th, 1, rep=1).run_series(spy[:50])[0]
    assert a != b


def test_degenerate_inputs(oracle):
    # beta <= 0 -> logG = -inf for every particle -> NaN log-lik (reference: PMMH rejects, ada_pmmh_mvn.h:349)
    f = oracle.Filter(oracle.MODEL_SVOL, 64, [-1.0, 0.5, 0.1], 1)
    assert np.isnan(f.step(0.3))
    # |phi| >= 1 -> stationary sd NaN -> NaN
    f = oracle.Filter(oracle.MODEL_SVOL, 64, [1.0, 1.5, 0.1], 1)
    assert np.isnan(f.step(0.3))
    # y = 0 is legal (11 exact zeros in spy_returns.csv)
    f = oracle.Filter(oracle.MODEL_SVOL, 64, [1.0, 0.5, 0.1], 1)
    assert np.isfinite(f.step(0.0))
    # N = 1 and N not a multiple of anything
    for n in (1, 3, 2047, 2049):
        f = oracle.Filter(oracle.MODEL_SVOL, n, [1.0, 0.9, 0.2], 2)
        assert np.isfinite(f.run_series(np.array([0.1, -0.2, 0.3]))[0])


# ---- golden vectors ------------------------------------------------------------------------------------
@pytest.mark.parametrize("tname", ["start", "real"])
@pytest.mark.parametrize("n", [64, 500, 4096])
@pytest.mark.parametrize("rs", [("mn", 0), ("sys", 1)])
def test_oracle_reproduces_golden(oracle, golden, spy, tname, n, rs):
    th = golden[f"theta_{tname}"]
    f = oracle.Filter(oracle.MODEL_SVOL, n, th, int(golden["seed"][0]), resampler=rs[1])
    lls = [f.step(spy[t]) for t in range(8)]
    k = f"svol_{tname}_n{n}_{rs[0]}"
    np.testing.assert_array_equal(np.array(lls), golden[k + "_ll"])
    st = f.state()
    for name in ("x", "logw", "cdf", "anc"):
        np.testing.assert_array_equal(st[name], golden[k + "_" + name])


def test_oracle_full_series_golden(oracle, golden, spy):
    for tname in ("start", "real"):
        f = oracle.Filter(oracle.MODEL_SVOL, 500, golden[f"theta_{tname}"], int(golden["seed"][0]))
        ll, per = f.run_series(spy)
        assert ll == golden[f"svol_{tname}_n500_full_ll"][0]
        np.testing.assert_array_equal(per, golden[f"svol_{tname}_n500_full_per"])
    assert len(spy) == 3084


# ---- Liu-West oracle (liu_west_filter.h:971-1159 restated; test_liu_west.cpp model) --------------------------------
def _sim_leverage(T, phi, mu, sig, rho, seed):
    rng = np.random.default_rng(seed)
    x, y = np.zeros(T), np.zeros(T)
    x[0] = mu + sig / np.sqrt(1 - phi * phi) * rng.normal()
    y[0] = np.exp(x[0] / 2) * rng.normal()
    for t in range(1, T):
        x[t] = (mu + phi * (x[t - 1] - mu) + rho * sig * y[t - 1] * np.exp(-x[t - 1] / 2)
                + sig * np.sqrt(1 - rho * rho) * rng.normal())
        y[t] = np.exp(x[t] / 2) * rng.normal()
    return y, np.concatenate([[0.0], y[:-1]])


def test_liu_west_mode_a_vs_mode_b(oracle):
    """The kernel-matched Liu-West oracle (Philox, exact cdf) and the reference-faithful restatement (mt19937,
    discrete_distribution) estimate the same log-likelihood and posterior means within Monte-Carlo error."""
    y, z = _sim_leverage(120, 0.95, 0.0, 0.05, -0.3, 4)
    n, reps = 4000, 6
    la, ma = [], []
    for s in range(reps):
        ll, per, means = oracle.lw_ref_run(n, y, z, seed=100 + s)
        assert np.isfinite(ll) and abs(per.sum() - ll) < 1e-9
        la.append(ll); ma.append(means)
    lb, mb = [], []
    for s in range(reps):
        f = oracle.LWFilter(n, 7, rep=s)
        lb.append(sum(f.step(y[t], z[t]) for t in range(y.size)))
        mb.append(f.param_means())
    la, lb, ma, mb = np.array(la), np.array(lb), np.array(ma), np.array(mb)
    se = np.sqrt(la.var(ddof=1) / reps + lb.var(ddof=1) / reps)
    assert abs(la.mean() - lb.mean()) < 4 * se + 0.3
    sem = np.sqrt(ma.var(0, ddof=1) / reps + mb.var(0, ddof=1) / reps)
    assert np.all(np.abs(ma.mean(0) - mb.mean(0)) < 4 * sem + [0.01, 0.01, 0.005, 0.03])
    # parameters stay inside the support the transforms impose (test_liu_west.cpp:165 priors)
    assert np.all((mb[:, 0] > 0) & (mb[:, 0] < 1) & (mb[:, 2] > 0) & (np.abs(mb[:, 3]) < 1))


def test_liu_west_oracle_deterministic_and_replicates_differ(oracle):
    y, z = _sim_leverage(10, 0.95, 0.0, 0.05, -0.3, 5)
    def run(rep):
        f = oracle.LWFilter(700, 3, rep=rep)
        return [f.step(y[t], z[t]) for t in range(y.size)], f.state()
    a, sa_ = run(0); b, sb = run(0); c, _ = run(1)
    assert a == b and a != c
    assert np.array_equal(sa_["kidx"], sb["kidx"]) and np.array_equal(sa_["anc"], sb["anc"])
    assert sa_["kidx"].max() < 700 and sa_["anc"].max() < 700
    assert np.all(np.diff(sa_["anc"].astype(np.int64)) >= 0)        # sorted-uniform multinomial draws come out ordered
    L = sa_["L"]
    assert np.all(np.diag(L) > 0) and np.allclose(np.triu(L, 1), 0)


def test_liu_west_delta_one_freezes_parameters(oracle):
    """delta = 1 => a = 1, h^2 = 0: no shrinkage and no jitter, parameters are only resampled (liu_west_filter.h:960,1195)."""
    y, z = _sim_leverage(5, 0.95, 0.0, 0.05, -0.3, 6)
    f = oracle.LWFilter(500, 9, delta=1.0)
    f.step(y[0], z[0])
    th0 = f.state()["theta"].copy()
    f.step(y[1], z[1])
    s = f.state()
    for d in range(4):
        assert set(np.unique(s["theta"][d]).tolist()) <= set(np.unique(th0[d]).tolist())


# ---- oracle/_ref: the reference's own thread_pool.h (the one hot-path header that builds here) -----------------------
def test_log_mean_exp_pinned_by_reference_thread_pool(oracle):
    """thread_pool<>::work (thread_pool.h:189-215,263-268) computes the replicate log-mean-exp; the oracle's and the
    product host code's versions must agree with it.  Skipped only if oracle/_ref could not be built or shipped."""
    if oracle.build_ref() is None:
        pytest.skip("oracle/_ref not available (no /root/reference and no prebuilt library)")
    from ssme_amd.parallel import log_mean_exp
    assert abs(oracle.ref_log_mean_exp(np.full(10000, 3.0)) - 3.0) < 1e-3        # the reference's own KAT, test_thread_pool.cpp:39-46
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 100, 1000):
        for scale in (1e-3, 1.0, 50.0, 700.0):
            v = -4700.0 + scale * rng.standard_normal(n)
            want = oracle.ref_log_mean_exp(v)
            assert abs(oracle.log_mean_exp(v) - want) <= 1e-12 * abs(want)
            assert abs(log_mean_exp(v) - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("n", [300, 5000])
@pytest.mark.parametrize("delta", [0.99, 0.9])
def test_liu_west_oracle_reproduces_golden(oracle, n, delta):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "liu_west_golden.npz"))
    f = oracle.LWFilter(n, int(g["seed"][0]), rep=1, delta=delta)
    lls = [f.step(g["y"][t], g["z"][t]) for t in range(g["y"].size)]
    k = f"lw_n{n}_d{int(round(delta * 100))}"
    np.testing.assert_array_equal(np.array(lls), g[k + "_ll"])
    st = f.state()
    for name in ("x", "theta", "kidx", "anc", "thetabar", "L"):
        np.testing.assert_array_equal(st[name], g[k + "_" + name])
