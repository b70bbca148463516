"""GPU parity suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs and against the committed golden vectors.

Bar: particles, log-weights, the exact integer cdf and ANCESTOR INDICES bit-exact;
per-step and series log-likelihood |delta| <= 1e-9 (observed 0: the device math mirrors the oracle's
IEEE operation sequence).  At BASELINE sizes: size-independent properties (Kalman anchor, replicate
independence, graph == eager, step API == series API)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

TOL_LL = 1e-9


@pytest.fixture(scope="module")
def sa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import ssme_amd
    from ssme_amd import _capi
    assert _capi.lib() is not None        # the in-tree HIP library is what runs
    return ssme_amd


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern differs"
    bad = (_bits(a) != _bits(b)) & ~nan_a
    assert not bad.any(), f"{what}: {bad.sum()} of {a.size} values differ, first at {np.argmax(bad)}: " \
                          f"{a.flat[np.argmax(bad)]!r} vs {b.flat[np.argmax(bad)]!r}"


# ---- device primitives ----------------------------------------------------------------------------
def test_device_philox_kat(sa, oracle):
    from ssme_amd import _capi
    for ctr, key in (([0, 0, 0, 0], [0, 0]), ([0xffffffff] * 4, [0xffffffff] * 2),
                     ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]),
                     ([5, 77, 3, 1], [20260101, 9])):
        c, k, o = np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
        _capi.check(_capi.lib().ssme_pf_test_philox(0, _capi.u32ptr(c), _capi.u32ptr(k), _capi.u32ptr(o)))
        np.testing.assert_array_equal(o, oracle.philox(ctr, key))


def _dev_math(fn, x):
    from ssme_amd import _capi
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _capi.check(_capi.lib().ssme_pf_test_math(0, fn, _capi.dptr(x), _capi.dptr(out), x.size))
    return out


def test_device_math_bit_exact(sa, oracle):
    rng = np.random.default_rng(11)
    with np.errstate(all="ignore"):
        x = np.concatenate([rng.uniform(-750, 715, 200000), rng.uniform(-40, 40, 200000),
                            [0.0, -0.0, np.inf, -np.inf, np.nan, 709.782712893384, -745.1332191019412, 1e-320, 710.0, -746.0]])
        assert_bits_equal(_dev_math(0, x), oracle.exp(x), "exp")
        assert_bits_equal(_dev_math(7, x), oracle.exp_t(x), "exp (table form)")
        u = np.concatenate([rng.uniform(0, 1, 200000), np.exp(rng.uniform(-740, 700, 200000)),
                            (np.arange(1, 4097) * 2.0 ** -53), [1.0, 0.0, np.inf, 5e-324, 1e-310, -1.0, np.nan]])
        assert_bits_equal(_dev_math(1, u), oracle.log(u), "log")
        un = u[(u > 2.3e-308) & np.isfinite(u)]
        assert_bits_equal(_dev_math(5, un), oracle.log(un), "log (normal-only core)")
        uu = np.concatenate([rng.uniform(0, 1, 400000), (np.arange(0, 4096) + 0.5) * 2.0 ** -32, 1 - (np.arange(0, 4096) + 0.5) * 2.0 ** -40,
                             np.exp(rng.uniform(-40, 0, 100000))])
        assert_bits_equal(_dev_math(6, uu), oracle.log_u(uu), "log of a uniform (table, no division)")
        v = np.concatenate([rng.uniform(0, 1, 400000), np.arange(0, 4096) * 2.0 ** -53, 1 - np.arange(1, 4097) * 2.0 ** -53,
                            [0.0, 0.125, 0.25, 0.5, 0.75, 0.875]])
        s, c = oracle.sincos2pi(v)
        assert_bits_equal(_dev_math(2, v), s, "sin2pi")
        assert_bits_equal(_dev_math(3, v), c, "cos2pi")
        w = np.concatenate([rng.uniform(0, 100, 200000), np.exp(rng.uniform(-700, 700, 100000)), [0.0, 4.0, 2.0, 1e-320]])
        assert_bits_equal(_dev_math(4, w), np.sqrt(w), "sqrt (IEEE, correctly rounded)")
        # round-3 draws: the Box-Muller angle by table (every 24-bit angle), the spacing log, the range-restricted sqrt
        k = np.concatenate([np.arange(0, 1 << 24, 5, dtype=np.float64), rng.integers(0, 1 << 24, 300000).astype(np.float64),
                            [0.0, 131071.0, 131072.0, 131073.0, 16777215.0, 4194304.0, 8388608.0, 12582912.0]])
        sk, ck = oracle.sincos_k24(k)
        assert_bits_equal(_dev_math(8, k), sk, "sin of the 24-bit angle (table)")
        assert_bits_equal(_dev_math(9, k), ck, "cos of the 24-bit angle (table)")
        u32 = np.concatenate([(rng.integers(0, 1 << 32, 400000).astype(np.float64) + 0.5) * 2.0 ** -32, (np.arange(0, 4096) + 0.5) * 2.0 ** -32,
                              1 - (np.arange(0, 4096) + 0.5) * 2.0 ** -32])
        assert_bits_equal(_dev_math(10, u32), oracle.log_u32(u32), "log of a spacing uniform")
        rad = np.concatenate([-2.0 * oracle.log_u(uu[(uu > 0) & (uu < 1)]), rng.uniform(2.0 ** -40, 60, 300000), np.exp(rng.uniform(-28, 4.1, 300000))])
        assert_bits_equal(_dev_math(11, rad), np.sqrt(rad), "sqrt of a Box-Muller radicand == IEEE sqrt")


def test_device_quantize_bit_exact(sa, oracle):
    from ssme_amd import _capi
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.exponential(8.0, 300000), [0.0, -0.0, -745.0, -800.0, -np.inf, np.nan]])
    for shift in (30, 38, 41):
        q = np.empty(x.size, dtype=np.uint64)
        _capi.check(_capi.lib().ssme_pf_test_quantize(0, _capi.dptr(x), shift, _capi.u64ptr(q), x.size))
        np.testing.assert_array_equal(q, oracle.quantize(x, shift))


@pytest.mark.parametrize("threads", [256, 512, 1024])
def test_device_block_scan_exact(sa, threads):
    """fp64 DPP wave scans + segment prefixes of integers == numpy's exact integer cumulative sum."""
    from ssme_amd import _capi
    rng = np.random.default_rng(4)
    for hi in (1 << 10, 1 << 30, 1 << 41):
        v = rng.integers(0, hi, 2048, dtype=np.uint64)
        v[rng.integers(0, 2048, 100)] = 0
        incl, tot = np.empty(2048, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
        _capi.check(_capi.lib().ssme_pf_test_block_scan(0, threads, _capi.u64ptr(v), _capi.u64ptr(incl), _capi.u64ptr(tot)))
        ref = np.cumsum(v, dtype=np.uint64)
        np.testing.assert_array_equal(incl, ref)
        assert tot[0] == ref[-1]


def test_device_rescale_bit_exact(sa, oracle):
    """A'_b = rint((double)A_b exp(m_b - m) 2^(rg-51)): the only floating-point step across tiles."""
    from ssme_amd import _capi
    rng = np.random.default_rng(6)
    A = rng.integers(0, 1 << 52, 20000, dtype=np.uint64)
    dm = np.concatenate([-rng.exponential(3.0, 19990), [0.0, -0.0, -800.0, -np.inf, np.nan, -1e-300, -745.0, -30.0, -1.0, -2.0]])
    for shift in (-9, -2, 0):
        out = np.empty(A.size, dtype=np.uint64)
        _capi.check(_capi.lib().ssme_pf_test_rescale(0, _capi.u64ptr(A), _capi.dptr(dm), shift, _capi.u64ptr(out), A.size))
        np.testing.assert_array_equal(out, oracle.rescale(A, dm, shift))


def test_device_gamma_bit_exact(sa, oracle):
    from ssme_amd import _capi
    for shape, n in ((2048.0, 4096), (1.0, 4096), (3.0, 2048), (1500.0, 1024)):
        out = np.empty(n)
        _capi.check(_capi.lib().ssme_pf_test_gamma(0, 20260101, 3, 17, shape, n, _capi.dptr(out)))
        assert_bits_equal(out, oracle.gamma_draws(20260101, 3, 17, shape, n), f"gamma({shape})")


# ---- filter parity vs oracle ------------------------------------------------------------------------
def _compare_state(bank, of, what, ancestors=True, logw=True):
    g = bank.state(0, ancestors=ancestors, logw=logw)
    o = of.state()
    assert_bits_equal(g["x"], o["x"], what + " particles")
    if logw:
        assert_bits_equal(g["logw"], o["logw"], what + " log-weights")
    np.testing.assert_array_equal(g["cdf"], o["cdf"], err_msg=what + " integer cdf")
    np.testing.assert_array_equal(g["A"], o["A"], err_msg=what + " tile sums")
    assert_bits_equal(g["mb"], o["mb"], what + " tile maxima")
    assert_bits_equal([g["m"]], [o["m"]], what + " max log-weight")
    assert g["S"] == o["S"] and g["rshift"] == o["rshift"], what + " integer weight sum"
    return g, o


@pytest.mark.parametrize("n", [1, 3, 64, 100, 500, 2047, 2048, 2049, 4096, 5000])
@pytest.mark.parametrize("resampler", [0, 1])
def test_step_parity_small(sa, oracle, spy, n, resampler):
    th = [1.0, 0.95, 0.25]
    seed = 20260101 + n
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed, resampler)
    bank.set_debug(True)
    bank.set_params(th)
    of = oracle.Filter(oracle.MODEL_SVOL, n, th, seed, resampler=resampler)
    for t in range(6):
        lg = bank.step(spy[t])[0]
        lo = of.step(spy[t])
        assert abs(lg - lo) <= TOL_LL and lg == lo, (t, lg, lo)
        g, o = _compare_state(bank, of, f"n={n} t={t}")
        if t > 0:
            np.testing.assert_array_equal(g["anc"], o["anc"], err_msg=f"ancestors n={n} t={t}")
            assert g["anc"].max() < n
    assert abs(bank.loglik()[0] - of.loglik) <= TOL_LL
    bank.close()


@pytest.mark.parametrize("resampler", [0, 1, 2, 3])
@pytest.mark.parametrize("model", [0, 1, 2])
def test_models_and_resamplers(sa, oracle, spy, model, resampler):
    th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}[model]
    n, seed = 3000, 77
    z = np.concatenate([[0.0], spy[:-1]])
    bank = sa.ParticleFilterBank(model, n, 1, seed, resampler)
    bank.set_debug(True)
    bank.set_params(th)
    of = oracle.Filter(model, n, th, seed, resampler=resampler)
    for t in range(5):
        zt = z[t] if model == 1 else None
        lg = bank.step(spy[t], zt)[0]
        lo = of.step(spy[t], 0.0 if zt is None else zt)
        assert lg == lo, (model, resampler, t, lg, lo)
    g, o = _compare_state(bank, of, f"model={model} rs={resampler}")
    np.testing.assert_array_equal(g["anc"], o["anc"])
    bank.close()


def test_series_matches_oracle_and_step_api(sa, oracle, spy):
    th = [1.0, 0.95, 0.25]
    n, seed, T = 5000, 5, 200
    of = oracle.Filter(oracle.MODEL_SVOL, n, th, seed)
    ll_o, per_o = of.run_series(spy[:T])
    for graph in (False, True):
        bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed)
        bank.set_graph_mode(graph)
        bank.set_params(th)
        ll = bank.run_series(spy[:T])[0]
        assert ll == ll_o and abs(ll - ll_o) <= TOL_LL
        assert_bits_equal(bank.per_step()[0], per_o, "per-step log-lik")
        # a second run on the same handle (graph replay) reproduces it
        assert bank.run_series(spy[:T])[0] == ll_o
        _compare_state(bank, of, "after series", ancestors=False, logw=False)
        bank.close()
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed)
    bank.set_params(th)
    s = 0.0
    for t in range(T):
        s += bank.step(spy[t])[0]
    assert bank.loglik()[0] == ll_o
    bank.close()


def test_block_size_tuning_is_result_invariant(sa, oracle, spy):
    """Exact integer cdf: the kernel's block shape cannot change any result bit."""
    th = [1.0, 0.95, 0.25]
    n, seed, T = 9000, 12, 25
    of = oracle.Filter(oracle.MODEL_SVOL, n, th, seed, tile=2048)
    ll_o, per_o = of.run_series(spy[:T])
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed, tile=2048)       # 2048-particle tiles run 256 / 512 / 1024 threads
    bank.set_debug(True)
    bank.set_params(th)
    for nt in (256, 512, 1024):
        bank.set_tuning(nt)
        assert bank.run_series(spy[:T])[0] == ll_o, nt
        g, o = _compare_state(bank, of, f"nt={nt}")
        np.testing.assert_array_equal(g["anc"], o["anc"])
    bank.close()


@pytest.mark.parametrize("tile", [512, 1024, 2048])
@pytest.mark.parametrize("n,rs,model", [(2049, 0, 0), (5000, 0, 0), (5000, 1, 1), (40000, 0, 0), (40000, 2, 2), (70000, 3, 0), (300000, 0, 1)])
def test_tile_sizes_against_oracle(sa, oracle, spy, tile, n, rs, model):
    """Both tile sizes, chosen explicitly (the default is by N): particles, integer cdf, ancestors, tile sums and per-step
    log conditional likelihoods bit-exact against the oracle configured with the same tile size; series == steps."""
    th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}[model]
    T = 6
    y = spy[:T]
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    bank = sa.ParticleFilterBank(model, n, 2, 9, rs, tile=tile)
    assert bank.tile == tile and bank.n_tiles == (n + tile - 1) // tile
    bank.set_debug(True, True)
    bank.set_params(th)
    of = oracle.Filter(model, n, th, 9, rep=1, resampler=rs, tile=tile)
    for t in range(T):
        lg = bank.step(y[t], None if z is None else z[t])[1]
        lo = of.step(y[t], 0.0 if z is None else z[t])
        assert_bits_equal([lg], [lo], f"tile {tile} logcondlike t={t}")
    sg, so = bank.state(1, ancestors=True), of.state()
    for key in ("x", "logw"):
        assert_bits_equal(sg[key], so[key], f"tile {tile}: {key}")
    np.testing.assert_array_equal(sg["cdf"], so["cdf"])
    np.testing.assert_array_equal(sg["anc"], so["anc"])
    np.testing.assert_array_equal(sg["A"], so["A"])
    per_steps = None
    ll = bank.run_series(y, z)
    assert_bits_equal(bank.per_step()[1], oracle.Filter(model, n, th, 9, rep=1, resampler=rs, tile=tile).run_series(y, z)[1], "series")
    bank.close()


def test_default_tile_rule_matches_the_oracle(sa, oracle):
    """The tile size is part of the arithmetic specification, so the rule that picks it from (N, n_filters) is too."""
    seen = set()
    for n, R in [(500, 1), (2048, 4), (2049, 1), (65536, 1), (131072, 1), (131073, 1), (262144, 1), (524288, 1), (524289, 1),
                 (1 << 20, 1), (65536, 2), (65536, 8), (65536, 9), (16384, 512), (16384, 8), (16384, 32), (3000, 43), (3000, 86), (3000, 200)]:
        bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, R, 1)
        assert bank.tile == oracle.default_tile(n, R), (n, R)
        seen.add(bank.tile)
        bank.close()
    assert seen == {512, 1024, 2048}


def test_a_filter_does_not_depend_on_how_its_bank_is_split(sa, oracle, spy):
    """the default tile follows (N, bank size), and the tile is part of the arithmetic -- so a bank that
    is split over GPUs / handles must declare its size (ssme_pf_config::n_filters_total, parallel.sharded_bank).  32 filters
    of 2^14 particles: alone the default tile is 1024; a handle holding 8 of them would pick 512 by itself.  Filter 13 gives
    the same bits in the whole bank, in every split that declares the bank, and in the oracle with the bank's tile."""
    from ssme_amd import parallel
    n, R, seed, T, th = 16384, 32, 5, 12, [1.0, 0.95, 0.25]
    y = spy[:T]
    tile = oracle.default_tile(n, R)
    assert tile == sa.default_tile(n, R) == 1024 and sa.default_tile(n, 8) == 512
    whole = sa.ParticleFilterBank(sa.MODEL_SVOL, n, R, seed)
    whole.set_params(th)
    ll = whole.run_series(y)
    assert whole.tile == tile
    whole.close()
    assert ll[13] == oracle.Filter(oracle.MODEL_SVOL, n, th, seed, rep=13, tile=tile).run_series(y)[0]
    for world in (2, 4, 8, 32):
        parts = []
        for rank in range(world):
            bank, first, cnt = parallel.sharded_bank(sa.MODEL_SVOL, n, R, world, rank, seed=seed)
            assert bank.tile == tile, (world, rank)
            bank.set_params(th)
            parts.append(bank.run_series(y))
            bank.close()
        assert_bits_equal(np.concatenate(parts), ll, f"bank split over {world} handles")
    alone = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed, first_filter_id=13)      # undeclared: another tile, other bits
    assert alone.tile == 512
    alone.close()


def _student_t_oracle(oracle, n, seed, rep, resampler, tile, th=(1.1, 0.95, 0.25, 7.0)):
    """tests/models/svol_student_t.h restated with the oracle's own libm-free functions (the operation order is the header's)."""
    import ctypes, math
    libm = ctypes.CDLL("libm.so.6")               # the header's std::lgamma is libm's (CPython's math.lgamma is its own code)
    libm.lgamma.restype = ctypes.c_double
    libm.lgamma.argtypes = [ctypes.c_double]
    beta, phi, sigma, nu = th
    e1 = lambda f, v: float(f(np.array([v]))[0])
    a2 = sigma / math.sqrt(1.0 - phi * phi)
    a3 = ((libm.lgamma(0.5 * (nu + 1.0)) - libm.lgamma(0.5 * nu)) - 0.5 * e1(oracle.log, nu * math.pi)) - e1(oracle.log, beta)
    a4 = 1.0 / (nu * (beta * beta))
    a5 = 0.5 * (nu + 1.0)
    prop = lambda x, zn, zcov: phi * x + zn * sigma
    logg = lambda y, x: (a3 - 0.5 * x) - a5 * e1(oracle.log, 1.0 + ((y * y) * a4) * e1(oracle.exp_t, -x))
    return oracle.UserModelFilter(n, seed, a2, prop, logg, rep=rep, resampler=resampler, tile=tile)


@pytest.mark.parametrize("n,rs,tile", [(1500, 0, 2048), (6000, 0, 512), (6000, 1, 2048)])
def test_user_model_extension_point_bit_exact_vs_oracle(sa, oracle, spy, tmp_path, n, rs, tile):
    """a FOURTH model (SVOL with Student-t observations, tests/models/svol_student_t.h) compiled in through the
    extension point of ssme_amd/csrc/model_api.h -- one header, no kernel edited -- runs as SSME_MODEL_USER0 through the C ABI
    (one-tile series kernel, tiled step kernel, both resamplers) and agrees with the oracle's callback-driven restatement to
    the bit: particles, log-weights, integer cdf, ancestors, per-step and series log-likelihood.  The stock library refuses
    the model id (SSME_ERR_UNSUPPORTED)."""
    import subprocess, sys
    from ssme_amd import build, _capi
    with pytest.raises(_capi.SsmeError) as ei:
        sa.ParticleFilterBank(sa.MODEL_USER0, 1000, 1, 1)
    assert ei.value.status == _capi.ERR_UNSUPPORTED and _capi.lib().ssme_pf_user_model_n_theta() == 0
    so = build.build_user_model(os.path.join(ROOT, "tests", "models", "svol_student_t.h"), "student_t")
    out = str(tmp_path / "um.npz")
    T, seed = 10, 33
    env = dict(os.environ, SSME_PF_LIB=so)
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "user_model_worker.py"), out, str(n), str(T), str(seed), str(rs), str(tile)],
                   env=env, check=True, timeout=600)
    r = np.load(out)
    of = _student_t_oracle(oracle, n, seed, 1, rs, tile)
    lls = [of.step(spy[t]) for t in range(T)]
    assert_bits_equal(r["lls"], lls, "user model: per-step log conditional likelihoods")
    so_ = of.state()
    assert_bits_equal(r["x"], so_["x"], "user model: particles")
    assert_bits_equal(r["logw"], so_["logw"], "user model: log-weights")
    np.testing.assert_array_equal(r["cdf"], so_["cdf"])
    np.testing.assert_array_equal(r["anc"], so_["anc"])
    for rep in range(2):
        ll, per = _student_t_oracle(oracle, n, seed, rep, rs, tile).run_series(spy[:T])
        assert r["series"][rep] == ll
        assert_bits_equal(r["per_step"][rep], per, "user model: series per-step")


def _two_factor_oracle(oracle, n, seed, rep, resampler, tile, sched, th=(1.1, 0.95, 0.9, 0.2, 0.15, -0.4)):
    """tests/models/svol_two_factor.h (dim_x = 2, dim_y = 2) restated with the oracle's own functions, in the header's operation order."""
    import math
    beta, phi1, phi2, s1, s2, rho = th
    e1 = lambda f, v: float(f(np.array([v]))[0])
    a2, a3, a4 = s1, s2 * rho, s2 * math.sqrt(1.0 - rho * rho)
    a5, a6 = e1(oracle.log, beta), 1.0 / (beta * beta)
    half_log_2pi = 0.91893853320467274178
    init = lambda zn: np.array([zn[0] * a2, zn[1] * a4])
    prop = lambda x, zn, zcov: np.array([phi1 * x[0] + zn[0] * a2, (phi2 * x[1] + zn[0] * a3) + zn[1] * a4])

    def logg(y, x):
        u1, u2 = x[0] + x[1], x[1]
        l1 = (-(a5 + 0.5 * u1) - half_log_2pi) - 0.5 * (((y[0] * y[0]) * a6) * e1(oracle.exp_t, -u1))
        l2 = (-(a5 + 0.5 * u2) - half_log_2pi) - 0.5 * (((y[1] * y[1]) * a6) * e1(oracle.exp_t, -u2))
        return l1 + l2
    return oracle.UserVectorModelFilter(n, seed, 2, 2, init, prop, logg, rep=rep, resampler=resampler, resamp_sched=sched, tile=tile)


@pytest.mark.parametrize("n,rs,tile,sched", [(1500, 0, 2048, 1), (6000, 0, 512, 1), (6000, 1, 2048, 1), (5000, 0, 1024, 2), (3 * 2048 + 77, 3, 2048, 1)])
def test_vector_user_model_bit_exact_vs_oracle(sa, oracle, spy, tmp_path, n, rs, tile, sched):
    """BSFilter<nparts, dimx, dimy, ...> with dimx = dimy = 2 through the extension point (tests/models/svol_two_factor.h: two
    volatility factors with correlated innovations, two observed series): particles as dim_x planes gathered at the ancestor's index,
    the second normal from one more Philox call per pair, vector observations -- step API and whole series, one tile and several,
    all resamplers, a resampling schedule -- against the oracle's callback-driven restatement, bit for bit."""
    import subprocess, sys
    from ssme_amd import build
    so = build.build_user_model(os.path.join(ROOT, "tests", "models", "svol_two_factor.h"), "two_factor")
    out = str(tmp_path / "uv.npz")
    T, seed = 8, 21
    env = dict(os.environ, SSME_PF_LIB=so)
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "user_vec_model_worker.py"), out, str(n), str(T), str(seed), str(rs), str(tile), str(sched)],
                   env=env, check=True, timeout=600)
    r = np.load(out)
    y = np.stack([spy[:T], spy[100:100 + T]], axis=1)
    of = _two_factor_oracle(oracle, n, seed, 1, rs, tile, sched)
    ll, per = of.run_series(y)
    so_ = of.state()
    assert_bits_equal(r["lls"], per, "vector model: per-step log conditional likelihoods (step API)")
    assert_bits_equal(r["x"], so_["x"], "vector model: particles, both components")
    assert_bits_equal(r["logw"], so_["logw"], "vector model: log-weights")
    np.testing.assert_array_equal(r["cdf"], so_["cdf"])
    np.testing.assert_array_equal(r["anc"], so_["anc"])
    # what a host-side functional of the whole state gets: every component, and weights proportional to exp(logw)
    assert_bits_equal(r["xw"], so_["x"], "vector model: download_weights particles")
    wo = np.exp(so_["logw"] - so_["logw"].max())
    np.testing.assert_allclose(r["w"] / r["w"].sum(), wo / wo.sum(), rtol=0, atol=1e-11)
    for rep in range(2):
        o2 = _two_factor_oracle(oracle, n, seed, rep, rs, tile, sched)
        ll2, per2 = o2.run_series(y)
        assert r["series"][rep] == ll2
        assert_bits_equal(r["per_step"][rep], per2, "vector model: series per-step")
        if rep == 0:
            assert_bits_equal(r["x_series"], o2.state()["x"], "vector model: particles after the series")


def _kalman_loglik_sum3(phi, sig, tau, y):
    """Exact log-likelihood of tests/models/lin_gauss_3d.h: three independent AR(1) components observed through their sum."""
    F = phi * np.eye(3)
    Q = np.diag(np.square(sig))
    H = np.ones((1, 3))
    m = np.zeros(3)
    P = Q / (1.0 - phi * phi)
    ll = 0.0
    for t, yt in enumerate(y):
        if t > 0:
            m = F @ m
            P = F @ P @ F.T + Q
        s = (H @ P @ H.T).item() + tau * tau
        e = yt - (H @ m).item()
        ll += -0.5 * (np.log(2.0 * np.pi * s) + e * e / s)
        K = (P @ H.T) / s
        m = m + K[:, 0] * e
        P = P - K @ H @ P
    return ll


@pytest.mark.parametrize("n,rs", [(5000, 0), (3 * 2048 + 77, 1)])
def test_vector_user_model_of_odd_shape_vs_oracle_and_kalman(sa, oracle, tmp_path, n, rs):
    """dim_x = 3, dim_y = 1 (tests/models/lin_gauss_3d.h): against the oracle's restatement bit for bit, and -- the model being linear
    and Gaussian -- against the EXACT log-likelihood of the Kalman filter within Monte-Carlo error over 32 replicate filters: an anchor
    that shares no code with the device or the oracle (it would catch, say, correlated normals across the components)."""
    import subprocess, sys
    from ssme_amd import build
    so = build.build_user_model(os.path.join(ROOT, "tests", "models", "lin_gauss_3d.h"), "lin_gauss_3d")
    T, seed, nseeds = 40, 5, 32
    phi, sig, tau = 0.9, (0.5, 0.3, 0.2), 0.7
    rng = np.random.default_rng(11)
    x = rng.normal(size=3) * np.array(sig) / np.sqrt(1.0 - phi * phi)
    y = np.empty(T)
    for t in range(T):
        if t > 0:
            x = phi * x + rng.normal(size=3) * np.array(sig)
        y[t] = x.sum() + tau * rng.normal()
    np.save(str(tmp_path / "y3.npy"), y)
    out = str(tmp_path / "uv3.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "user_vec3_model_worker.py"), out, str(n), str(T), str(seed), str(rs), str(nseeds)],
                   env=dict(os.environ, SSME_PF_LIB=so), check=True, timeout=600)
    r = np.load(out)
    e1 = lambda f, v: float(f(np.array([v]))[0])
    a4 = 1.0 / np.sqrt(1.0 - phi * phi)
    a5, a6 = e1(oracle.log, tau), 1.0 / tau
    init = lambda zn: np.array([zn[0] * (sig[0] * a4), zn[1] * (sig[1] * a4), zn[2] * (sig[2] * a4)])
    prop = lambda xx, zn, zcov: np.array([phi * xx[0] + zn[0] * sig[0], phi * xx[1] + zn[1] * sig[1], phi * xx[2] + zn[2] * sig[2]])

    def logg(yy, xx):
        d = (yy[0] - ((xx[0] + xx[1]) + xx[2])) * a6
        return (-a5 - 0.91893853320467274178) - 0.5 * (d * d)
    of = oracle.UserVectorModelFilter(n, seed, 3, 1, init, prop, logg, resampler=rs)
    ll, per = of.run_series(y)
    st = of.state()
    assert float(r["ll"][0]) == ll
    assert_bits_equal(r["per"][0], per, "3-d model: per-step")
    assert_bits_equal(r["x"], st["x"], "3-d model: particles, three components")
    np.testing.assert_array_equal(r["cdf"], st["cdf"])
    np.testing.assert_array_equal(r["anc"], st["anc"])
    exact = _kalman_loglik_sum3(phi, np.array(sig), tau, y)
    lls = r["lls"]
    se = lls.std(ddof=1) / np.sqrt(lls.size)
    assert abs(lls.mean() - exact) < 4.0 * se + 0.02, (lls.mean(), exact, se)       # (+ the O(1/N) bias of a log of an unbiased estimate)


def test_split_level2_with_1024_particle_tiles(sa, oracle, spy):
    """More than 2048 tiles of 1024 particles: the level-2 plan kernel path of the middle tile size."""
    n, th = 2100000, [1.0, 0.95, 0.25]
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 5, tile=1024)
    bank.set_params(th)
    assert bank.n_tiles > 2048
    assert_bits_equal(bank.run_series(spy[:4]), [oracle.Filter(oracle.MODEL_SVOL, n, th, 5, tile=1024).run_series(spy[:4])[0]], "loglik")
    bank.close()


def test_resample_schedule(sa, oracle, spy):
    th = [1.0, 0.95, 0.25]
    for rs in (2, 3):
        bank = sa.ParticleFilterBank(sa.MODEL_SVOL, 3000, 1, 9, sa.RESAMP_MULTINOMIAL, rs)
        bank.set_params(th)
        of = oracle.Filter(oracle.MODEL_SVOL, 3000, th, 9, resamp_sched=rs)
        ll_o, per_o = of.run_series(spy[:40])
        assert bank.run_series(spy[:40])[0] == ll_o
        assert_bits_equal(bank.per_step()[0], per_o, f"per-step rs={rs}")
        bank.close()


def test_replicates_and_filter_ids(sa, oracle, spy):
    """R filters in one handle == R oracle filters with replicate ids first_filter_id + r
    (thread_pool num_pfilters / swarm nparamparts); per-filter theta rows (pswarm)."""
    n, seed, T, R = 2500, 31, 30, 5
    z = np.concatenate([[0.0], spy[:-1]])
    rng = np.random.default_rng(2)
    thetas = np.stack([rng.uniform(.8, .99, R), rng.uniform(-.1, .1, R), rng.uniform(.01, .1, R),
                       rng.uniform(-.5, -.01, R)], axis=1)       # prior of test_pswarm.cpp:244
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL_LEVERAGE, n, R, seed, sa.RESAMP_MULTINOMIAL, 1, 0, first_filter_id=10)
    bank.set_params(thetas)
    ll = bank.run_series(spy[:T], z[:T])
    for r in range(R):
        of = oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, n, thetas[r], seed, rep=10 + r)
        assert ll[r] == of.run_series(spy[:T], z[:T])[0]
    # shared theta (PMMH replicates) + log-mean-exp aggregation (thread_pool.h:263-268)
    bank.set_params(thetas[0])
    ll = bank.run_series(spy[:T], z[:T])
    assert len(set(ll.tolist())) == R
    assert abs(bank.log_mean_exp() - oracle.log_mean_exp(ll)) < 1e-12
    bank.close()


def test_expectations(sa, oracle, spy):
    th = [0.9, 0.0, 1.0, -0.1]
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL_LEVERAGE, 1000, 2, 3)
    bank.set_params(th)
    ofs = [oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, 1000, th, 3, rep=r) for r in range(2)]
    for t in range(3):
        bank.step(spy[t], 0.0 if t == 0 else spy[t - 1])
        for of in ofs:
            of.step(spy[t], 0.0 if t == 0 else spy[t - 1])
    np.testing.assert_allclose(bank.expectations(3), 42.0, atol=1e-4)     # test_pswarm.cpp:252
    for kind in (0, 1, 2):
        np.testing.assert_allclose(bank.expectations(kind), [of.expectation(kind) for of in ofs], rtol=1e-12)
    bank.close()


def test_swarm_aggregate_as_the_reference_pool_computes_it(sa, spy):
    """pswarm_filter.h:96-160 + thread_pool.h:443-447: member i runs on thread i % num_threads, every thread averages its members,
    the thread averages are averaged -- the plain mean only when num_threads divides the member count (VERDICT r2 missing 6)."""
    R, T = 7, 3
    rng = np.random.default_rng(1)
    th = np.stack([rng.uniform(.8, .99, R), rng.uniform(-.1, .1, R), rng.uniform(.01, .1, R), rng.uniform(-.5, -.01, R)], axis=1)
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL_LEVERAGE, 700, R, 3)
    bank.set_params(th)
    for t in range(4):
        lcl = bank.step(spy[t], 0.0 if t == 0 else spy[t - 1])
    ex = bank.expectations_multi([0, 1])
    plain_ll, plain_ex = bank.swarm_aggregate([0, 1])
    assert abs(plain_ll - lcl.mean()) <= 1e-14 * abs(lcl.mean()) and np.allclose(plain_ex, ex.mean(axis=1), rtol=1e-13)
    ll3, ex3 = bank.swarm_aggregate([0, 1], num_threads=T)
    groups = [np.arange(R)[np.arange(R) % T == j] for j in range(T)]
    want_ll = np.mean([lcl[g].mean() for g in groups])
    want_ex = [np.mean([ex[f][g].mean() for g in groups]) for f in range(2)]
    assert abs(ll3 - want_ll) <= 1e-13 * abs(want_ll) and np.allclose(ex3, want_ex, rtol=1e-13)
    assert abs(ll3 - plain_ll) > 1e-9 * abs(plain_ll)                       # 7 members on 3 threads: not the plain mean
    ll7, _ = bank.swarm_aggregate([0], num_threads=7)
    assert abs(ll7 - plain_ll) <= 1e-14 * abs(plain_ll)
    bank.close()


def test_expectations_multi_weights_and_host_functionals(sa, oracle, spy):
    """All functionals of filter(y, z, fs) in one device pass; the (x, weights) download for arbitrary host-side h
    (pswarm_filter.h:44,87-89: h is a std::function) gives the same expectations without debug mode; several tiles."""
    th = [0.9, 0.0, 1.0, -0.1]
    n, R = 5000, 3
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL_LEVERAGE, n, R, 8)
    bank.set_params(th)
    ofs = [oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, n, th, 8, rep=r) for r in range(R)]
    for t in range(4):
        bank.step(spy[t], 0.0 if t == 0 else spy[t - 1])
        for of in ofs:
            of.step(spy[t], 0.0 if t == 0 else spy[t - 1])
    em = bank.expectations_multi([3, 0, 2, 1])
    assert em.shape == (4, R)
    np.testing.assert_allclose(em[0], 42.0, rtol=1e-12)
    for row, kind in ((1, 0), (2, 2), (3, 1)):
        np.testing.assert_allclose(em[row], [of.expectation(kind) for of in ofs], rtol=1e-12)
        assert_bits_equal(em[row], bank.expectations(kind), "multi == single")
    lcl, ex = bank.swarm_aggregate([0, 2])
    np.testing.assert_allclose(ex, [em[1].mean(), em[2].mean()], rtol=1e-13)
    for r in range(R):
        x, w = bank.weights(r)
        so = ofs[r].state()
        assert_bits_equal(x, so["x"], "weights(): particles")
        wo = np.exp(so["logw"] - so["logw"].max())
        np.testing.assert_allclose(w, wo, rtol=0, atol=2.0 ** -40)            # fixed point 2^-41 of the tile's largest weight
        np.testing.assert_allclose((x * x * w).sum() / w.sum(), em[3][r], rtol=1e-12)
    bank.close()
    # the reference-style model object with a callable h next to built-ins
    m = sa.svol_leverage(*th, nparts=3000, seed=5)
    m.filter(spy[0], 0.0, fs=[3, lambda xv: np.array([[xv, 2.0 * xv], [1.0, xv * xv]]), 1])
    e = m.getExpectations()
    assert abs(e[0] - 42.0) < 1e-9 and e[1].shape == (2, 2)
    assert abs(e[1][0, 1] - 2.0 * e[1][0, 0]) < 1e-12 and abs(e[1][1, 0] - 1.0) < 1e-12 and abs(e[1][1, 1] - e[2]) < 1e-12 * abs(e[2])


def test_reference_style_interface(sa, oracle, spy):
    """Reads like the reference's caller: mod.filter(y); logLike += mod.getLogCondLike() (estimate_univ_svol.h:121-127)."""
    mod = sa.svol_bs.from_pack([1.0, 0.5, 2.0e-4], nparts=500, seed=4)
    of = oracle.Filter(oracle.MODEL_SVOL, 500, [1.0, 0.5, float(np.sqrt(2.0e-4))], 4)
    logLike = 0.0
    for row in range(20):
        mod.filter(spy[row])
        logLike += mod.getLogCondLike()
        assert mod.getLogCondLike() == of.step(spy[row])
    assert logLike ** 2 > 0      # the reference's own assertion, test_pswarm.cpp:251
    lev = sa.svol_leverage(0.9, 0.0, 1.0, -0.1, nparts=10, seed=1)
    lev.filter(spy[0], 0.0, fs=[3])
    assert abs(lev.getExpectations()[0] - 42.0) < 1e-4 and lev.getLogCondLike() ** 2 > 0
    with pytest.raises(ValueError):
        sa.log_like_eval([1.0, .5, 2e-4], np.array([]))
    v = sa.log_like_eval([1.0, 0.95, 0.0625], spy[:50], nparts=500, num_pfilters=4, seed=8)
    assert np.isfinite(v)


def test_degenerate_inputs(sa, oracle):
    for th in ([-1.0, 0.5, 0.1], [1.0, 1.5, 0.1]):
        bank = sa.ParticleFilterBank(sa.MODEL_SVOL, 300, 1, 1)
        bank.set_params(th)
        assert np.isnan(bank.step(0.3)[0])
        assert np.isnan(bank.step(0.1)[0])          # the search must not hang or fault on NaN weights
        bank.close()
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, 300, 1, 1)
    bank.set_params([1.0, 0.5, 0.1])
    of = oracle.Filter(oracle.MODEL_SVOL, 300, [1.0, 0.5, 0.1], 1)
    for y in (0.0, 0.0, 13.56, -10.36, 0.0):        # exact zeros and the extremes of spy_returns.csv
        assert bank.step(y)[0] == of.step(y)
    bank.close()
    with pytest.raises(sa.SsmeError):
        sa.ParticleFilterBank(sa.MODEL_SVOL, 300, 1, 1).set_params([1.0, 0.5])      # wrong theta length


# ---- golden vectors -------------------------------------------------------------------------------------
@pytest.mark.parametrize("tname", ["start", "real"])
@pytest.mark.parametrize("n", [64, 500, 4096])
@pytest.mark.parametrize("rs", [("mn", 0), ("sys", 1)])
def test_gpu_reproduces_golden(sa, golden, spy, tname, n, rs):
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, int(golden["seed"][0]), rs[1])
    bank.set_debug(True)
    bank.set_params(golden[f"theta_{tname}"])
    lls = [bank.step(spy[t])[0] for t in range(8)]
    k = f"svol_{tname}_n{n}_{rs[0]}"
    assert_bits_equal(lls, golden[k + "_ll"], k + " ll")
    st = bank.state(0, ancestors=True)
    for name in ("x", "logw"):
        assert_bits_equal(st[name], golden[k + "_" + name], k + " " + name)
    np.testing.assert_array_equal(st["cdf"], golden[k + "_cdf"])
    np.testing.assert_array_equal(st["anc"], golden[k + "_anc"])
    bank.close()


def test_gpu_full_series_golden(sa, golden, spy):
    """Full spy_returns.csv series (T = 3084): N = 500 (shipped example size) and N = 2^16 (config 3)."""
    seed = int(golden["seed"][0])
    for tname in ("start", "real"):
        bank = sa.ParticleFilterBank(sa.MODEL_SVOL, 500, 1, seed)
        bank.set_params(golden[f"theta_{tname}"])
        assert bank.run_series(spy)[0] == golden[f"svol_{tname}_n500_full_ll"][0]
        assert_bits_equal(bank.per_step()[0], golden[f"svol_{tname}_n500_full_per"], "per-step")
        bank.close()
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, 1 << 16, 1, seed)
    bank.set_params(golden["theta_real"])
    ll = bank.run_series(spy)[0]
    assert abs(ll - golden["svol_real_n65536_full_ll"][0]) <= TOL_LL and ll == golden["svol_real_n65536_full_ll"][0]
    assert_bits_equal(bank.per_step()[0], golden["svol_real_n65536_full_per"], "per-step 2^16")
    bank.close()
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL_LEVERAGE, 4096, 1, seed, first_filter_id=3)
    bank.set_debug(True)
    bank.set_params([0.9, 0.0, 1.0, -0.1])
    z = np.concatenate([[0.0], spy[:-1]])
    lls = [bank.step(spy[t], z[t])[0] for t in range(8)]
    assert_bits_equal(lls, golden["lev_n4096_ll"], "leverage ll")
    st = bank.state(0, ancestors=True)
    for name in ("x", "logw"):
        assert_bits_equal(st[name], golden["lev_n4096_" + name], "leverage " + name)
    np.testing.assert_array_equal(st["cdf"], golden["lev_n4096_cdf"])
    np.testing.assert_array_equal(st["anc"], golden["lev_n4096_anc"])
    bank.close()


# ---- BASELINE sizes: size-independent properties -------------------------------------------------------
def test_full_size_properties(sa, oracle, spy):
    """N = 2^20 (config 2): first steps bit-exact vs oracle; Kalman anchor at 2^20; graph == eager."""
    n, seed = 1 << 20, 20260101
    th = [1.0, 0.95, 0.25]
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed)
    bank.set_debug(True)
    bank.set_params(th)
    of = oracle.Filter(oracle.MODEL_SVOL, n, th, seed)
    for t in range(3):
        assert bank.step(spy[t])[0] == of.step(spy[t])
    g, o = _compare_state(bank, of, "N=2^20")
    np.testing.assert_array_equal(g["anc"], o["anc"])
    a = bank.run_series(spy[:400])[0]
    bank.set_graph_mode(False)
    assert bank.run_series(spy[:400])[0] == a
    bank.close()
    # linear-Gaussian: at N = 2^20 the estimate is within 0.05 of the exact Kalman log-likelihood
    rng = np.random.default_rng(5)
    T, phi, sig, tau = 200, 0.9, 0.5, 0.7
    x = np.zeros(T)
    x[0] = rng.normal() * sig / np.sqrt(1 - phi ** 2)
    for t in range(1, T):
        x[t] = phi * x[t - 1] + sig * rng.normal()
    y = x + tau * rng.normal(size=T)
    exact, _ = oracle.kalman_loglik(phi, sig, tau, y)
    for rs in (0, 1):
        bank = sa.ParticleFilterBank(sa.MODEL_LIN_GAUSS, n, 1, 3, rs)
        bank.set_params([phi, sig, tau])
        assert abs(bank.run_series(y)[0] - exact) < 0.05
        bank.close()


def test_degenerate_weights_many_tile_span(sa, oracle):
    """One dominant particle region: an output tile's ancestors span many cdf tiles (general search path)."""
    n = 20000
    th = [0.5, 0.1, 0.01]          # linear Gaussian with tiny observation noise -> very uneven weights
    for rs in (0, 1, 2, 3):
        bank = sa.ParticleFilterBank(sa.MODEL_LIN_GAUSS, n, 1, 5, rs)
        bank.set_debug(True)
        bank.set_params(th)
        of = oracle.Filter(oracle.MODEL_LIN_GAUSS, n, th, 5, resampler=rs)
        for y in (0.3, 0.25, 0.31, -0.2):
            assert bank.step(y)[0] == of.step(y)
        g, o = _compare_state(bank, of, f"degenerate rs={rs}")
        np.testing.assert_array_equal(g["anc"], o["anc"])
        bank.close()


def test_max_tiles_per_filter(sa, spy):
    """Largest supported filter (2048 tiles = 2^22 particles): runs, finite, ancestors in range."""
    n = 1 << 22
    bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 1, sa.RESAMP_SYSTEMATIC)
    bank.set_debug(True)
    bank.set_params([1.0, 0.95, 0.25])
    ll = bank.run_series(spy[:4])[0]
    assert np.isfinite(ll)
    st = bank.state(0, ancestors=True)
    assert st["anc"].max() < n and (np.diff(st["anc"].astype(np.int64)) >= 0).all()   # systematic: sorted
    bank.close()


# ---- Liu-West filter --------------------------------------------------
def _lw_series(T, seed=3):
    rng = np.random.default_rng(seed)
    y = rng.normal(0.0, 0.02, T)
    z = np.concatenate([[0.0], y[:-1]])            # lagged return as the covariate (test_liu_west.cpp usage)
    return y, z


@pytest.mark.parametrize("n", [100, 2048, 5000, 40000])
def test_liu_west_bit_exact_vs_oracle(sa, oracle, n):
    """Every step: particles, transformed parameters, the k-draw and resampling indices, theta-bar, the Cholesky
    factor and the log conditional likelihood are bit-identical to the oracle's kernel-matched Liu-West filter."""
    T = 6
    y, z = _lw_series(T)
    g = sa.svol_lw_1_par(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=77)
    g.set_debug(True)
    o = oracle.LWFilter(n, 77, rep=0, delta=0.99)
    for t in range(T):
        g.filter(y[t], z[t])
        lo = o.step(y[t], z[t])
        sg, so = g.state(0, indices=True), o.state()
        assert_bits_equal(sg["x"], so["x"], f"LW x t={t}")
        assert_bits_equal(sg["theta"], so["theta"], f"LW theta t={t}")
        if t > 0:
            np.testing.assert_array_equal(sg["anc"], so["anc"], err_msg=f"LW ancestors t={t}")
            np.testing.assert_array_equal(sg["kidx"], so["kidx"], err_msg=f"LW k t={t}")
            assert_bits_equal(sg["thetabar"], so["thetabar"], f"LW thetabar t={t}")
            assert_bits_equal(np.tril(sg["L"]), np.tril(so["L"]), f"LW chol t={t}")
        assert_bits_equal([g.getLogCondLike()], [lo], f"LW logcondlike t={t}")
    # convenience reduction outside filter(): summation order differs (strided block sum vs sequential), fp64 tolerance
    np.testing.assert_allclose(g.param_means()[0], o.param_means(), rtol=1e-12, atol=0)
    g.close()


@pytest.mark.parametrize("transforms", [(1, 0, 2, 1), (3, 0, 3, 0), (2, 0, 3, 1)])
@pytest.mark.parametrize("form,n", [(0, 5000), (1, 700), (0, 600 * 2048 - 3)])
def test_liu_west_transform_sets_bit_exact_vs_oracle(sa, oracle, transforms, form, n):
    """param::pack's transform kinds per dimension (parameters.h:27: null 0, twice_fisher 1, logit 2, log 3).  The reference's test
    models use (logit, null, log, twice_fisher), which the stage kernels also hold as a compile-time constant; any other set takes
    the run-time switch.  Both against the oracle, both forms, in-kernel and split level-2 sizes."""
    T = 5 if n < 100000 else 3
    y, z = _lw_series(T, seed=3)
    lo, hi = (0.8, -0.1, 0.01, 0.01), (0.99, 0.1, 0.1, 0.5)          # inside the support of every kind used above
    cls = sa.svol_lw_2_par if form == 1 else sa.svol_lw_1_par
    g = cls(0.97, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3], nparts=n, seed=12, transforms=transforms)
    g.run_series(y, z)
    per = g.per_step()[0]
    st = g.state(0)
    g.close()
    o = oracle.LWFilter(n, 12, delta=0.97, transforms=transforms, lo=lo, hi=hi, form=form)
    po = np.array([o.step(y[t], z[t]) for t in range(T)])
    so = o.state()
    assert_bits_equal(per, po, f"LW transforms {transforms}: per-step")
    assert_bits_equal(st["x"], so["x"], f"LW transforms {transforms}: x")
    assert_bits_equal(st["theta"], so["theta"], f"LW transforms {transforms}: theta")


@pytest.mark.parametrize("form,rs", [(1, 1), (0, 3), (1, 3), (1, 2)])
@pytest.mark.parametrize("n", [700, 5000])
def test_liu_west_forms_and_schedules_bit_exact_vs_oracle(sa, oracle, form, rs, n):
    """The SISR form (LWFilter2WithCovs::filter, liu_west_filter.h:2191-2343, model svol_lw_2_par) and the resampling
    schedule m_rs (:1139-1140, :2317-2318) for both forms: particles, transformed parameters, indices and per-step log
    conditional likelihoods bit-identical to the oracle; series API == step API; expectations of built-in functionals."""
    T = 9
    y, z = _lw_series(T, seed=21)
    cls = sa.svol_lw_2_par if form == 1 else sa.svol_lw_1_par
    g = cls(0.98, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=13, rs=rs)
    g.set_debug(True)
    o = oracle.LWFilter(n, 13, rep=0, delta=0.98, form=form, resamp_sched=rs)
    steps = []
    for t in range(T):
        g.filter(y[t], z[t])
        lo = o.step(y[t], z[t])
        steps.append(g.getLogCondLike())
        assert_bits_equal([g.getLogCondLike()], [lo], f"LW form {form} rs {rs}: logcondlike t={t}")
        sg, so = g.state(0, indices=True), o.state()
        assert_bits_equal(sg["x"], so["x"], f"x t={t}")
        assert_bits_equal(sg["theta"], so["theta"], f"theta t={t}")
        if t > 0:
            np.testing.assert_array_equal(sg["anc"], so["anc"], err_msg=f"ancestors t={t}")
            np.testing.assert_array_equal(sg["kidx"], so["kidx"], err_msg=f"k t={t}")
    ex = g.expectations([3, 0, 1, 2, 4, 5, 6, 7])[:, 0]
    assert abs(ex[0] - 42.0) < 1e-9                                       # test/test_liu_west.cpp:199,401
    np.testing.assert_allclose(ex[1:], [o.expectation(i) for i in (0, 1, 2, 4, 5, 6, 7)], rtol=1e-11)
    np.testing.assert_allclose(ex[4:], g.param_means()[0], rtol=1e-15)
    xs, ths, ws = g.weights(0)                         # host-side h(x, z, theta) = sigma * x from the downloaded (x, theta, w)
    so = o.state()
    wo = np.exp(so["logw"] - so["logw"].max())
    sig_o = np.array([oracle.inv_transform(3, v) for v in so["theta"][2]])          # log transform of sigma (parameters.h:27)
    np.testing.assert_allclose((ths[2] * xs * ws).sum() / ws.sum(), (sig_o * so["x"] * wo).sum() / wo.sum(), rtol=1e-9)
    ll = g.run_series(y, z)
    assert_bits_equal(g.per_step()[0], np.array(steps), "LW series API == step API")
    assert abs(ll[0] - sum(steps)) < 1e-9
    g.close()


def test_liu_west_series_equals_steps_and_replicates(sa, oracle):
    n, T, R = 3000, 12, 3
    y, z = _lw_series(T, seed=5)
    g = sa.svol_lw_1_par(0.95, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, n_filters=R, seed=5)
    ll = g.run_series(y, z)
    per = g.per_step()
    np.testing.assert_allclose(per.sum(axis=1), ll, rtol=0, atol=1e-9)
    pm = g.param_means()
    assert len(set(ll.tolist())) == R                       # replicates are independent streams
    for r in range(R):
        o = oracle.LWFilter(n, 5, rep=r, delta=0.95)
        po = np.array([o.step(y[t], z[t]) for t in range(T)])
        assert_bits_equal(per[r], po, f"LW per-step filter {r}")
        np.testing.assert_allclose(pm[r], o.param_means(), rtol=1e-12, atol=0)
    g.reset()
    steps = []
    for t in range(T):
        g.filter(y[t], z[t])
        steps.append(g.getLogCondLike().copy())
    assert_bits_equal(np.array(steps).T, per, "LW step API == series API")
    g.close()


@pytest.mark.parametrize("cls,form", [("svol_lw_1_par", 0), ("svol_lw_2_par", 1)])
def test_liu_west_filter_with_functionals_as_the_reference_tests_call_it(sa, oracle, cls, form):
    """test/test_liu_west.cpp:176-200, :381-403: mod.filter(y, z, fs); mod.getExpectations()[0] == 42 -- with callables
    h(x, z, theta) on untransformed parameters (host sum over the downloaded weights) next to the device ids; and the
    no-covariate call filter(y, fs=...) (LWFilter::filter, liu_west_filter.h:238)."""
    n, T = 2500, 5
    y, z = _lw_series(T, seed=9)
    g = getattr(sa, cls)(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, 10, nparts=n, seed=4)
    o = oracle.LWFilter(n, 4, form=form)
    fs = [lambda x, zt, th: 42.0, 0, 6, lambda x, zt, th: np.array([[x, th[2] * x], [zt, np.sin(x) + th[0]]])]
    for t in range(T):
        g.filter(y[t], z[t], fs)
        assert g.getLogCondLike() == o.step(y[t], z[t])
    e = g.getExpectations()
    assert abs(e[0] - 42.0) < 1e-9
    assert abs(e[1] - o.expectation(0)) <= 1e-10 * abs(o.expectation(0))
    assert abs(e[2] - o.expectation(6)) <= 1e-10 * abs(o.expectation(6))
    st = o.state()
    w = np.exp(st["logw"] - st["logw"].max())
    phi, sig = 1.0 / (1.0 + np.exp(-st["theta"][0])), np.exp(st["theta"][2])       # transforms: logit, null, log, twice_fisher
    want = np.array([[(st["x"] * w).sum(), (sig * st["x"] * w).sum()], [z[T - 1] * w.sum(), ((np.sin(st["x"]) + phi) * w).sum()]]) / w.sum()
    np.testing.assert_allclose(e[3], want, rtol=1e-9)
    g.reset()
    o2 = oracle.LWFilter(n, 4, form=form)
    for t in range(3):
        g.filter(y[t], fs=[1])                              # no covariate
        assert g.getLogCondLike() == o2.step(y[t], 0.0)
    assert abs(g.getExpectations()[0] - o2.expectation(1)) <= 1e-10 * abs(o2.expectation(1))
    g.close()


def test_liu_west_rejects_bad_config(sa):
    from ssme_amd import SsmeError
    with pytest.raises(SsmeError):
        sa.svol_lw_1_par(1.5, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=100)      # delta outside (0,1]
    with pytest.raises(SsmeError):
        sa.svol_lw_1_par(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=0)


def test_liu_west_sharded_handle_rejects_unsharded_entry_points(sa):
    """A handle from ssme_lw_shard_create owns no particle buffers: the unsharded entry points must return
    SSME_ERR_STATE instead of launching kernels through null pointers."""
    import ctypes as C
    from ssme_amd import _capi as capi
    L = capi.lib()
    cfg = capi.LwConfig(n_particles=4096, n_filters=1, seed=1, device=0, first_filter_id=0, delta=0.99)
    cfg.transforms[:] = [2, 0, 3, 1]
    cfg.prior_lo[:] = [0.8, -0.1, 0.01, -0.5]
    cfg.prior_hi[:] = [0.99, 0.1, 0.1, -0.01]
    h = C.c_void_p()
    assert L.ssme_lw_shard_create(C.byref(cfg), 0, 1, C.byref(h)) == capi.OK
    y = np.array([0.1, 0.2]); out = np.zeros(4)
    assert L.ssme_lw_step(h, capi.dptr(y), capi.dptr(y), capi.dptr(out)) == capi.ERR_STATE
    assert L.ssme_lw_run_series(h, capi.dptr(y), capi.dptr(y), 2, capi.dptr(out)) == capi.ERR_STATE
    assert L.ssme_lw_get_param_means(h, capi.dptr(out)) == capi.ERR_STATE
    assert L.ssme_lw_download_state(h, 0, capi.dptr(np.zeros(4096)), None, None, None, None, None) == capi.ERR_STATE
    assert L.ssme_lw_set_debug(h, 1) == capi.ERR_STATE
    assert L.ssme_lw_destroy(h) == capi.OK


def test_set_seed_reuses_graph_and_matches_fresh_handle(sa, oracle, spy):
    """ssme_pf_set_seed: same handle (and captured graph), new stream == a fresh handle created with that seed."""
    y = spy[:40]
    a = sa.ParticleFilterBank(sa.MODEL_SVOL, 5000, 2, 11, sa.RESAMP_MULTINOMIAL)
    a.set_params([1.0, 0.95, 0.25])
    l11 = a.run_series(y)
    a.set_seed(12)
    l12 = a.run_series(y)
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, 5000, 2, 12, sa.RESAMP_MULTINOMIAL)
    b.set_params([1.0, 0.95, 0.25])
    assert_bits_equal(l12, b.run_series(y), "set_seed vs fresh handle")
    assert not np.array_equal(l11, l12)
    a.set_seed(11)
    assert_bits_equal(a.run_series(y), l11, "seed restored")
    o = oracle.Filter(oracle.MODEL_SVOL, 5000, [1.0, 0.95, 0.25], 12, rep=1)
    assert_bits_equal([l12[1]], [o.run_series(y)[0]], "set_seed vs oracle")
    a.close(); b.close()


# ---- one-tile filters: whole series in one launch (pf_small.h) ---------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 100, 256, 257, 500, 1000, 1024, 1025, 2048])
@pytest.mark.parametrize("rs", [0, 1, 2, 3])
def test_small_series_kernel_matches_tiled_kernel_and_oracle(sa, oracle, spy, n, rs):
    """N <= 2048: the single-launch series kernel, the tiled per-step kernel and the oracle agree bit for bit
    (log-likelihoods, per-step values, final particles, integer cdf, log-weights, last ancestors)."""
    y = spy[:25]
    th = [1.0, 0.95, 0.25]
    outs = []
    for small in (True, False):
        b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 3, 1234, rs)
        b.set_small_series(small)
        b.set_debug(True, True)
        b.set_params(th)
        ll = b.run_series(y)
        outs.append((ll, b.per_step(), [b.state(f, ancestors=True) for f in range(3)]))
        b.close()
    (l0, p0, s0), (l1, p1, s1) = outs
    assert_bits_equal(l0, l1, "series log-lik small vs tiled")
    assert_bits_equal(p0, p1, "per-step small vs tiled")
    for f in range(3):
        for key in ("x", "logw"):
            assert_bits_equal(s0[f][key][:n], s1[f][key][:n], f"{key} filter {f}")
        np.testing.assert_array_equal(s0[f]["cdf"], s1[f]["cdf"])
        np.testing.assert_array_equal(s0[f]["A"], s1[f]["A"])
        assert_bits_equal(s0[f]["mb"], s1[f]["mb"], "tile max")
        assert s0[f]["S"] == s1[f]["S"] and (s0[f]["m"] == s1[f]["m"] or (np.isnan(s0[f]["m"]) and np.isnan(s1[f]["m"])))
        np.testing.assert_array_equal(s0[f]["anc"][:n], s1[f]["anc"][:n])
    o = oracle.Filter(oracle.MODEL_SVOL, n, th, 1234, rep=2, resampler=rs)
    lo, po = o.run_series(y)
    assert_bits_equal([l0[2]], [lo], "small vs oracle")
    assert_bits_equal(p0[2], po, "per-step small vs oracle")


@pytest.mark.parametrize("model,th", [(1, [0.9, 0.0, 1.0, -0.1]), (2, [0.9, 0.5, 0.7])])
def test_small_series_kernel_other_models_and_schedules(sa, oracle, spy, model, th):
    y = spy[:30]
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    for sched in (1, 3):
        b = sa.ParticleFilterBank(model, 700, 2, 99, sa.RESAMP_MULTINOMIAL, sched)
        b.set_params(th)
        ll = b.run_series(y, z)
        o = oracle.Filter(model, 700, th, 99, rep=1, resampler=0, resamp_sched=sched)
        lo, po = o.run_series(y, z)
        assert_bits_equal([ll[1]], [lo], f"model {model} sched {sched}")
        assert_bits_equal(b.per_step()[1], po, "per-step")
        # the step API continues from the state the series kernel handed over
        l_next = b.step(y[5], None if z is None else z[5])
        lo_next = o.step(y[5], 0.0 if z is None else z[5])
        assert_bits_equal([l_next[1]], [lo_next], "step after small series")
        b.close()


def test_small_series_kernel_degenerate_weights(sa, oracle):
    """NaN observation => NaN log-likelihood from that step on, no hang; huge |y| => finite or -inf handled as the oracle does."""
    y = np.array([0.1, -0.2, np.nan, 0.3, 0.1])
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, 300, 1, 5)
    b.set_params([1.0, 0.95, 0.25])
    b.run_series(y)
    o = oracle.Filter(oracle.MODEL_SVOL, 300, [1.0, 0.95, 0.25], 5)
    _, po = o.run_series(y)
    assert_bits_equal(b.per_step()[0], po, "NaN propagation")
    y2 = np.array([0.1, 1e200, 0.2, -1e160, 0.0])
    b.run_series(y2)
    _, po2 = oracle.Filter(oracle.MODEL_SVOL, 300, [1.0, 0.95, 0.25], 5).run_series(y2)
    assert_bits_equal(b.per_step()[0], po2, "extreme observations")
    b.close()


# ---- particle swarm (pswarm_filter.h:325-560; test_pswarm.cpp:236-252) ------------------------------------------------
def test_swarm_with_covs_matches_oracle_members(sa, oracle, spy):
    from ssme_amd import _capi
    R, n, T = 12, 1500, 6
    sw = sa.svol_swarm_1([_capi.H_CONST42, _capi.H_X, _capi.H_VOL], .8, .99, -.1, .1, .01, .1, -.5, -.01, 10, nstateparts=n,
                         nparamparts=R, prior_seed=3, seed=21)
    y = spy[:T]
    z = np.concatenate([[0.0], y[:-1]])
    sw.update(y[0], z[0])                                  # the reference's test: one update, E[42] = 42, lcl^2 > 0
    assert sw.getLogCondLike() ** 2 > 0.0 and abs(sw.getExpectations()[0] - 42.0) < 1e-4
    th = sw.params
    assert th.shape == (R, 4) and np.all(th[:, 0] >= .8) and np.all(th[:, 0] <= .99) and np.all(th[:, 3] <= -.01)
    members = [oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, n, th[r], 21, rep=r) for r in range(R)]
    lls = np.array([m.step(y[0], z[0]) for m in members])
    assert_bits_equal(sw._member_lcl, lls, "member log-cond-likes, t=0")
    for t in range(1, T):
        sw.update(y[t], z[t])
        lls = np.array([m.step(y[t], z[t]) for m in members])
        assert_bits_equal(sw._member_lcl, lls, f"member log-cond-likes, t={t}")
        # the mean over members is reduced on the device (fixed tree): equal to the host mean to fp64 rounding
        assert abs(sw.getLogCondLike() - float(np.sum(lls) / R)) <= 1e-13 * abs(float(np.sum(lls) / R))
        ex = sw.getExpectations()
        assert abs(ex[0] - 42.0) < 1e-4
        for k, kind in ((1, 0), (2, 2)):
            want = sum(m.expectation(kind) for m in members) / R
            assert abs(ex[k] - want) <= 1e-9 * max(1.0, abs(want))
    sw.close()


def test_device_log_mean_exp_against_reference_thread_pool(sa, oracle, spy):
    """ssme_pf_log_mean_exp over R replicate filters == the reference's thread_pool<>::work on the same R values."""
    if oracle.build_ref() is None:
        pytest.skip("oracle/_ref not available")
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, 3000, 16, 8)
    b.set_params([1.0, 0.95, 0.25])
    ll = b.run_series(spy[:200])
    want = oracle.ref_log_mean_exp(ll)
    assert abs(b.log_mean_exp() - want) <= 1e-12 * abs(want)
    b.close()


@pytest.mark.parametrize("n", [300, 5000])
@pytest.mark.parametrize("delta", [0.99, 0.9])
def test_liu_west_device_reproduces_golden(sa, n, delta):
    """The HIP Liu-West filter against the committed golden vectors (no oracle involved at run time)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "liu_west_golden.npz"))
    f = sa.svol_lw_1_par(delta, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=int(g["seed"][0]), first_filter_id=1)
    f.set_debug(True)
    lls = []
    for t in range(g["y"].size):
        f.filter(g["y"][t], g["z"][t])
        lls.append(f.getLogCondLike())
    k = f"lw_n{n}_d{int(round(delta * 100))}"
    assert_bits_equal(lls, g[k + "_ll"], "LW golden log-cond-likes")
    st = f.state(0, indices=True)
    assert_bits_equal(st["x"], g[k + "_x"], "LW golden x")
    assert_bits_equal(st["theta"], g[k + "_theta"], "LW golden theta")
    np.testing.assert_array_equal(st["kidx"], g[k + "_kidx"])
    np.testing.assert_array_equal(st["anc"], g[k + "_anc"])
    f.close()


# ---- split level-2 (filters of more than 2048 tiles; forced here at moderate sizes) --------------------------------------
@pytest.mark.parametrize("model,th,rs,n", [(0, [1.0, 0.95, 0.25], 0, 40000), (0, [1.0, 0.95, 0.25], 1, 40000),
                                           (0, [1.0, 0.95, 0.25], 2, 9000), (0, [1.0, 0.95, 0.25], 3, 9000),
                                           (1, [0.9, 0.0, 1.0, -0.1], 0, 30000), (2, [0.9, 0.5, 0.7], 0, 5000)])
def test_split_level2_matches_in_kernel_level2_and_oracle(sa, oracle, spy, model, th, rs, n):
    y = spy[:16]
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    outs = []
    for split in (False, True):
        b = sa.ParticleFilterBank(model, n, 2, 31, rs)
        b.set_debug(True, True, split_level2=split)
        b.set_params(th)
        ll = b.run_series(y, z)
        st = b.state(1, ancestors=True)
        # the step API continues with the same policy
        nxt = b.step(y[3], None if z is None else z[3])
        outs.append((ll, b.per_step(), st, nxt))
        b.close()
    (l0, p0, s0, n0), (l1, p1, s1, n1) = outs
    assert_bits_equal(l0, l1, "split level-2: series log-lik")
    assert_bits_equal(p0, p1, "split level-2: per-step")
    assert_bits_equal(n0, n1, "split level-2: step API")
    assert_bits_equal(s0["x"], s1["x"], "x")
    np.testing.assert_array_equal(s0["cdf"], s1["cdf"])
    np.testing.assert_array_equal(s0["anc"], s1["anc"])
    assert s0["S"] == s1["S"]
    o = oracle.Filter(model, n, th, 31, rep=1, resampler=rs)
    lo, po = o.run_series(y, z)
    assert_bits_equal([l1[1]], [lo], "split level-2 vs oracle")
    assert_bits_equal(p1[1], po, "split level-2 per-step vs oracle")


def test_filter_of_more_than_2048_tiles(sa, oracle, spy):
    """N > 2^22 (2054 tiles): the split level-2 path against the oracle, bit for bit."""
    n = (1 << 22) + 5 * 2048 + 77
    y = spy[:4]
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 77)
    b.set_params([1.0, 0.95, 0.25])
    ll = b.run_series(y)
    o = oracle.Filter(oracle.MODEL_SVOL, n, [1.0, 0.95, 0.25], 77)
    lo, po = o.run_series(y)
    assert_bits_equal(ll, [lo], "N > 2^22 log-lik")
    assert_bits_equal(b.per_step()[0], po, "N > 2^22 per-step")
    b.close()


@pytest.mark.parametrize("rs", [0, 1])
def test_source_ranges_found_by_the_step_kernel(sa, oracle, spy, rs):
    """4100 tiles (five level-2 workgroups, one launch): the step kernel's own range search -- the 64-tile window around its
    tile id, and with an outlier's degenerate weights the 64-ary descent -- against the table kernels and the oracle; particles
    and cdf of the last step as well."""
    n, T = 4100 * 2048 - 11, 4
    y = spy[:T].copy()
    y[1] *= 40.0
    outs = []
    for pol in (None, "tables"):
        b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 5, rs)
        b.set_debug(False, True, split_level2=pol)
        b.set_params([1.0, 0.95, 0.25])
        ll = b.run_series(y)
        st = b.state(0)
        outs.append((ll, b.per_step(), st["x"], st["cdf"]))
        b.close()
    for k in range(3):
        assert_bits_equal(outs[0][k], outs[1][k], f"ranges in the step kernel vs tables [{k}]")
    np.testing.assert_array_equal(outs[0][3], outs[1][3])
    po = oracle.Filter(oracle.MODEL_SVOL, n, [1.0, 0.95, 0.25], 5, resampler=rs).run_series(y)[1]
    assert_bits_equal(outs[0][1][0], po, "ranges in the step kernel vs oracle")


def test_level2_policy_by_size_is_result_invariant(sa, spy):
    """1200 tiles: split level-2 (the default above 1024 tiles) == in-kernel level-2 (four tile sums per thread)."""
    n, y = 1200 * 2048 - 5, spy[:5]
    res = []
    for pol in (None, False, True):
        b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 3)
        b.set_debug(False, False, split_level2=pol)
        b.set_params([1.0, 0.95, 0.25])
        res.append((b.run_series(y), b.per_step()))
        b.close()
    for ll, per in res[1:]:
        assert_bits_equal(ll, res[0][0], "level-2 policy: log-lik")
        assert_bits_equal(per, res[0][1], "level-2 policy: per-step")


@pytest.mark.parametrize("rs,model", [(0, 1), (1, 0), (2, 0)])
def test_multi_workgroup_level2_matches_in_kernel_level2_and_oracle(sa, oracle, spy, rs, model):
    """More than 1024 tiles, two filters: the split level-2 runs as k_l2_scan_blocks + k_l2_ranges (two workgroups per filter
    here); all resamplers with a planned range, the leverage model, the step API; against the in-kernel level-2 and the oracle."""
    n, T = 1100 * 2048 + 3, 4
    th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1]}[model]
    y = spy[:T].copy()
    y[2] *= 30.0                                           # an outlier: unbalanced weights, wide source ranges
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    outs = []
    for split in (True, False, "tables"):
        b = sa.ParticleFilterBank(model, n, 2, 17, rs)
        b.set_debug(False, False, split_level2=split)
        b.set_params(th)
        ll = b.run_series(y, z)
        per = b.per_step()
        nxt = b.step(spy[T], None if z is None else y[T - 1])
        outs.append((ll, per, nxt))
        b.close()
    for k in (1, 2):
        assert_bits_equal(outs[0][0], outs[k][0], "multi-workgroup level-2: log-lik")
        assert_bits_equal(outs[0][1], outs[k][1], "multi-workgroup level-2: per-step")
        assert_bits_equal(outs[0][2], outs[k][2], "multi-workgroup level-2: step API")
    o = oracle.Filter(model, n, th, 17, rep=1, resampler=rs)
    assert_bits_equal(outs[0][1][1], o.run_series(y, z)[1], "multi-workgroup level-2 vs oracle")


@pytest.mark.parametrize("n", [30000, 700 * 2048 - 9])
def test_liu_west_split_level2_matches_in_kernel_level2(sa, oracle, n):
    """Liu-West with the level-2 of both draws in k_level2_plan == in-kernel level-2 (and the oracle at the small size)."""
    T = 5
    y, z = _lw_series(T, seed=8)
    outs = []
    for split in (False, True):
        g = sa.svol_lw_1_par(0.97, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, n_filters=2, seed=44)
        g.set_debug(True, split_level2=split)
        ll = g.run_series(y, z)
        st = g.state(1, indices=True)
        outs.append((ll, g.per_step(), st))
        g.close()
    (l0, p0, s0), (l1, p1, s1) = outs
    assert_bits_equal(l0, l1, "LW split level-2: log-lik")
    assert_bits_equal(p0, p1, "LW split level-2: per-step")
    for key in ("x", "theta", "thetabar"):
        assert_bits_equal(s0[key], s1[key], f"LW split level-2: {key}")
    np.testing.assert_array_equal(s0["kidx"], s1["kidx"])
    np.testing.assert_array_equal(s0["anc"], s1["anc"])
    if n <= 30000:
        o = oracle.LWFilter(n, 44, rep=1, delta=0.97)
        po = np.array([o.step(y[t], z[t]) for t in range(T)])
        assert_bits_equal(p1[1], po, "LW split level-2 vs oracle")


def test_liu_west_of_more_than_2048_tiles(sa, oracle):
    n = 2050 * 2048 - 3
    y, z = _lw_series(3, seed=2)
    g = sa.svol_lw_1_par(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=6)
    g.run_series(y, z)
    per = g.per_step()[0]
    o = oracle.LWFilter(n, 6, delta=0.99)
    po = np.array([o.step(y[t], z[t]) for t in range(3)])
    assert_bits_equal(per, po, "LW N > 2^22 vs oracle")
    g.close()


def test_split_level2_with_resampling_schedule(sa, oracle, spy):
    """resamp_sched = 3 with the split level-2: non-resampling steps carry log-weights, the plan still accounts every step."""
    y = spy[:14]
    n, th = 25000, [1.0, 0.95, 0.25]
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 2, 17, sa.RESAMP_MULTINOMIAL, 3)
    b.set_debug(False, False, split_level2=True)
    b.set_params(th)
    ll = b.run_series(y)
    o = oracle.Filter(oracle.MODEL_SVOL, n, th, 17, rep=1, resampler=0, resamp_sched=3)
    lo, po = o.run_series(y)
    assert_bits_equal([ll[1]], [lo], "split level-2, schedule 3: log-lik")
    assert_bits_equal(b.per_step()[1], po, "split level-2, schedule 3: per-step")
    b.close()


def test_randomized_configurations_against_oracle(sa, oracle, spy):
    """A deterministic sweep of 36 mixed configurations (sizes around tile boundaries, several filters, all resamplers and
    models, resampling schedules, both level-2 policies, both series paths): per-step log-likelihoods bit-exact."""
    rng = np.random.default_rng(20261004)
    sizes = [1, 3, 255, 256, 2047, 2048, 2049, 4095, 4097, 6000, 10240, 12289, 20481]
    thetas = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}
    for case in range(36):
        model = int(rng.integers(0, 3))
        n = int(sizes[rng.integers(0, len(sizes))])
        r = int(rng.integers(1, 5))
        rs = int(rng.integers(0, 4))
        sched = int(rng.choice([1, 1, 1, 2, 5]))
        T = int(rng.integers(2, 12))
        seed = int(rng.integers(1, 1 << 40))
        split = [None, True, False][int(rng.integers(0, 3))]
        small = bool(rng.integers(0, 2))
        y = spy[case:case + T]
        z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
        b = sa.ParticleFilterBank(model, n, r, seed, rs, sched)
        b.set_small_series(small)
        b.set_debug(False, False, split_level2=split)
        b.set_params(thetas[model])
        ll = b.run_series(y, z)
        per = b.per_step()
        rep = int(rng.integers(0, r))
        o = oracle.Filter(model, n, thetas[model], seed, rep=rep, resampler=rs, resamp_sched=sched)
        lo, po = o.run_series(y, z)
        what = f"case {case}: model {model} N {n} R {r} rs {rs} sched {sched} T {T} split {split} small {small}"
        assert_bits_equal(per[rep], po, what)
        assert_bits_equal([ll[rep]], [lo], what)
        b.close()


def test_randomized_liu_west_configurations_against_oracle(sa, oracle):
    rng = np.random.default_rng(4)
    sizes = [2, 100, 2047, 2048, 2049, 4100, 9000, 12289]
    for case in range(14):
        n = int(sizes[rng.integers(0, len(sizes))])
        r = int(rng.integers(1, 4))
        delta = float(rng.choice([0.9, 0.95, 0.99, 1.0]))
        T = int(rng.integers(2, 8))
        seed = int(rng.integers(1, 1 << 40))
        split = [None, True, False][int(rng.integers(0, 3))]
        y, z = _lw_series(T, seed=case)
        g = sa.svol_lw_1_par(delta, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, n_filters=r, seed=seed)
        g.set_debug(False, split_level2=split)
        g.run_series(y, z)
        per = g.per_step()
        rep = int(rng.integers(0, r))
        o = oracle.LWFilter(n, seed, rep=rep, delta=delta)
        po = np.array([o.step(y[t], z[t]) for t in range(T)])
        assert_bits_equal(per[rep], po, f"LW case {case}: N {n} R {r} delta {delta} T {T} split {split}")
        g.close()
