"""Full per-GPU shapes of BASELINE.json configs[3] and configs[4] under test: the parity tests
elsewhere use sizes the oracle covers cheaply; here the SHAPES are the real ones and the oracle checks what it can reach
in seconds (a few filters / a few steps), plus size-independent properties (graph == eager, step API == series API)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import ssme_amd
    return ssme_amd


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_config4_per_gpu_slice_512_leverage_filters_of_2e14(dev, oracle, spy):
    """pswarm config: 4096 SVOL-leverage filters x 2^14 particles over 8 GPUs = 512 per GPU.
    theta_r from the test's prior (test/test_pswarm.cpp:244), z_t = y_{t-1}.  Per-step log conditional likelihoods of
    the first, the last and 3 randomly chosen filters == oracle; graph replay == eager launches == step API, all 512."""
    R, N, T = 512, 1 << 14, 4
    rng = np.random.default_rng(4096)
    lo, hi = np.array([0.8, -0.1, 0.01, -0.5]), np.array([0.99, 0.1, 0.1, -0.01])
    theta = lo + (hi - lo) * rng.random((R, 4))
    y = spy[:T]
    z = np.concatenate([[0.0], y[:-1]])
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL_LEVERAGE, N, R, seed=2026, first_filter_id=1024)   # e.g. rank 2 of 8
    assert bank.tile == 2048                     # 512 filters x 8 tiles fill the chip: the default keeps 2048-particle tiles
    bank.set_params(theta)
    ll_graph = bank.run_series(y, z)
    per_graph = bank.per_step()
    assert np.all(np.isfinite(ll_graph)) and len(set(ll_graph.tolist())) == R
    pick = sorted({0, R - 1, *rng.choice(R, 3, replace=False).tolist()})
    for r in pick:
        o = oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, N, theta[r], 2026, rep=1024 + r, tile=bank.tile)
        lo_, po = o.run_series(y, z)
        assert np.array_equal(_bits(per_graph[r]), _bits(po)), f"filter {r}: per-step differs from the oracle"
        assert ll_graph[r] == lo_
    bank.set_graph_mode(False)
    ll_eager = bank.run_series(y, z)
    assert np.array_equal(_bits(ll_eager), _bits(ll_graph))
    assert np.array_equal(_bits(bank.per_step()), _bits(per_graph))
    bank.reset()
    steps = np.array([bank.step(y[t], z[t]) for t in range(T)])            # [T, R], the unchanged-caller path
    assert np.array_equal(_bits(steps.T), _bits(per_graph))
    # swarm aggregation over the 512 members (pswarm_filter.h:103,136): expectation of the constant functional is 42
    e42 = bank.expectations(3)
    assert np.allclose(e42, 42.0, rtol=1e-12, atol=0)
    bank.close()


def test_config5_per_gpu_slice_liu_west_2e21(dev, oracle, spy):
    """Liu-West config: N = 2^24 over 8 GPUs = 2^21 per GPU.  The first steps of ONE 2^21-particle filter == oracle
    (per-step log conditional likelihoods bit-exact; posterior means of the parameters to fp64 rounding)."""
    N, T = 1 << 21, 3
    y = spy[:T]
    z = np.concatenate([[0.0], y[:-1]])
    g = dev.svol_lw_1_par(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=N, seed=2024)
    ll = g.run_series(y, z)[0]
    per = g.per_step()[0]
    pm = g.param_means()[0]
    g.close()
    o = oracle.LWFilter(N, 2024, delta=0.99)
    po = np.array([o.step(y[t], z[t]) for t in range(T)])
    assert np.array_equal(_bits(per), _bits(po))
    assert abs(ll - o.loglik) <= 1e-9
    np.testing.assert_allclose(pm, o.param_means(), rtol=1e-12, atol=0)


def test_config2_headline_shape_first_steps_vs_oracle(dev, oracle, spy):
    """The bench workload itself (N = 2^20, multinomial, one filter): the first steps == oracle, bit for bit, including
    the final particle array and the integer cdf."""
    N, T = 1 << 20, 6
    th = [1.0, 0.95, 0.25]
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, N, 1, seed=20260101)
    bank.set_params(th)
    ll = bank.run_series(spy[:T])[0]
    st = bank.state(0, logw=False)
    per = bank.per_step()[0]
    bank.close()
    o = oracle.Filter(oracle.MODEL_SVOL, N, th, 20260101)
    lo, po = o.run_series(spy[:T])
    so = o.state()
    assert ll == lo and np.array_equal(_bits(per), _bits(po))
    assert np.array_equal(_bits(st["x"]), _bits(so["x"]))
    assert np.array_equal(st["cdf"], so["cdf"])
