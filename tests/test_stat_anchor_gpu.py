"""GPU vs the reference-faithful restatement (oracle mode A: mt19937 + <random>, reference operation order) -- the only
reference-facing evidence for filter outputs, at the strength SURVEY.md section 8d specifies: >= 200 seeds,
|difference of mean log-likelihoods| <= 3 SE, no additive slack.  The 200 device filters are n_filters = 200 of ONE
handle (one launch per step); the 200 mode-A filters run on the host cores in parallel threads.

Filter outputs stay "parity unpinned" against the reference itself: its only fixtures for this path are
test/test_pswarm.cpp:251-252 and test/test_liu_west.cpp:172,198-199 (loglike^2 > 0, E[42] = 42, uninitialised inputs)."""
import numpy as np
import pytest

import stat_anchor as sa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import ssme_amd
    from ssme_amd import _capi
    assert _capi.lib() is not None
    return ssme_amd


def test_svol_bs_device_vs_mode_a(dev, oracle, spy):
    """svol_bs (example/univ_svol_bootstrap_filter.h:17-103), N = 500 as shipped (example/main.cpp:9), multinomial
    resampling every step (estimate_univ_svol.h:119), first 300 rows of spy_returns.csv."""
    th = [1.0, 0.95, 0.25]
    y = spy[:300]
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, 500, sa.SEEDS, seed=20260101)
    bank.set_params(th)
    g = bank.run_series(y)
    bank.close()
    a = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 500, y)
    sa.assert_same_mean(a, g, "svol_bs device vs mode A")


@pytest.mark.parametrize("n", [100, 500])
def test_svol_bs_float_configuration_device_vs_mode_a_in_float(dev, oracle, spy, n):
    """BASELINE.json configs[0] as shipped: FLOATTYPE float (example/main.cpp:13), N = 100 / 500.  The device handle is
    SSME_F32 (float at the boundary, fp64 arithmetic); mode A runs entirely in float, as the reference would."""
    th = [1.0, 0.95, 0.25]
    y = spy[:300]
    # 400 seeds a side is this test's declared count (round 2 raised it from 200 after a 3.3-SE excursion of the first 200 at
    # N = 500; a bias would grow with the seed count, and at 1000 seeds a side the same configuration sits at z = 0.65 (N = 100)
    # and z = -1.73 (N = 500): profiles/r03_anchor_1000_seeds.txt, tests/anchor_extended.py)
    seeds = 2 * sa.SEEDS
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, n, seeds, seed=314, dtype=dev._capi.F32)
    bank.set_params(th)
    g = bank.run_series(y)
    bank.close()
    assert np.array_equal(g, g.astype(np.float32).astype(np.float64))            # what comes back is float
    a = np.array(sa.pmap(lambda s: oracle.ref_run_series(oracle.MODEL_SVOL, th, n, y, None, seed=1 + s, use_float=True)[0],
                         range(seeds)))
    sa.assert_same_mean(a, g, f"svol_bs float configuration (N = {n}) device vs mode A in float")


def test_float_configuration_is_the_double_one_on_rounded_inputs(dev, spy):
    """SSME_F32 semantics: y, z, theta rounded to float on entry, results rounded to float on exit, nothing else."""
    f32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    th = [0.9, 0.0, 1.0, -0.1]
    y = spy[:40]
    z = np.concatenate([[0.0], y[:-1]])
    b32 = dev.ParticleFilterBank(dev.MODEL_SVOL_LEVERAGE, 3000, 3, seed=5, dtype=dev._capi.F32)
    b64 = dev.ParticleFilterBank(dev.MODEL_SVOL_LEVERAGE, 3000, 3, seed=5)
    b32.set_params(th)
    b64.set_params(f32(th))
    np.testing.assert_array_equal(b32.run_series(y, z), f32(b64.run_series(f32(y), f32(z))))
    np.testing.assert_array_equal(b32.per_step(), f32(b64.per_step()))
    np.testing.assert_array_equal(b32.expectations_multi([0, 1, 3]), f32(b64.expectations_multi([0, 1, 3])))
    b32.reset(); b64.reset()
    for t in range(5):
        np.testing.assert_array_equal(b32.step(y[t], z[t]), f32(b64.step(f32(y[t]), f32(z[t]))))
    b32.close(); b64.close()


@pytest.mark.parametrize("tile", [512, 2048])
def test_svol_bs_tiled_kernel_device_vs_mode_a(dev, oracle, spy, tile):
    """The same through the tiled step kernel with several tiles per filter (N = 5000: level-2 rescale + tile search),
    for both tile sizes (the per-tile weight scale and the per-tile Gamma draw must not move the mean)."""
    th = [1.0, 0.95, 0.25]
    y = spy[:100]
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, 5000, sa.SEEDS, seed=7, tile=tile)
    bank.set_params(th)
    g = bank.run_series(y)
    bank.close()
    a = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 5000, y)
    sa.assert_same_mean(a, g, "svol_bs (N = 5000, tiled kernel) device vs mode A")


def test_svol_leverage_device_vs_mode_a(dev, oracle):
    """svol_leverage (test/test_pswarm.cpp:80-134) with z_t = y_{t-1}, on a series drawn from the model."""
    th = [0.95, 0.0, 0.2, -0.3]
    y, z = sa.sim_leverage(300, *th, seed=11)
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL_LEVERAGE, 500, sa.SEEDS, seed=99)
    bank.set_params(th)
    g = bank.run_series(y, z)
    bank.close()
    a = sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL_LEVERAGE, th, 500, y, z)
    sa.assert_same_mean(a, g, "svol_leverage device vs mode A")


@pytest.mark.parametrize("delta", [0.99, 0.95])
def test_liu_west_device_vs_mode_a(dev, oracle, delta):
    """Liu-West (liu_west_filter.h:971-1159, model test/test_liu_west.cpp:22-157): log-likelihood and the posterior
    means of phi, mu, sigma, rho; 200 device filters in one handle against 200 mode-A seeds, 3 SE each, no slack."""
    y, z = sa.sim_leverage(100, 0.95, 0.0, 0.05, -0.3, seed=9)
    g = dev.svol_lw_1_par(delta, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=2000, n_filters=sa.SEEDS, seed=31)
    ll = g.run_series(y, z)
    pm = g.param_means()
    g.close()
    a = sa.mode_a_liu_west(oracle, 2000, y, z, delta=delta)
    sa.assert_same_mean(a, np.column_stack([ll, pm]), f"Liu-West delta={delta} device vs mode A (loglik, phi, mu, sigma, rho)")


@pytest.mark.parametrize("form,rs", [(1, 1), (0, 3)])
def test_liu_west_forms_device_vs_mode_a(dev, oracle, form, rs):
    """The SISR form (LWFilter2WithCovs, liu_west_filter.h:2191-2343) and a resampling schedule m_rs = 3 of the auxiliary
    form against the reference-faithful restatement: 200 seeds, 3 SE, log-likelihood and the four posterior means."""
    y, z = sa.sim_leverage(100, 0.95, 0.0, 0.05, -0.3, seed=9)
    cls = dev.svol_lw_2_par if form == 1 else dev.svol_lw_1_par
    g = cls(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=2000, n_filters=sa.SEEDS, seed=57, rs=rs)
    ll = g.run_series(y, z)
    pm = g.param_means()
    g.close()

    def one(s):
        l, _, m = oracle.lw_ref_run(2000, y, z, seed=1 + s, delta=0.99, form=form, resamp_sched=rs)
        return (l,) + tuple(m)
    a = np.array(sa.pmap(one, range(sa.SEEDS)))
    sa.assert_same_mean(a, np.column_stack([ll, pm]), f"Liu-West form {form} rs {rs} device vs mode A")
