"""Shared helpers of the statistical anchor (last row): the only reference-facing evidence for
filter outputs, because the reference pins none (test/test_pswarm.cpp:251-252, test/test_liu_west.cpp:172,198-199
assert loglike^2 > 0 and E[42] = 42 on uninitialised inputs with clock-seeded RNGs).  Mode A of the oracle (mt19937 +
<random>, reference operation order) stands in for the reference; the bar is the one SURVEY states:
|difference of means| <= 3 SE over >= 200 seeds, with NO additive slack."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SEEDS = 200
N_SE = 3.0
WORKERS = min(16, os.cpu_count() or 1)          # ctypes releases the GIL: seeds run in parallel threads


def pmap(fn, items):
    with ThreadPoolExecutor(WORKERS) as ex:
        return list(ex.map(fn, items))


def sim_leverage(T, phi, mu, sig, rho, seed):
    """A series drawn from the SVOL-leverage model itself (test/test_pswarm.cpp:90-108), so that the filters are
    compared where the model fits."""
    rng = np.random.default_rng(seed)
    x, y = np.zeros(T), np.zeros(T)
    x[0] = mu + sig / np.sqrt(1 - phi * phi) * rng.normal()
    y[0] = np.exp(x[0] / 2) * rng.normal()
    for t in range(1, T):
        x[t] = mu + phi * (x[t - 1] - mu) + rho * sig * y[t - 1] * np.exp(-x[t - 1] / 2) + sig * np.sqrt(1 - rho * rho) * rng.normal()
        y[t] = np.exp(x[t] / 2) * rng.normal()
    return y, np.concatenate([[0.0], y[:-1]])


def assert_same_mean(a, b, what):
    """a, b: [seeds] or [seeds, k] samples of two estimators with (claimed) the same law."""
    a, b = np.atleast_2d(np.asarray(a, dtype=np.float64).T).T, np.atleast_2d(np.asarray(b, dtype=np.float64).T).T
    assert a.shape[0] >= SEEDS and b.shape[0] >= SEEDS, (a.shape, b.shape)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)), what
    se = np.sqrt(a.var(0, ddof=1) / a.shape[0] + b.var(0, ddof=1) / b.shape[0])
    d = a.mean(0) - b.mean(0)
    assert np.all(np.abs(d) <= N_SE * se), f"{what}: difference of means {d} exceeds {N_SE} SE = {N_SE * se} " \
                                           f"(means {a.mean(0)} vs {b.mean(0)})"
    return d / se


def mode_a_bootstrap(O, model, theta, n, y, z=None, seeds=SEEDS, seed0=1, fast_resampler=False):
    """log-likelihoods of `seeds` reference-faithful (mode A) bootstrap filters."""
    return np.array(pmap(lambda s: O.ref_run_series(model, theta, n, y, z, seed=seed0 + s, fast_resampler=fast_resampler)[0],
                         range(seeds)))


def mode_a_liu_west(O, n, y, z, delta=0.99, seeds=SEEDS, seed0=1):
    """[seeds, 5]: log-likelihood and the posterior means of (phi, mu, sigma, rho) of mode-A Liu-West filters."""
    def one(s):
        ll, _, pm = O.lw_ref_run(n, y, z, seed=seed0 + s, delta=delta)
        return (ll,) + tuple(pm)
    return np.array(pmap(one, range(seeds)))
