"""The statistical anchor of tests/test_stat_anchor_gpu.py at five times the seeds (not collected by pytest):
device filters against the reference-faithful restatement (oracle mode A), 1000 seeds a side; prints the z score
(difference of means over its standard error) of every compared quantity.
    python tests/anchor_extended.py [SEEDS]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ssme_amd as dev
from oracle import oracle
import stat_anchor as sa

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
spy = np.loadtxt(os.path.join(ROOT, "tests/golden/spy_returns.csv"))


def z_scores(a, b):
    a, b = np.atleast_2d(np.asarray(a, dtype=np.float64).T).T, np.atleast_2d(np.asarray(b, dtype=np.float64).T).T
    se = np.sqrt(a.var(0, ddof=1) / a.shape[0] + b.var(0, ddof=1) / b.shape[0])
    return (a.mean(0) - b.mean(0)) / se, a.mean(0), b.mean(0), se


def report(name, a, b):
    z, ma, mb, se = z_scores(a, b)
    print(f"{name}: z = {np.array2string(z, precision=2)}  mode A {np.array2string(ma, precision=5)}  device {np.array2string(mb, precision=5)}"
          f"  SE {np.array2string(se, formatter={'float_kind': lambda v: '%.2g' % v})}", flush=True)
    return np.abs(z).max()


worst = 0.0
th = [1.0, 0.95, 0.25]
y = spy[:300]
bank = dev.ParticleFilterBank(dev.MODEL_SVOL, 500, S, seed=20260101); bank.set_params(th); g = bank.run_series(y); bank.close()
worst = max(worst, report("svol_bs N=500 T=300 (one-tile kernel)", sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 500, y, seeds=S), g))
# BASELINE.json configs[0] as shipped: FLOATTYPE float (example/main.cpp:13), N = 100 and 500 -- SSME_F32 device handles against
# mode A run entirely in float (the pytest anchor of this configuration uses 400 seeds a side; VERDICT r2 asked for it here)
for n in (100, 500):
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, n, S, seed=314, dtype=dev._capi.F32); bank.set_params(th); g = bank.run_series(y); bank.close()
    a = np.array(sa.pmap(lambda s: oracle.ref_run_series(oracle.MODEL_SVOL, th, n, y, None, seed=1 + s, use_float=True)[0], range(S)))
    worst = max(worst, report(f"svol_bs FLOAT configuration N={n} T=300 (SSME_F32 device vs mode A in float)", a, g))
for tile in (512, 1024, 2048):
    bank = dev.ParticleFilterBank(dev.MODEL_SVOL, 5000, S, seed=7, tile=tile); bank.set_params(th); g = bank.run_series(spy[:100]); bank.close()
    worst = max(worst, report(f"svol_bs N=5000 T=100 tile {tile}", sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL, th, 5000, spy[:100], seeds=S), g))
thl = [0.95, 0.0, 0.2, -0.3]
yl, zl = sa.sim_leverage(300, *thl, seed=11)
bank = dev.ParticleFilterBank(dev.MODEL_SVOL_LEVERAGE, 500, S, seed=99); bank.set_params(thl); g = bank.run_series(yl, zl); bank.close()
worst = max(worst, report("svol_leverage N=500 T=300", sa.mode_a_bootstrap(oracle, oracle.MODEL_SVOL_LEVERAGE, thl, 500, yl, zl, seeds=S), g))
yw, zw = sa.sim_leverage(100, 0.95, 0.0, 0.05, -0.3, seed=9)
for form, rs in [(0, 1), (1, 1), (0, 3)]:
    cls = dev.svol_lw_2_par if form == 1 else dev.svol_lw_1_par
    gw = cls(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=2000, n_filters=S, seed=57, rs=rs)
    ll = gw.run_series(yw, zw); pm = gw.param_means(); gw.close()

    def one(s):
        l, _, m = oracle.lw_ref_run(2000, yw, zw, seed=1 + s, delta=0.99, form=form, resamp_sched=rs)
        return (l,) + tuple(m)
    a = np.array(sa.pmap(one, range(S)))
    worst = max(worst, report(f"Liu-West form {form} m_rs {rs} N=2000 T=100 (loglik, phi, mu, sigma, rho)", a, np.column_stack([ll, pm])))
print(f"extended anchor done: {S} seeds a side, largest |z| = {worst:.2f}", flush=True)
