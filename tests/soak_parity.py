import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd as sa
from oracle import oracle
spy = np.loadtxt(os.path.join(ROOT, "tests/golden/spy_returns.csv"))
def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 160        # bootstrap cases
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 40         # Liu-West cases
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 777)
sizes = [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 8193, 10000, 16385, 30000]
thetas = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}
bad = 0
big_sizes = [70000, 131073, 262144, 300001, 524288, 700000]
for case in range(NB):
    model = int(rng.integers(0, 3)); n = int(sizes[rng.integers(0, len(sizes))]); r = int(rng.integers(1, 6))
    if rng.random() < 0.08: n = int(big_sizes[rng.integers(0, len(big_sizes))]); r = int(rng.integers(1, 3))
    tile = int(rng.choice([0, 0, 512, 1024, 2048])) if n > 2048 else 0
    rs = int(rng.integers(0, 4)); sched = int(rng.choice([1, 1, 1, 2, 3, 7])); T = int(rng.integers(1, 14))
    seed = int(rng.integers(1, 1 << 50)); split = [None, True, False][int(rng.integers(0, 3))]; small = bool(rng.integers(0, 2))
    off = int(rng.integers(0, 3000)); y = spy[off:off + T].copy()
    if rng.random() < 0.1: y[int(rng.integers(0, T))] *= 1e3          # outlier
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    b = sa.ParticleFilterBank(model, n, r, seed, rs, sched, tile=tile)
    b.set_small_series(small); b.set_debug(False, False, split_level2=split); b.set_params(thetas[model])
    if rng.random() < 0.5:
        ll = b.run_series(y, z); per = b.per_step()
    else:
        per = np.array([b.step(y[t], None if z is None else z[t]) for t in range(T)]).T
    rep = int(rng.integers(0, r))
    o = oracle.Filter(model, n, thetas[model], seed, rep=rep, resampler=rs, resamp_sched=sched, tile=b.tile)
    _, po = o.run_series(y, z)
    ok = np.array_equal(bits(per[rep])[~np.isnan(po)], bits(po)[~np.isnan(po)]) and np.array_equal(np.isnan(per[rep]), np.isnan(po))
    if not ok:
        bad += 1; print("MISMATCH", case, model, n, r, rs, sched, T, seed, split, small, tile, per[rep], po, flush=True)
    b.close()
    if (case + 1) % 1000 == 0: print("bootstrap case", case + 1, "mismatches so far", bad, flush=True)     # progress (the GPU box kills silent runs)
print("bootstrap soak done, mismatches:", bad, flush=True)
bad = 0
for case in range(NL):
    n = int(sizes[rng.integers(2, len(sizes))]); r = int(rng.integers(1, 4)); delta = float(rng.choice([0.5, 0.9, 0.95, 0.99, 1.0]))
    if rng.random() < 0.08: n = int(big_sizes[rng.integers(0, 3)]); r = 1
    form = int(rng.integers(0, 2)); lrs = int(rng.choice([1, 1, 2, 3]))
    T = int(rng.integers(1, 9)); seed = int(rng.integers(1, 1 << 50)); split = [None, True, False][int(rng.integers(0, 3))]
    yy = np.random.default_rng(case).normal(0, 0.02, T); zz = np.concatenate([[0.0], yy[:-1]])
    g = (sa.svol_lw_2_par if form else sa.svol_lw_1_par)(delta, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, n_filters=r,
                                                         seed=seed, rs=lrs)
    g.set_debug(False, split_level2=split)
    if rng.random() < 0.5:
        g.run_series(yy, zz); per = g.per_step()
    else:
        per = []
        for t in range(T):
            g.filter(yy[t], zz[t]); per.append(np.atleast_1d(g.getLogCondLike()).copy())
        per = np.array(per).T
    rep = int(rng.integers(0, r))
    o = oracle.LWFilter(n, seed, rep=rep, delta=delta, form=form, resamp_sched=lrs)
    po = np.array([o.step(yy[t], zz[t]) for t in range(T)])
    if not np.array_equal(bits(per[rep]), bits(po)):
        bad += 1; print("LW MISMATCH", case, n, r, delta, T, seed, split, form, lrs, per[rep], po, flush=True)
    g.close()
    if (case + 1) % 500 == 0: print("liu-west case", case + 1, "mismatches so far", bad, flush=True)
print("liu-west soak done, mismatches:", bad, flush=True)
