"""Whole-series parity at the shapes the short tests cannot afford: every log conditional likelihood of the full
spy_returns.csv series (T = 3084), device against oracle, bit for bit.  Not collected by pytest (minutes of CPU):
    python tests/long_parity.py            ->  one line per shape, then "long parity done, mismatches: K" """
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd as sa
from oracle import oracle

spy = np.loadtxt(os.path.join(ROOT, "tests/golden/spy_returns.csv"))
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
zlag = np.concatenate([[0.0], spy[:-1]])
# (model, N, filters, resampler, schedule, tile, T)
shapes = [(0, 100, 3, 0, 1, 0, 3084), (0, 500, 3, 0, 1, 0, 3084), (1, 500, 2, 1, 1, 0, 3084), (0, 2000, 2, 0, 1, 0, 3084),
          (0, 500, 2, 0, 3, 0, 3084), (0, 65536, 2, 0, 1, 0, 3084), (1, 16384, 4, 0, 1, 0, 3084), (0, 262144, 1, 0, 1, 0, 3084),
          (0, 65536, 1, 1, 1, 2048, 3084), (0, 300000, 1, 2, 1, 512, 1000)]
thetas = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1]}
bad = 0
for model, n, r, rs, sched, tile, T in shapes:
    y, z = spy[:T], (zlag[:T] if model == 1 else None)
    t0 = time.time()
    b = sa.ParticleFilterBank(model, n, r, 4242, rs, sched, tile=tile)
    b.set_params(thetas[model])
    ll = b.run_series(y, z)
    per = b.per_step()
    tl = b.tile
    b.close()
    def ref(rep):
        return oracle.Filter(model, n, thetas[model], 4242, rep=rep, resampler=rs, resamp_sched=sched, tile=tl).run_series(y, z)
    with ThreadPoolExecutor(r) as ex:
        refs = list(ex.map(ref, range(r)))
    ok = all(np.array_equal(bits(per[k]), bits(refs[k][1])) and bits([ll[k]])[0] == bits([refs[k][0]])[0] for k in range(r))
    bad += 0 if ok else 1
    print(f"model {model} N {n} filters {r} resampler {rs} schedule {sched} tile {tl} T {T}: log-lik {ll[0]!r} "
          f"{'== oracle (all ' + str(r * T) + ' per-step values, bit for bit)' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
# Liu-West, both forms
yy = spy[:600]; zz = zlag[:600]
for form, n, rs in [(0, 20000, 1), (1, 20000, 2), (0, 300000, 1)]:
    T = 600 if n < 100000 else 120
    t0 = time.time()
    g = (sa.svol_lw_2_par if form else sa.svol_lw_1_par)(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=77, rs=rs)
    g.run_series(yy[:T], zz[:T])
    per = g.per_step()[0]
    g.close()
    o = oracle.LWFilter(n, 77, form=form, resamp_sched=rs)
    po = np.array([o.step(yy[t], zz[t]) for t in range(T)])
    ok = np.array_equal(bits(per), bits(po))
    bad += 0 if ok else 1
    print(f"Liu-West form {form} N {n} m_rs {rs} T {T}: sum {per.sum()!r} {'== oracle (every step, bit for bit)' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
print("long parity done, mismatches:", bad, flush=True)
