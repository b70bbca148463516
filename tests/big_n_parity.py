"""More than 2048 tiles over MANY steps against the oracle (the one-launch level-2's arrival counters are reused every step; the pytest
cases run 3-4 steps): series and step API, two resamplers.  Not collected by pytest (the oracle needs minutes):   python tests/big_n_parity.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd as sa
from oracle import oracle

spy = np.loadtxt(os.path.join(ROOT, "tests/golden/spy_returns.csv"))
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
bad = 0
for n, T, rs, via_step in [((1 << 22) + 3 * 2048 + 5, 48, 0, False), (2500 * 2048 - 1, 40, 1, True), ((1 << 23) + 77, 16, 0, False)]:
    y = spy[200:200 + T].copy()
    y[T // 2] *= 25.0                                    # an outlier half way: wide source ranges once
    t0 = time.time()
    b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 909, rs)
    b.set_params([1.0, 0.95, 0.25])
    if via_step:
        per = np.array([b.step(y[t])[0] for t in range(T)])
    else:
        b.run_series(y)
        per = b.per_step()[0]
    b.close()
    print(f"N {n} ({-(-n // 2048)} tiles) T {T} resampler {rs} {'step API' if via_step else 'series'}: device done, oracle running ...", flush=True)
    po = oracle.Filter(oracle.MODEL_SVOL, n, [1.0, 0.95, 0.25], 909, resampler=rs).run_series(y)[1]
    k = int((bits(per) != bits(po)).sum())
    bad += k
    print(f"  per-step values differing: {k} of {T}   sum {per.sum()!r}  [{time.time() - t0:.0f} s]", flush=True)
print("big-N parity done, mismatches:", bad, flush=True)
