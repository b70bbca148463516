import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ssme_amd as sa
y = np.loadtxt("tests/golden/spy_returns.csv")
for n in (1 << 22, 1 << 23, 1 << 25):
    for T in (16, 128, 1024):
        b = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, 20260101)
        b.set_debug(False, True)
        b.set_params([1.0, 0.95, 0.25])
        b.run_series(y[:T])
        st = b.state(0)
        A = st["A"].astype(np.float64) * np.exp(st["mb"] - st["m"])
        T_ = np.cumsum(A); S = T_[-1]; B = A.size
        tgt = S * np.arange(B) / B
        lo = np.searchsorted(T_, tgt, side="left")
        d = lo - np.arange(B)
        print(f"N=2^{int(np.log2(n))} T={T} B={B}: deviation lo-b min {d.min()} max {d.max()} | frac |d|>32: {(abs(d)>32).mean():.3f} |d|>256: {(abs(d)>256).mean():.3f}", flush=True)
        b.close()
