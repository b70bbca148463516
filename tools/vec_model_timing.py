"""us per step of the vector test model (tests/models/svol_two_factor.h, dim_x = dim_y = 2) -- run with SSME_PF_LIB=build/user/libssme_pf_two_factor.so"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssme_amd as sa
spy = np.loadtxt(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "spy_returns.csv"))
T = 512
y = np.stack([spy[:T], spy[100:100 + T]], axis=1)
for n in (1 << 16, 1 << 20, 1 << 22):
    b = sa.ParticleFilterBank(sa.MODEL_USER0, n, 1, 20260101)
    b.set_params([1.1, 0.95, 0.9, 0.2, 0.15, -0.4])
    best = 1e9
    for _ in range(3):
        ll = b.run_series(y)
        best = min(best, b.last_elapsed_ms())
    print(f"two-factor SV (dim_x = 2, dim_y = 2) N={n}: {best * 1e3 / T:.2f} us per step, {n * T / (best * 1e-3):.3g} particle-steps/s, loglik {ll[0]:.6f}", flush=True)
    b.close()
