"""Runs the library built with tests/models/svol_two_factor.h (a VECTOR user model: dim_x = 2, dim_y = 2) in its own process and writes
what the parity test compares with the oracle.      python tests/user_vec_model_worker.py OUT.npz N T SEED RESAMPLER TILE SCHED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd  # noqa: E402
from ssme_amd import _capi  # noqa: E402
import ctypes as C  # noqa: E402

out, n, T, seed, rs, tile, sched = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
dx, dy = C.c_int32(), C.c_int32()
assert _capi.lib().ssme_pf_user_model_n_theta() == 6
assert _capi.lib().ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)) == 0 and (dx.value, dy.value) == (2, 2)
spy = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))
y = np.stack([spy[:T], spy[100:100 + T]], axis=1)          # two observed series: [T, 2]
th = [1.1, 0.95, 0.9, 0.2, 0.15, -0.4]            # beta, phi1, phi2, sigma1, sigma2, rho
bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_USER0, n, 2, seed, rs, sched, tile=tile)
bank.set_debug(True, True)
bank.set_params(th)
lls = [bank.step(y[t])[1] for t in range(T)]
st = bank.state(1, ancestors=True)
xw, w = bank.weights(1)                          # all components + normalisable weights: a host-side E[h(x)] of the whole state
series = bank.run_series(y)
st2 = bank.state(0, ancestors=False)
np.savez(out, lls=np.array(lls), x=st["x"], logw=st["logw"], cdf=st["cdf"], anc=st["anc"], series=series, per_step=bank.per_step(), x_series=st2["x"], xw=xw, w=w)
bank.close()
