"""Runs the user-model library (SSME_PF_LIB = a build of ssme_amd/csrc with -DSSME_USER_MODEL_HEADER=tests/models/svol_student_t.h)
in its own process -- a process binds ONE libssme_pf.so -- and writes what the parity test compares with the oracle.
    python tests/user_model_worker.py OUT.npz N T SEED RESAMPLER TILE"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd  # noqa: E402
from ssme_amd import _capi  # noqa: E402

out, n, T, seed, rs, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
assert _capi.lib().ssme_pf_user_model_n_theta() == 4
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:T]
th = [1.1, 0.95, 0.25, 7.0]                       # beta, phi, sigma, nu
bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_USER0, n, 2, seed, rs, tile=tile)
bank.set_debug(True, True)
bank.set_params(th)
lls = [bank.step(y[t])[1] for t in range(T)]
st = bank.state(1, ancestors=True)
series = bank.run_series(y)
np.savez(out, lls=np.array(lls), x=st["x"], logw=st["logw"], cdf=st["cdf"], anc=st["anc"], series=series, per_step=bank.per_step())
bank.close()
