"""Runs the library built with tests/models/lin_gauss_3d.h (dim_x = 3, dim_y = 1) in its own process.
    python tests/user_vec3_model_worker.py OUT.npz N T SEED RESAMPLER NSEEDS"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd  # noqa: E402
from ssme_amd import _capi  # noqa: E402
import ctypes as C  # noqa: E402

out, n, T, seed, rs, nseeds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
dx, dy = C.c_int32(), C.c_int32()
assert _capi.lib().ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)) == 0 and (dx.value, dy.value) == (3, 1)
y = np.load(os.path.join(os.path.dirname(out), "y3.npy"))[:T]
th = [0.9, 0.5, 0.3, 0.2, 0.7]                    # phi, sigma_1..3, tau
bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_USER0, n, 1, seed, rs)
bank.set_debug(True, True)
bank.set_params(th)
ll = bank.run_series(y)
st = bank.state(0, ancestors=True)
per = bank.per_step()
bank.close()
# the same series under other seeds, many replicate filters at once: the Monte-Carlo spread around the exact (Kalman) value
bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_USER0, n, nseeds, seed + 1, rs)
bank.set_params(th)
lls = bank.run_series(y)
bank.close()
np.savez(out, ll=ll, per=per, x=st["x"], logw=st["logw"], cdf=st["cdf"], anc=st["anc"], lls=lls)
