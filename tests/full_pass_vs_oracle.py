"""The WHOLE headline pass (BASELINE.json configs[1]: N = 2^20, T = 3084, 3.2e9 particle-steps) on the device against the oracle's
kernel-matched mode, every per-step log conditional likelihood bit for bit.  Not collected by pytest (the oracle needs ~8 minutes
of one CPU core):     python tests/full_pass_vs_oracle.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd as sa
from oracle import oracle

y = np.loadtxt(os.path.join(ROOT, "tests/golden/spy_returns.csv"))
n, th, seed = 1 << 20, [1.0, 0.95, 0.25], 20260101
bank = sa.ParticleFilterBank(sa.MODEL_SVOL, n, 1, seed)
bank.set_params(th)
t0 = time.time(); ll_dev = bank.run_series(y)[0]; t_dev = time.time() - t0
per_dev = bank.per_step()[0]
bank.close()
print(f"device: log-likelihood {ll_dev!r}  ({t_dev:.3f} s incl. graph capture)", flush=True)
import threading
stop = threading.Event()
def heartbeat():                              # the GPU box kills runs that stay silent for minutes
    while not stop.wait(60.0):
        print(f"  ... oracle running, {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
threading.Thread(target=heartbeat, daemon=True).start()
ll_o, per_o = oracle.Filter(oracle.MODEL_SVOL, n, th, seed).run_series(y)
stop.set()
print(f"oracle: log-likelihood {ll_o!r}  ({time.time() - t0:.0f} s on one core)", flush=True)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
print("series log-likelihood equal to the bit:", ll_dev == ll_o, "| per-step values differing:", int((bits(per_dev) != bits(per_o)).sum()), "of", y.size, flush=True)
