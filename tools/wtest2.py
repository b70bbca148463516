import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssme_amd
y = np.loadtxt("tests/golden/spy_returns.csv")[:6]
bank = ssme_amd.ParticleFilterBank(0, 1 << 20, 1, 1, 1)
bank.set_params([1.0, 0.95, 0.25])
for t in range(6):
    bank.step(y[t])
bank.close()
