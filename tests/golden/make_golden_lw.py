"""Generates tests/golden/liu_west_golden.npz from the CPU oracle's kernel-matched Liu-West filter (oracle.LWFilter).

Same status as make_golden.py: the reference's Liu-West filter cannot be built here (pf, Eigen3 absent) and its test
(test/test_liu_west.cpp:160-200) pins no output, so these vectors pin the HIP path and the oracle against regressions,
not against the reference ("parity unpinned").  Run from the repo root:  python tests/golden/make_golden_lw.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED = 20260101


def main():
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:10]
    z = np.concatenate([[0.0], y[:-1]])
    out = {"seed": np.array([SEED], dtype=np.uint64), "y": y, "z": z}
    for n in (300, 5000):
        for delta in (0.99, 0.9):
            f = O.LWFilter(n, SEED, rep=1, delta=delta)          # priors / transforms of test_liu_west.cpp:70,165
            lls = [f.step(y[t], z[t]) for t in range(y.size)]
            st = f.state()
            k = f"lw_n{n}_d{int(round(delta * 100))}"
            out[k + "_ll"] = np.array(lls)
            out[k + "_means"] = f.param_means()
            for name in ("x", "theta", "kidx", "anc", "thetabar", "L"):
                out[k + "_" + name] = st[name]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "liu_west_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
