"""Generates tests/golden/bsfilter_golden.npz from the CPU oracle in kernel-matched (Philox) mode.

The reference cannot be built or imported here (pf, Eigen3, Catch2 absent) and none of its
tests pins a filter output, so these vectors are produced by this repo's own restatement
(oracle/ssme_oracle.cpp) -- "parity unpinned" against the reference, bit-exact pin for the HIP path.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED = 20260101
THETAS = {"start": [1.0, 0.5, float(np.sqrt(2.0e-4))],   # chain start, estimate_univ_svol.h:153-155
          "real": [1.0, 0.95, 0.25]}


def main():
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))
    out = {"seed": np.array([SEED], dtype=np.uint64)}
    for tname, th in THETAS.items():
        out[f"theta_{tname}"] = np.array(th)
        for n in (64, 500, 4096):
            for rs_name, rs in (("mn", O.RESAMP_MULTINOMIAL), ("sys", O.RESAMP_SYSTEMATIC)):
                f = O.Filter(O.MODEL_SVOL, n, th, SEED, resampler=rs)
                lls = [f.step(y[t]) for t in range(8)]
                st = f.state()
                k = f"svol_{tname}_n{n}_{rs_name}"
                out[k + "_ll"] = np.array(lls)
                for name in ("x", "logw", "cdf", "anc"):
                    out[k + "_" + name] = st[name]
        f = O.Filter(O.MODEL_SVOL, 500, th, SEED)
        ll, per = f.run_series(y)
        out[f"svol_{tname}_n500_full_ll"] = np.array([ll])
        out[f"svol_{tname}_n500_full_per"] = per
    # N = 2^16, full series, theta "real" (config 3's filter)
    f = O.Filter(O.MODEL_SVOL, 1 << 16, THETAS["real"], SEED)
    ll, per = f.run_series(y)
    out["svol_real_n65536_full_ll"] = np.array([ll])
    out["svol_real_n65536_full_per"] = per
    # leverage model (pswarm filter), z_t = y_{t-1}
    z = np.concatenate([[0.0], y[:-1]])
    thl = [0.9, 0.0, 1.0, -0.1]          # test/test_svol_leverage_samples.csv row
    f = O.Filter(O.MODEL_SVOL_LEVERAGE, 4096, thl, SEED, rep=3)
    lls = [f.step(y[t], z[t]) for t in range(8)]
    st = f.state()
    out["lev_n4096_ll"] = np.array(lls)
    for name in ("x", "logw", "cdf", "anc"):
        out["lev_n4096_" + name] = st[name]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bsfilter_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
