#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: a short series at full N (default N=2^20, T=48, 2 passes)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssme_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--T", type=int, default=48)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--resampler", type=int, default=0)
ap.add_argument("--filters", type=int, default=1)
ap.add_argument("--model", type=int, default=0)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--nt", type=int, default=0)
ap.add_argument("--tile", type=int, default=0, help="particles per tile: 0 = by N, 512, 1024 or 2048")
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--split", type=int, default=-1, help="1: force the split level-2 (k_level2_plan), 0: force the in-kernel one, 2: split with the table kernels")
ap.add_argument("--lw", action="store_true", help="Liu-West filter instead of the bootstrap filter")
a = ap.parse_args()
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:a.T]
z = np.concatenate([[0.0], y[:-1]]) if a.model == 1 else None
th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}[a.model]
if a.lw:
    zz = np.concatenate([[0.0], y[:-1]])
    g = ssme_amd.svol_lw_1_par(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=a.n, n_filters=a.filters, seed=20260101)
    if a.split >= 0:
        g.set_debug(False, split_level2=bool(a.split))
    best = 1e9
    for _ in range(a.passes):
        ll = g.run_series(y, zz)
        best = min(best, g.last_elapsed_ms())
    print(f"liu-west N={a.n} R={a.filters} loglik {ll[:2]} best ms {best:.3f} us/step {best*1e3/a.T:.2f} "
          f"p-s/s {a.n * a.filters * a.T / (best * 1e-3):.4g} means {g.param_means()[0]}")
    sys.exit(0)
bank = ssme_amd.ParticleFilterBank(a.model, a.n, a.filters, 20260101, a.resampler, tile=a.tile)
if a.eager:
    bank.set_graph_mode(False)
if a.split >= 0:
    bank.set_debug(False, False, split_level2=("tables" if a.split == 2 else bool(a.split)))
bank.set_params(th)
combos = [256, 512, 1024] if a.sweep else [a.nt]
for nt in combos:
    if nt:
        bank.set_tuning(nt)
    best = 1e9
    for _ in range(a.passes):
        ll = bank.run_series(y, z)
        best = min(best, bank.last_elapsed_ms())
    prof = bank.profile_series(y, z) if a.sweep else {}
    print(f"nt={nt} loglik {ll[:2]} best ms {best:.3f} us/step {best*1e3/a.T:.2f} p-s/s {a.n * a.filters * a.T / (best * 1e-3):.4g}",
          {k: round(v, 2) for k, v in prof.items()})
bank.close()
