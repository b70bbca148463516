#!/usr/bin/env python3
"""the CPU-baseline table, measured on the host cores of the box this runs on.
Times the reference-faithful restatement (oracle mode A: mt19937 + <random>, scalar, -O3 as example/CMakeLists.txt:11;
the reference itself cannot be built: pf, Eigen3, Catch2 are absent) and reports particle-steps/s and the log-likelihood
mean +- SE over seeds.  TEST / MEASUREMENT INFRASTRUCTURE (uses oracle/).   python3 tools/cpu_baseline_table.py > profiles/r02_cpu_baseline_table.md"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))
T = y.size
cores = bench.usable_cores()
START, REAL = bench.THETA_START, bench.THETA
SEEDS = 16


def timed(fn):
    t0 = time.perf_counter()
    r = fn()
    return r, time.perf_counter() - t0


def stats(v):
    v = np.asarray(v, dtype=np.float64)
    return f"{v.mean():.2f} ± {v.std(ddof=1) / np.sqrt(v.size):.2f}"


def row(name, inputs, ncores, rate, ll):
    print(f"| {name} | {inputs} | {ncores} | {rate:.3g} | {ll} |")


print(f"CPU: {bench._cpu_model()}, {cores} usable cores (affinity / cgroup quota); oracle mode A, -O3, T = {T} (spy_returns.csv)\n")
print("| Config | Inputs | Cores | particle-steps/s | log-lik (mean ± SE over 16 seeds) |")
print("|---|---|---|---|---|")
for n, tag in ((100, "C1a"), (500, "C1b")):
    for fl in (True, False):
        for th, tname in ((START, "chain start"), (REAL, "(1, .95, .25)")):
            (lls), dt = timed(lambda: [O.ref_run_series(O.MODEL_SVOL, th, n, y, seed=1 + s, use_float=fl, o3=True)[0] for s in range(SEEDS)])
            row(tag, f"N = {n}, {'fp32' if fl else 'fp64'}, θ = {tname}", 1, SEEDS * n * T / dt, stats(lls))
# C1c: N = 500, R = 100 replicates threaded over replicates (thread_pool with mc = true)
with ThreadPoolExecutor(cores) as ex:
    lls, dt = timed(lambda: list(ex.map(lambda s: O.ref_run_series(O.MODEL_SVOL, REAL, 500, y, seed=100 + s, use_float=True, o3=True)[0], range(100))))
row("C1c", "N = 500, fp32, R = 100 replicates, one per thread", cores, 100 * 500 * T / dt, stats(lls))
# C2-cpu: N = 2^20, first 24 steps (bounded sample)
K = 24
(ll1), dt1 = timed(lambda: O.ref_run_series(O.MODEL_SVOL, REAL, 1 << 20, y[:K], seed=1, o3=True)[0])
row("C2-cpu", f"N = 2^20, fp64, first {K} steps", 1, (1 << 20) * K / dt1, f"{ll1:.3f} (1 seed, {K} steps)")
with ThreadPoolExecutor(cores) as ex:
    lls, dt = timed(lambda: list(ex.map(lambda s: O.ref_run_series(O.MODEL_SVOL, REAL, 1 << 20, y[:8], seed=2 + s, o3=True)[0], range(cores))))
row("C2-cpu", "N = 2^20, fp64, first 8 steps, one replicate per thread", cores, cores * (1 << 20) * 8 / dt, stats(lls) + " (8 steps)")
# C3-cpu: N = 2^16, one likelihood evaluation
(ll3), dt3 = timed(lambda: O.ref_run_series(O.MODEL_SVOL, REAL, 1 << 16, y, seed=3, o3=True)[0])
row("C3-cpu", f"N = 2^16, fp64, one likelihood evaluation = {dt3:.1f} s (10 000 iterations ≈ {dt3 * 1e4 / 3600:.1f} h, extrapolated)", 1, (1 << 16) * T / dt3, f"{ll3:.2f} (1 seed)")
with ThreadPoolExecutor(cores) as ex:
    lls, dt = timed(lambda: list(ex.map(lambda s: O.ref_run_series(O.MODEL_SVOL, REAL, 1 << 16, y, seed=10 + s, o3=True)[0], range(cores))))
row("C3-cpu", f"N = 2^16, fp64, {cores} evaluations at once = {dt:.1f} s", cores, cores * (1 << 16) * T / dt, stats(lls))
