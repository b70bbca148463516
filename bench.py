#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the univariate-SVOL bootstrap filter (BASELINE.json metric).

Workload (BASELINE.json configs[1]): svol_bs bootstrap filter, N = 2^20 particles, fp64, multinomial
resampling every step, the reference's own spy_returns.csv (T = 3084), one filter per GPU.
A "step" of the bench contract = one full pass of the filter over the series (log_like_eval,
example/estimate_univ_svol.h:108-131) = N*T particle-steps per GPU.  With --gpus N every rank runs
its own independent replicate filter (thread_pool's num_pfilters semantics, thread_pool.h:189-215)
with no data-path collective; the R log-likelihoods are gathered once at the end and log-mean-exp'd
(thread_pool.h:263-268) => weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PARTICLES = 1 << 20
SEED = 20260101
THETA = [1.0, 0.95, 0.25]           # (beta, phi, sigma): realistic point of SURVEY.md section 8d
# Algorithmic bytes per particle-step of the fused step kernel: read cdf 8 + gather x 8 +
# write x' 8 + write cdf 8.  SURVEY.md section 8d budgets 48 B for a three-kernel split (it adds a 16 B log-weight
# round trip); the fused kernel keeps log-weights in registers, so the smaller figure is the honest one.
BYTES_PER_PSTEP = 32.0
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(y, budget_steps):
    """Reference-faithful CPU restatement (oracle mode A: mt19937 + normal_distribution +
    discrete_distribution, scalar, 1 thread = shipped main.cpp multicore=false) on a bounded sample."""
    from oracle import oracle as O
    O.build()
    t0 = time.perf_counter()
    O.ref_run_series(O.MODEL_SVOL, THETA, N_PARTICLES, y[:budget_steps], seed=1)
    dt = time.perf_counter() - t0
    return {"value": N_PARTICLES * budget_steps / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle mode A (mt19937/<random>, fp64, scalar -O2), N=2^20, first {budget_steps} steps of "
                      f"spy_returns.csv, {dt:.1f} s on {os.cpu_count()} visible cores (1 used)"}


def oracle_delta(bank, y, steps):
    """|log-lik(GPU) - log-lik(oracle Philox mode)| on the first `steps` observations at full N."""
    from oracle import oracle as O
    of = O.Filter(O.MODEL_SVOL, N_PARTICLES, THETA, SEED, rep=0)
    lo, _ = of.run_series(y[:steps])
    lg = bank.run_series(y[:steps])[0]
    return abs(lg - lo), lg, lo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=N_PARTICLES, help=argparse.SUPPRESS)
    ap.add_argument("--resampler", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-steps", type=int, default=40, help=argparse.SUPPRESS)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    dev = torch.device("cuda", local_rank)

    import ssme_amd
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))
    T = y.size
    n = args.particles

    # one independent replicate filter per GPU; filter id = rank enters the Philox counter
    bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_SVOL, n, 1, SEED, args.resampler, 1, local_rank,
                                       first_filter_id=rank)
    bank.set_params(THETA)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        bank.run_series(y)
    sync()
    t0 = time.perf_counter()
    ll = None
    dev_ms = 0.0
    for _ in range(args.steps):
        ll = bank.run_series(y)[0]           # includes the 24 KB H2D of y and the 8-byte D2H of the result
        dev_ms += bank.last_elapsed_ms()
    sync()
    dt = time.perf_counter() - t0
    from ssme_amd import parallel
    dt = parallel.max_over_ranks(dt)                              # MAX over ranks
    lls = parallel.gather_logliks([ll], world)                    # the one small collective of a pass
    lme = parallel.log_mean_exp(lls)                              # thread_pool.h:263-268

    if rank == 0:
        psteps_per_pass = float(n) * T * world
        value = psteps_per_pass * args.steps / dt
        out = {
            "metric": "particle-steps/sec (NxT) univ-SVOL bootstrap filter; log-lik delta vs CPU ref",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "spy_returns.csv (the reference's own 3084-observation series)",
            "config": {"workload": "univ-SVOL bootstrap filter, N=2^20 particles, fp64, T=3084 (spy_returns.csv), "
                                   "multinomial resampling every step, 1 filter per GPU (BASELINE.json configs[1])",
                       "n_particles": n, "T": T, "filters_per_gpu": 1, "resampler": int(args.resampler),
                       "theta": THETA, "seed": SEED, "parallelism": f"replicates x{world} (no data-path collective)"},
            "loglik_log_mean_exp": lme,
            "device_ms_per_step": dev_ms / args.steps,
        }
        # launch duration of the step kernel, live, HIP events on the handle's stream
        prof = bank.profile_series(y)
        k_us = prof["filter_step_us"]
        achieved = BYTES_PER_PSTEP * n / (k_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("filter_step_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "k_filter_step", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "algorithmic_bytes_per_launch": BYTES_PER_PSTEP * n, "mean_launch_us": k_us,
                           "launches": prof["launches"],
                           "graph_replay_us_per_step": dev_ms * 1e3 / args.steps / T,
                           "whole_pass_GBps": BYTES_PER_PSTEP * value / world / 1e9}
        d, lg, lo = oracle_delta(bank, y, 12)
        out["loglik_delta_vs_oracle"] = {"abs_delta": d, "gpu": lg, "oracle": lo,
                                         "sample": "first 12 steps, N=2^20, oracle Philox mode (bit-matched)"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(y, args.cpu_steps)
        print(json.dumps(out), flush=True)
    bank.close()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
