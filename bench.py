#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the univariate-SVOL bootstrap filter (BASELINE.json metric).

Default workload (BASELINE.json configs[1]): svol_bs bootstrap filter, N = 2^20 particles, fp64, multinomial
resampling every step, the reference's own spy_returns.csv (T = 3084), one filter per GPU.
A "step" of the bench contract = one full pass of the filter over the series (log_like_eval,
example/estimate_univ_svol.h:108-131) = N*T particle-steps.

    --mode replicas (default)  every rank runs its own independent replicate filter (thread_pool's num_pfilters
                               semantics, thread_pool.h:189-215), no data-path collective; the log-likelihoods are gathered
                               once at the end and log-mean-exp'd (thread_pool.h:263-268)            => "scaling": "weak"
    --mode sharded             ONE N-particle filter with its particles sharded over the ranks (SURVEY.md section 8e row 2:
                               per-step all_gather of the tile sums + tile exchange)                 => "scaling": "strong"
    --lw                       with --mode sharded: the Liu-West filter of BASELINE.json configs[4] (N = 2^24 in total by
                               default, --lw-steps time steps of the series per pass)

--gpus N (N > 1) without a launcher: this process starts N ranks itself through torch.distributed.run BEFORE it touches
the GPU (a process that has initialised the GPU is never re-exec'ed) and relays rank 0's JSON line.  Under a launcher
(WORLD_SIZE set) --gpus must equal WORLD_SIZE, otherwise the run fails loudly.  On a box with fewer than N GPUs the run
fails unless --rehearse is given (gloo ranks sharing cuda:0; never a performance number, flagged in the line).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PARTICLES = 1 << 20
N_PARTICLES_LW = 1 << 24
SEED = 20260101
THETA = [1.0, 0.95, 0.25]           # (beta, phi, sigma): realistic point of SURVEY.md section 8d
THETA_START = [1.0, 0.5, 0.0141421356237]   # the example chain's start (estimate_univ_svol.h:153-155)
# Algorithmic bytes per particle-step of the fused step kernel: read cdf 8 + gather x 8 +
# write x' 8 + write cdf 8.  SURVEY.md section 8d budgets 48 B for a three-kernel split (it adds a 16 B log-weight
# round trip); the fused kernel keeps log-weights in registers, so the smaller figure is the honest one.
BYTES_PER_PSTEP = 32.0
BYTES_PER_PSTEP_LW = 176.0           # SURVEY.md section 8d, Liu-West with d_p = 4
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Host threads this process may really run at once: the affinity mask, cut by the cgroup CPU quota if there is one
    (a GPU box hands a 1-GPU job a share of its cores: oversubscribing 256 threads on a 16-CPU quota took 4 minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, (q + per - 1) // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, 64)                 # bounded sample: at most 64 replicate filters at once


def cpu_baseline(y, budget_steps):
    """Reference-faithful CPU restatement (oracle mode A: mt19937 + normal_distribution + discrete_distribution, scalar,
    -O3 as example/CMakeLists.txt:11) on a bounded sample of the workload.  The reference binary itself cannot be built
    (pf, Eigen3, Catch2 are absent), so kind = "port".  Headline leg: N = 2^20, ONE core = what the shipped main.cpp does
    (multicore = false, example/main.cpp:42).  Further legs: all cores by replicate-level threading
    (thread_pool with mc = true, thread_pool.h:133) and config 1 (N = 100 / 500, fp32 and fp64, whole series)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    cores = usable_cores()
    ys = y[:budget_steps]
    t0 = time.perf_counter()
    O.ref_run_series(O.MODEL_SVOL, THETA, N_PARTICLES, ys, seed=1, o3=True)
    dt1 = time.perf_counter() - t0
    # all cores: `cores` replicate filters at once, one per thread (ctypes releases the GIL)
    ysc = ys[:max(4, budget_steps // 3)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda s: O.ref_run_series(O.MODEL_SVOL, THETA, N_PARTICLES, ysc, seed=2 + s, o3=True), range(cores)))
    dtc = time.perf_counter() - t0
    legs = [{"leg": f"configs[1] N=2^20 fp64, all usable cores (one replicate filter per thread, first {ysc.size} steps)", "cores": cores,
             "value": cores * N_PARTICLES * ysc.size / dtc, "seconds": dtc}]
    for n in (100, 500):                               # config 1: the shipped example's sizes, whole series, chain start + realistic point
        for fl in (True, False):
            for th, tname in ((THETA_START, "chain start"), (THETA, "realistic")):
                t0 = time.perf_counter()
                ll = O.ref_run_series(O.MODEL_SVOL, th, n, y, seed=7, use_float=fl, o3=True)[0]
                dt = time.perf_counter() - t0
                legs.append({"leg": f"configs[0] N={n} {'fp32' if fl else 'fp64'} theta={tname}, T={y.size}, 1 core", "cores": 1,
                             "value": n * y.size / dt, "seconds": dt, "loglik": ll})
    return {"value": N_PARTICLES * budget_steps / dt1, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle mode A (mt19937/<random>, fp64, scalar, -O3), N=2^20, first {budget_steps} steps of "
                      f"spy_returns.csv, {dt1:.1f} s on 1 of {cores} usable cores ({_cpu_model()})",
            "legs": legs}


def oracle_delta(bank, y, steps, n):
    """|log-lik(GPU) - log-lik(oracle Philox mode)| on the first `steps` observations at full N."""
    from oracle import oracle as O
    of = O.Filter(O.MODEL_SVOL, n, THETA, SEED, rep=0)
    lo, _ = of.run_series(y[:steps])
    lg = bank.run_series(y[:steps])[0]
    return abs(lg - lo), lg, lo


def measure_traffic():
    """HBM bytes per launch of k_filter_step, measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
    separately, each with a calibration copy: gfx950 counts half the bytes of a wide streaming read) of a short series
    at the bench's N, in child processes started before this process touches the GPU.  None (+ reason) when rocprofv3
    is unavailable or this process is itself being profiled."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import traffic
    d = tempfile.mkdtemp(prefix="ssme_traffic_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
        traffic.collect(d, env=env, cwd=d, quiet=True)
        res = traffic.summarize(d, None, quiet=True)
        return res, None
    except Exception as e:                     # a failed profiler pass must not cost the bench line
        return None, f"PMC pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as a CHILD (torch.distributed.run), relay its output, exit with its code."""
    import socket
    import torch
    have = torch.cuda.device_count()                      # does not initialise the GPU on this image
    if have < args.gpus and not args.rehearse:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to report a "
                         f"{args.gpus}-GPU number (use --rehearse for a gloo rehearsal on one GPU)\n")
        sys.exit(2)
    if args.rehearse and args.gpus > 6:
        sys.stderr.write("bench.py: --rehearse keeps at most 6 ranks on one card\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas")
    ap.add_argument("--lw", action="store_true", help="Liu-West filter (BASELINE.json configs[4]); implies --mode sharded")
    ap.add_argument("--lw-steps", type=int, default=256, help="time steps of the series per Liu-West pass")
    ap.add_argument("--rehearse", action="store_true", help="gloo ranks sharing cuda:0 when the box has fewer GPUs than --gpus")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="sharded bootstrap filter: the C++ loop over RCCL (ssme_pf_shard_run_series) or the Python loop over torch.distributed")
    ap.add_argument("--particles", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--resampler", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-traffic", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-steps", type=int, default=24, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.lw:
        args.mode = "sharded"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                                  # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to run\n")
        sys.exit(2)

    # stdout carries ONE JSON line and nothing else: libraries that print banners on fd 1 (RCCL's version block) are sent
    # to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    n = args.particles or (N_PARTICLES_LW if args.lw else N_PARTICLES)
    # --- in-run HBM traffic of the dominant kernel (rank 0 at N = 1 only), before this process initialises the GPU ---
    traffic, traffic_note = None, "measured at --gpus 1, replicas mode, default N only"
    if world == 1 and args.mode == "replicas" and n == N_PARTICLES and not args.no_traffic:
        traffic, traffic_note = measure_traffic()

    import torch
    import torch.distributed as dist
    rehearse = bool(args.rehearse) and torch.cuda.device_count() < world
    dev_index = 0 if rehearse else local_rank
    if world > 1 or args.mode == "sharded":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    dev = torch.device("cuda", dev_index)
    n_ranks = dist.get_world_size() if dist.is_initialized() else 1          # the ranks the collective library actually sees
    assert n_ranks == world

    import ssme_amd
    from ssme_amd import parallel
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))
    if args.lw:
        y = y[:args.lw_steps]
    z = np.concatenate([[0.0], y[:-1]])
    T = y.size

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(device_ids=[dev_index]) if not rehearse else dist.barrier()

    dev_ms = 0.0
    if args.mode == "replicas":
        # one independent replicate filter per GPU; filter id = rank enters the Philox counter
        bank = ssme_amd.ParticleFilterBank(ssme_amd.MODEL_SVOL, n, 1, SEED, args.resampler, 1, dev_index, first_filter_id=rank)
        bank.set_params(THETA)
        run = lambda: bank.run_series(y)[0]   # includes the 24 KB H2D of y and the 8-byte D2H of the result
    else:
        from ssme_amd import sharded
        if args.lw:
            filt = sharded.ShardedLiuWest(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=SEED)
            native = args.driver == "native" and not rehearse       # RCCL needs one GPU per rank
            run = (lambda: filt.run_series_native(y, z)) if native else (lambda: filt.run_series(y, z))
        else:
            filt = sharded.ShardedParticleFilter(ssme_amd.MODEL_SVOL, n, SEED, args.resampler)
            filt.set_params(THETA)
            native = args.driver == "native" and not rehearse       # RCCL needs one GPU per rank
            run = (lambda: filt.run_series_native(y)) if native else (lambda: filt.run_series(y))

    for _ in range(args.warmup):
        run()
    sync()
    t0 = time.perf_counter()
    ll = None
    pass_lls = []
    for k in range(args.steps):
        if args.mode == "replicas":
            # seed 20260101 "and 4 more seeds for spread": pass k runs seed + k, the last pass the base
            # seed again (set_seed keeps the captured graph: the key lives in device memory), same workload every pass
            bank.set_seed(SEED + (k + 1) % args.steps)
        ll = run()
        pass_lls.append(float(ll))
        if args.mode == "replicas":
            dev_ms += bank.last_elapsed_ms()
    sync()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt)                              # MAX over ranks
    if args.mode == "replicas":
        lls = parallel.gather_logliks([ll], world)                # the one small collective of a pass
        lme = parallel.log_mean_exp(lls)                          # thread_pool.h:263-268
        psteps_per_pass = float(n) * T * world
    else:
        lme = ll                                                  # one filter: identical on every rank
        psteps_per_pass = float(n) * T

    short_ll = None
    if args.mode == "sharded" and not args.lw:
        short_ll = filt.run_series_native(y[:12]) if native else filt.run_series(y[:12])       # every rank takes part
    if rank == 0:
        value = psteps_per_pass * args.steps / dt
        if args.lw:
            wl = (f"Liu-West filter (svol_lw_1_par, d_p=4, delta=.99), N=2^{int(np.log2(n))} particles sharded over {world} GPU(s), fp64, "
                  f"first {T} steps of spy_returns.csv, resampling every step (BASELINE.json configs[4])")
        elif args.mode == "sharded":
            wl = (f"univ-SVOL bootstrap filter, ONE filter of N=2^{int(np.log2(n))} particles sharded over {world} GPU(s), fp64, T={T} "
                  f"(BASELINE.json configs[1] at {world} GPUs)")
        else:
            wl = ("univ-SVOL bootstrap filter, N=2^20 particles, fp64, T=3084 (spy_returns.csv), "
                  "multinomial resampling every step, 1 filter per GPU (BASELINE.json configs[1])")
        out = {
            "metric": "particle-steps/sec (NxT) univ-SVOL bootstrap filter; log-lik delta vs CPU ref",
            "value": value, "unit": "particle-steps/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.mode == "replicas" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "spy_returns.csv (the reference's own 3084-observation series)",
            "config": {"workload": wl, "mode": args.mode, "n_particles": n, "T": T,
                       "filters_per_gpu": 1 if args.mode == "replicas" else None, "resampler": int(args.resampler),
                       "theta": None if args.lw else THETA, "seed": SEED,
                       "parallelism": (f"replicates x{world} (no data-path collective)" if args.mode == "replicas" else
                                       f"particles sharded x{world} (per step: all_gather of tile sums/maxima + halo tile exchange)"),
                       "backend": "gloo rehearsal on one GPU (NOT a performance number)" if rehearse else ("rccl" if world > 1 or args.mode == "sharded" else "none")},
            "loglik_log_mean_exp": lme,
            "loglik_by_pass": pass_lls if args.mode == "replicas" else None,          # rank 0's filter, seeds SEED+1 .. , SEED (last)
        }
        if args.mode == "replicas":
            out["device_ms_per_step"] = dev_ms / args.steps
            # launch duration of the step kernel, live, HIP events on the handle's stream
            prof = bank.profile_series(y)
            k_us = prof["filter_step_us"]
            achieved = BYTES_PER_PSTEP * n / (k_us * 1e-6) / 1e9
            tbytes = None if traffic is None else traffic["filter_step_bytes_per_launch"]
            out["roofline"] = {"bound": "hbm", "kernel": "k_filter_step", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": tbytes,
                               "traffic_source": ("in-run rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE separately, read factor "
                                                  "calibrated on a streaming copy in the same pass)" if traffic is not None else traffic_note),
                               "traffic_detail": None if traffic is None else traffic["k_filter_step"],
                               "algorithmic_bytes_per_launch": BYTES_PER_PSTEP * n, "mean_launch_us": k_us,
                               "launches": prof["launches"],
                               "graph_replay_us_per_step": dev_ms * 1e3 / args.steps / T,
                               "whole_pass_GBps": BYTES_PER_PSTEP * value / world / 1e9}
            d, lg, lo = oracle_delta(bank, y, 12, n)
            out["loglik_delta_vs_oracle"] = {"abs_delta": d, "gpu": lg, "oracle": lo,
                                             "sample": "first 12 steps, full N, oracle Philox mode (bit-matched)"}
        else:
            # the step of a sharded filter includes its collectives: whole-step rate per rank, flagged as such
            bpp = BYTES_PER_PSTEP_LW if args.lw else BYTES_PER_PSTEP
            step_us = dt / args.steps / T * 1e6
            achieved = bpp * (n / world) / (step_us * 1e-6) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "sharded step (kernels + all_gather + tile exchange)", "achieved": achieved,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "includes_exchange": True, "algorithmic_bytes_per_rank_step": bpp * (n / world),
                               "mean_step_us": step_us}
            if not args.lw:
                if native:
                    _, _, path, exch = filt.native_state()
                    out["config"]["driver"] = "C++ over RCCL (ssme_pf_shard_run_series)"
                    out["roofline"]["path_last_pass"] = {1: "fixed halo, no host sync in the time loop", 2: "exact (host-planned) exchange"}.get(path, path)
                    out["roofline"]["tiles_received_last_pass_rank0"] = int(exch)
                else:
                    out["config"]["driver"] = "Python over torch.distributed"
                    out["roofline"]["tiles_received_last_pass_rank0"] = int(filt.exchanged_tiles)
                # the sharded filter must reproduce the unsharded oracle bit for bit: first 12 steps at full N
                from oracle import oracle as O
                lo = O.Filter(O.MODEL_SVOL, n, THETA, SEED, rep=0, resampler=args.resampler, tile=2048).run_series(y[:12])[0]
                out["loglik_delta_vs_oracle"] = {"abs_delta": abs(short_ll - lo), "gpu": short_ll, "oracle": lo,
                                                 "sample": "first 12 steps, full N, oracle Philox mode (bit-matched), unsharded oracle"}
            else:
                out["config"]["driver"] = ("C++ over RCCL (ssme_lw_shard_run_series): " + getattr(filt, "native_path", "?")) if native else "Python over torch.distributed"
                out["roofline"]["tiles_received_last_pass_rank0"] = int(filt.native_state()[2] if native else filt.exchanged_tiles)
        if not args.no_cpu_baseline and world == 1 and args.mode == "replicas":
            out["cpu_baseline"] = cpu_baseline(np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv")), args.cpu_steps)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    (bank if args.mode == "replicas" else filt).close()
    if dist.is_initialized():
        dist.barrier(device_ids=[dev_index]) if not rehearse else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
