#!/usr/bin/env python3
"""Throughput of ONE particle-sharded filter (ssme_amd/sharded.py).  Launch one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \
        tools/shard_bench.py --particles 4194304 --T 128 [--backend nccl|gloo]

With G = 1 it measures the host-driven per-step overhead of the sharded path against the graph-replayed unsharded one.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--resampler", type=int, default=0)
    ap.add_argument("--lw", action="store_true", help="Liu-West filter (BASELINE.json configs[4]) instead of the bootstrap filter")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local % ndev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssme_amd.sharded import ShardedLiuWest, ShardedParticleFilter
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:a.T]
    z = np.concatenate([[0.0], y[:-1]]) if a.lw else None
    if a.lw:
        f = ShardedLiuWest(0.99, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=a.particles, seed=20260101)
    else:
        f = ShardedParticleFilter(0, a.particles, seed=20260101, resampler=a.resampler)
        f.set_params([1.0, 0.95, 0.25])
    best, ll = 1e30, None
    for _ in range(a.passes):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        ll = f.run_series(y, z)
        torch.cuda.synchronize()
        dist.barrier()
        best = min(best, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"sharded_filter": "liu-west" if a.lw else "bootstrap", "world": world, "backend": a.backend, "particles": a.particles, "T": a.T,
                          "loglik": ll, "us_per_step": best / a.T * 1e6, "particle_steps_per_s": a.particles * a.T / best,
                          "tiles_received_from_peers": f.exchanged_tiles}), flush=True)
    f.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
