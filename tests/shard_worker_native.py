"""Worker of tests/test_sharded_gpu.py: the C++ RCCL driver of the sharded filter (ssme_pf_shard_run_series) with one
rank per GPU.  On the one-GPU test box that is world = 1 (RCCL refuses two ranks on one device); the driver's bench
exercises the real multi-GPU exchange (bench.py --mode sharded --gpus N).

usage: shard_worker_native.py OUT.npz MODEL N T RESAMPLER SEED MODE      (MODEL = -1: the Liu-West filter, RESAMPLER = delta x 1000)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, model, n, T, rs, seed, mode = sys.argv[1], *(int(v) for v in sys.argv[2:8])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from ssme_amd.sharded import ShardedLiuWest, ShardedParticleFilter
    if model < 0:
        y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:T]
        z = np.concatenate([[0.0], y[:-1]])
        f = ShardedLiuWest(rs / 1000.0, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=seed)
        ll = f.run_series_native(y, z)
        per = f.per_step()
        x, th, exchanged = f.native_state()
        ll_py = f.run_series(y, z)
        np.savez(out, ll=ll, per_step=per, x=x, theta=th, path=f.native_path, exchanged=exchanged, ll_py=ll_py)
        f.close()
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()
        return
    th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}[model]
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:T]
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    f = ShardedParticleFilter(model, n, seed, rs)
    f.set_params(th)
    ll = f.run_series_native(y, z, mode=mode)
    per = f.per_step()
    x, cdf, path, exchanged = f.native_state()
    ll_py = f.run_series(y, z)                       # the Python-driven loop on the same handle: same bits
    np.savez(out + (f".rank{rank}" if world > 1 else ""), ll=ll, per_step=per, x=x, cdf=cdf, path=path, exchanged=exchanged, ll_py=ll_py)
    f.close()
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
