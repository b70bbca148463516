"""Worker of tests/test_sharded_gpu.py: one rank of a particle-sharded filter (gloo rehearsal: every rank uses cuda:0).

usage: shard_worker.py RANK WORLD PORT OUT.npz MODEL N T RESAMPLER SEED [RESAMP_SCHED]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out, model, n, T, rs, seed = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ssme_amd.sharded import ShardedParticleFilter
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:T]
    z = np.concatenate([[0.0], y[:-1]]) if model == 1 else None
    th = {0: [1.0, 0.95, 0.25], 1: [0.9, 0.0, 1.0, -0.1], 2: [0.9, 0.5, 0.7]}[model]
    sched = int(sys.argv[10]) if len(sys.argv) > 10 else 1
    f = ShardedParticleFilter(model, n, seed=seed, resampler=rs, resamp_sched=sched)
    f.set_params(th)
    f.record_ancestors(True)
    ll = f.run_series(y, z)
    np.savez(out, ll=ll, per_step=f.per_step(), x=f.local_particles(), cdf=f.local_cdf(),
             anc=f.anc.reshape(-1)[:f.n_local].cpu().numpy(), exchanged=f.exchanged_tiles)
    f.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
