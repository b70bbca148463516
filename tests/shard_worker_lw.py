"""Worker of tests/test_sharded_gpu.py: one rank of a particle-sharded Liu-West filter (gloo rehearsal on cuda:0).

usage: shard_worker_lw.py RANK WORLD PORT OUT.npz N T SEED DELTA [FORM [RS]]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    n, T, seed, delta = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), float(sys.argv[8])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from ssme_amd.sharded import ShardedLiuWest
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:T]
    z = np.concatenate([[0.0], y[:-1]])
    form = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    rs = int(sys.argv[10]) if len(sys.argv) > 10 else 1
    f = ShardedLiuWest(delta, 0.8, 0.99, -0.1, 0.1, 0.01, 0.1, -0.5, -0.01, nparts=n, seed=seed, form=form, rs=rs)
    ll = f.run_series(y, z)
    np.savez(out, ll=ll, per_step=f.per_step(), x=f.local_particles(), theta=f.local_theta(), exchanged=f.exchanged_tiles)
    f.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
