"""Multi-GPU host logic: independent filters shard across ranks, no data-path collective.

The reference's only parallelism is over independent filters: thread_pool runs num_pfilters replicate
filters per likelihood evaluation and log-mean-exps them (include/ssme/thread_pool.h:189-215,263-268);
split_data_thread_pool deals the swarm's filters round-robin to threads (thread_pool.h:443-447).  Here
the same units (whole filters) are dealt to GPUs: one process per GPU, contiguous blocks of filter ids
(the id enters the Philox counter, so a filter's random stream does not depend on the sharding), and ONE
small all_gather of the per-filter log-likelihoods at the end of a pass.
"""
import numpy as np


def shard_filters(n_filters, world, rank):
    """Contiguous, balanced shard of filter ids: returns (first_filter_id, count) for `rank`."""
    if n_filters < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("bad shard request")
    base, rem = divmod(n_filters, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def sharded_bank(model, n_particles, n_filters, world, rank, seed=0, device=0, **kw):
    """This rank's part of a bank of n_filters independent filters: filter ids and Philox streams are the global ones and
    the tile size (part of the arithmetic specification) is chosen from the WHOLE bank (ssme_pf_config::n_filters_total), so
    filter r gives the same bits on 1, 2, 4 or 8 GPUs.  Returns (bank or None when this rank holds no filter, first id, count)."""
    from .filters import ParticleFilterBank
    first, count = shard_filters(n_filters, world, rank)
    if count == 0:
        return None, first, 0
    return ParticleFilterBank(model, n_particles, count, seed=seed, device=device, first_filter_id=first,
                              n_filters_total=n_filters, **kw), first, count


def log_mean_exp(v):
    """thread_pool.h:263-268: m + log(sum exp(v - m)) - log(n)."""
    v = np.asarray(v, dtype=np.float64)
    m = v.max()
    if not np.isfinite(m):
        return float(m)        # NaN stays NaN, all -inf stays -inf (reference: a value, not an error)
    return float(m + np.log(np.exp(v - m).sum()) - np.log(v.size))


def gather_logliks(local_lls, n_filters, group=None):
    """All ranks receive the log-likelihoods of all n_filters filters, in filter-id order."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local_lls, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    cap = max(shard_filters(n_filters, world, r)[1] for r in range(world))     # pad to equal length
    buf = torch.full((cap,), float("nan"), dtype=torch.float64, device=dev)
    buf[:local.size] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = [out[r][:shard_filters(n_filters, world, r)[1]].cpu().numpy() for r in range(world)]
    assert shard_filters(n_filters, world, rank)[1] == local.size
    return np.concatenate(parts)


def max_over_ranks(value, group=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
