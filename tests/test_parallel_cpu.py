"""CPU suite: the N > 1 host path (sharding of independent filters + one all_gather) on gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_filters_partition():
    from ssme_amd.parallel import shard_filters
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            ids = []
            for r in range(world):
                first, cnt = shard_filters(n, world, r)
                ids += list(range(first, first + cnt))
            assert ids == list(range(n))
            sizes = [shard_filters(n, world, r)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard_filters(4096, 8, 3) == (1536, 512)      # BASELINE.json configs[3]: 512 filters per GPU
    with pytest.raises(ValueError):
        shard_filters(4, 2, 2)


def test_log_mean_exp_matches_reference_kat():
    from ssme_amd.parallel import log_mean_exp
    assert abs(log_mean_exp(np.full(10000, 3.0)) - 3.0) < 1e-3     # test/test_thread_pool.cpp:39-46
    assert np.isnan(log_mean_exp([1.0, np.nan]))
    assert log_mean_exp([-np.inf, -np.inf]) == -np.inf


def _worker(rank, world, port, n_filters, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ssme_amd.parallel import gather_logliks, log_mean_exp, max_over_ranks, shard_filters
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = shard_filters(n_filters, world, rank)
    local = -100.0 - 0.25 * np.arange(first, first + cnt)        # stand-in for this rank's filter log-likelihoods
    allv = gather_logliks(local, n_filters)
    tmax = max_over_ranks(1.0 + rank)
    q.put((rank, allv.tolist(), log_mean_exp(allv), tmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_filters", [2, 5])
def test_gather_logliks_gloo_world2(n_filters):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_filters, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = (-100.0 - 0.25 * np.arange(n_filters)).tolist()
    from ssme_amd.parallel import log_mean_exp
    for rank, allv, lme, tmax in res:
        assert allv == expect                       # every rank sees all filters, in filter-id order
        assert lme == log_mean_exp(expect)
        assert tmax == 2.0                          # MAX over ranks, as bench.py does for the timing


# ---- particle-sharded filter: host-side exchange logic (ssme_amd/sharded.py) ---------------------------------------
def test_exchange_plan_covers_exactly_the_planned_windows():
    from ssme_amd.sharded import exchange_plan
    rng = np.random.default_rng(0)
    for world, Bl in ((2, 4), (4, 8), (8, 2), (3, 5)):
        B = world * Bl
        for _ in range(50):
            lo = np.sort(rng.integers(0, B, world))
            hi = np.maximum(lo, np.sort(rng.integers(0, B, world)))
            plan = list(zip(lo.tolist(), hi.tolist()))
            all_sends = [exchange_plan(plan, Bl, r)[0] for r in range(world)]
            for r in range(world):
                sends, recvs = exchange_plan(plan, Bl, r)
                tiles = [t for first, cnt in recvs for t in range(first, first + cnt)]
                assert tiles == list(range(plan[r][0], plan[r][1] + 1))          # contiguous window in global order
                for s in range(world):
                    assert recvs[s] == all_sends[s][r] or (recvs[s][1] == 0 and all_sends[s][r][1] == 0)
                    first, cnt = recvs[s]
                    assert cnt == 0 or (s * Bl <= first and first + cnt <= (s + 1) * Bl)


def test_shard_layout_of_sizes_that_are_not_whole_tiles_per_rank():
    """ceil(B / world) tiles per rank; the last rank owns what is left (fewer tiles, a ragged last tile) and at least one; plans
    over such a layout (windows end at tile B - 1) still give contiguous windows."""
    from ssme_amd.sharded import exchange_plan, shard_layout, TILE
    from ssme_amd import SsmeError
    for n, world in ((16384 + 2048 + 77, 2), (10 * 2048 + 5, 3), (1 << 24, 8), (2049, 2), (8 * 8 * 2048 - 9000, 8), (100000, 7)):
        lay = [shard_layout(n, r, world) for r in range(world)]
        B, Bl = lay[0][1], lay[0][2]
        assert B == -(-n // TILE) and Bl == -(-B // world)
        assert sum(l[3] for l in lay) == B and sum(l[4] for l in lay) == n
        assert all(l[3] == Bl for l in lay[:-1]) and 1 <= lay[-1][3] <= Bl
        assert all(l[4] == Bl * TILE for l in lay[:-1])
        rng = np.random.default_rng(n)
        for _ in range(20):
            lo = np.sort(rng.integers(0, B, world))
            hi = np.maximum(lo, np.sort(rng.integers(0, B, world)))
            plan = list(zip(lo.tolist(), hi.tolist()))
            for r in range(world):
                sends, recvs = exchange_plan(plan, Bl, r)
                tiles = [t for first, cnt in recvs for t in range(first, first + cnt)]
                assert tiles == list(range(plan[r][0], plan[r][1] + 1))
    for n, world in ((4 * 2048, 3), (100, 2), (2048, 2), (6 * 2048, 5), (8 * 4 * 2048 - 9000, 8)):          # the last rank would own nothing
        with pytest.raises(SsmeError):
            shard_layout(n, 0, world)


def _exchange_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssme_amd.sharded import exchange_plan, exchange_tiles
    Bl, W = 3, 8
    local = (torch.arange(Bl * W, dtype=torch.float64).reshape(Bl, W) + 1000.0 * rank)
    plan = [(0, 4), (2, 5)]                                   # rank 0 needs tiles 0..4, rank 1 needs 2..5
    sends, recvs = exchange_plan(plan, Bl, rank)
    win = exchange_tiles(local, rank * Bl, sends, recvs, rank)
    q.put((rank, win.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_tiles_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Bl, W = 3, 8
    full = np.concatenate([np.arange(Bl * W, dtype=np.float64).reshape(Bl, W) + 1000.0 * r for r in range(2)])
    np.testing.assert_array_equal(got[0], full[0:5])
    np.testing.assert_array_equal(got[1], full[2:6])


def _halo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssme_amd.sharded import HaloBuffer, exchange_halos
    Bl, W, margin = 4, 8, 1
    res = {}
    # (a) both windows fit the margins; (b) rank 1's window reaches 2 tiles into rank 0 (over ITS margin only):
    # the decision must be the same on both ranks (the grouping of sends/receives must not be mixed)
    for name, plan in (("fits", [(0, 4), (3, 7)]), ("rank1_over", [(0, 4), (2, 7)])):
        hb = HaloBuffer(Bl, W, margin, torch.device("cpu"), torch.float64)
        hb.own().copy_(torch.arange(Bl * W, dtype=torch.float64).reshape(Bl, W) + 1000.0 * rank)
        got = exchange_halos([hb], rank * Bl, plan, Bl, rank)
        res[name] = (got, hb.buf.numpy().copy())
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_halos_decision_is_global_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_halo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Bl, W = 4, 8
    full = np.concatenate([np.arange(Bl * W, dtype=np.float64).reshape(Bl, W) + 1000.0 * r for r in range(2)])
    # fits: rank 0's window is tiles 0..4 (own 0..3 + right halo = tile 4), rank 1's 3..7 (left halo = tile 3)
    assert got[0]["fits"][0] == (0, 5) and got[1]["fits"][0] == (3, 5)
    np.testing.assert_array_equal(got[0]["fits"][1][1:6], full[0:5])
    np.testing.assert_array_equal(got[1]["fits"][1][0:5], full[3:8])
    # one rank over its margin => BOTH ranks report "does not fit" (and fall back to the assembled window together)
    assert got[0]["rank1_over"][0] is None and got[1]["rank1_over"][0] is None
