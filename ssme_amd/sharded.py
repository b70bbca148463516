"""One particle filter sharded over the GPUs of a node.

One process per GPU; rank g owns Bl = ceil(B / G) consecutive tiles of 2048 particles (the last rank what is left of the
B = ceil(N / 2048): `shard_layout`).  A time step is

    1. all_gather of the per-tile weight sums and maxima of step t-1 (B x 16 bytes in total).  Every rank then runs
       the SAME exact level-2 (global max, rescaled integer tile sums, exact scan): the global log-sum-exp, hence
       log p(y_{t-1} | y_{1:t-2}), and the global weight cdf over tiles.  This is the north star's "allreduce of the
       global log-weight sum", done as gather + deterministic integer scan so that the result does not depend on G.
    2. ssme_pf_shard_plan: which source tiles each rank's resampling targets fall into (sorted targets => one
       contiguous range per rank, mostly its own tiles when the weights are balanced).
    3. exchange of those tiles (integer cdf + particles, 32 KiB per tile) with grouped point-to-point sends: the
       "all-to-all for post-resample particle redistribution", done BEFORE the search so that every rank resamples
       exactly its own N/G offspring and no rebalancing step is needed afterwards.
    4. ssme_pf_shard_step: the unsharded filter's fused step kernel on this rank's tiles.

RNG counters use global particle indices and the Gamma tables global tile ids, so a sharded run is bit-identical to
`ParticleFilterBank.run_series` with the same N and seed, for every G (tests/test_sharded_gpu.py).

The reference has no distributed filter; the single-process semantics reproduced are BSFilter::filter's
(example/estimate_univ_svol.h:121-127).  Collectives go through torch.distributed: backend "nccl" (= RCCL over xGMI)
exchanges device tensors; backend "gloo" (CPU rehearsal, several ranks on one GPU) stages them through host memory.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import SsmeError

TILE = 2048

__all__ = ["ShardedParticleFilter", "ShardedLiuWest", "exchange_plan", "shard_layout", "TILE"]


def shard_layout(n_particles, rank, world):
    """(N, B, Bl, tiles this rank owns, particles this rank owns): B = ceil(N / 2048) tiles, Bl = ceil(B / world) rows per rank
    in every layout; rank g owns tiles [g Bl, min((g + 1) Bl, B)) -- the last rank may own fewer tiles and a ragged last one,
    and must own at least one (include/ssme_pf.h: ssme_pf_shard_create)."""
    n = int(n_particles)
    B = (n + TILE - 1) // TILE
    Bl = (B + world - 1) // world
    if n < 1 or (world - 1) * Bl >= B:
        raise SsmeError(capi.ERR_UNSUPPORTED, f"{n} particles = {B} tiles of {TILE} do not give each of {world} ranks a tile "
                                              f"({Bl} per rank)")
    own = min(Bl, B - rank * Bl)
    return n, B, Bl, own, min(own * TILE, n - rank * Bl * TILE)


def exchange_plan(lo_hi, tiles_per_rank, rank):
    """Who sends which tiles to whom, from the planner's inclusive source ranges.

    lo_hi: sequence of (lo, hi) per destination rank (global tile ids).  Returns (sends, recvs):
      sends[d] = (first, count) of MY tiles (global ids) that rank d needs (count may be 0);
      recvs[s] = (first, count) of rank s's tiles that I need.
    The received pieces, concatenated in rank order, are the contiguous window lo_hi[rank][0] .. lo_hi[rank][1].
    """
    world = len(lo_hi)
    my_first, my_last = rank * tiles_per_rank, (rank + 1) * tiles_per_rank - 1
    sends, recvs = [], []
    for d in range(world):
        lo, hi = int(lo_hi[d][0]), int(lo_hi[d][1])
        a, b = max(lo, my_first), min(hi, my_last)
        sends.append((a, b - a + 1) if b >= a else (my_first, 0))
    lo, hi = int(lo_hi[rank][0]), int(lo_hi[rank][1])
    for s in range(world):
        s_first, s_last = s * tiles_per_rank, (s + 1) * tiles_per_rank - 1
        a, b = max(lo, s_first), min(hi, s_last)
        recvs.append((a, b - a + 1) if b >= a else (s_first, 0))
    assert sum(c for _, c in recvs) == hi - lo + 1
    return sends, recvs


def exchange_tiles(local, my_first_tile, sends, recvs, rank, group=None, stage_through_host=False):
    """Grouped point-to-point exchange of whole tiles.  local: [tiles_per_rank, W] tensor (this rank's tiles);
    returns the window [sum(recv counts), W] in global tile order.  Works for any backend with send/recv."""
    import torch
    import torch.distributed as dist
    world = len(sends)
    total = sum(c for _, c in recvs)
    comm_dev = torch.device("cpu") if stage_through_host else local.device
    window = torch.empty((total, local.shape[1]), dtype=local.dtype, device=comm_dev)
    src = local.to(comm_dev) if stage_through_host else local
    ops, off = [], 0
    for s in range(world):
        first, cnt = recvs[s]
        if cnt:
            piece = window[off:off + cnt]
            if s == rank:
                piece.copy_(src[first - my_first_tile:first - my_first_tile + cnt])
            else:
                ops.append(dist.P2POp(dist.irecv, piece, s if group is None else dist.get_global_rank(group, s), group))
            off += cnt
    for d in range(world):
        first, cnt = sends[d]
        if cnt and d != rank:
            piece = src[first - my_first_tile:first - my_first_tile + cnt].contiguous()
            ops.append(dist.P2POp(dist.isend, piece, d if group is None else dist.get_global_rank(group, d), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return window.to(local.device) if stage_through_host else window


def gather_flat(local_flat, world, group=None, stage_through_host=False):
    """all_gather of one contiguous 1-D tensor per rank -> [world, n] on the local tensor's device (one collective:
    all_gather_into_tensor where the backend has it (nccl), the list form otherwise)."""
    import torch
    import torch.distributed as dist
    src = local_flat.cpu() if stage_through_host else local_flat
    out = torch.empty((world, src.numel()), dtype=src.dtype, device=src.device)
    if stage_through_host:
        parts = list(out.unbind(0))
        dist.all_gather(parts, src.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    return out.to(local_flat.device) if stage_through_host else out


def make_native_comm(rank, world, device_index, group=None):
    """An RCCL communicator for the C++ drivers: rank 0 makes the unique id, torch.distributed ships its 128 bytes."""
    import torch.distributed as dist
    L = capi.lib()
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        capi.check(L.ssme_shard_comm_get_unique_id(buf))
    box = [bytes(buf)]
    if world > 1:
        dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    idbuf = (C.c_ubyte * 128).from_buffer_copy(box[0])
    comm = C.c_void_p()
    capi.check(L.ssme_shard_comm_init(idbuf, rank, world, device_index, C.byref(comm)))
    return comm


class HaloBuffer:
    """This rank's tiles with room for neighbours' tiles on both sides: [margin | own tiles | margin] rows of W doubles.

    The step kernels write the rank's output tiles straight into the middle and read their source window [lo, hi] from the
    same buffer, so the rank's own tiles are never copied: only the halo tiles that other ranks own travel.  A window that
    does not fit the margins (very unbalanced weights) falls back to `exchange_tiles` (a freshly assembled window)."""

    def __init__(self, tiles, width, margin, device, dtype, storage=None):
        import torch
        self.tiles, self.width, self.margin = tiles, width, margin
        self.buf = torch.zeros((tiles + 2 * margin, width), dtype=dtype, device=device) if storage is None else storage

    def own(self):
        return self.buf[self.margin:self.margin + self.tiles]

    def fits(self, lo, hi, tile0):
        return tile0 - lo <= self.margin and hi - (tile0 + self.tiles - 1) <= self.margin

    def rows(self, first, count, tile0):
        """View of global tiles first .. first+count-1 (own or halo)."""
        a = self.margin + (first - tile0)
        return self.buf[a:a + count]


def exchange_halos(bufs, tile0, plan, tiles_per_rank, rank, group=None, stage_through_host=False):
    """One grouped exchange for several HaloBuffers that share a plan: every piece another rank owns is received straight
    into the margin rows, every piece another rank needs is sent from the own rows.  Returns (win_tile0, win_tiles) or
    None if the window does not fit the margins."""
    import torch.distributed as dist
    lo, hi = plan[rank]
    # The decision is GLOBAL: every rank holds the whole plan and evaluates every rank's window against the (identical)
    # margins, so that all ranks take the same path and post the same grouping of sends/receives (a mixed grouping is
    # benign on gloo but not something to rely on under RCCL).
    if not all(b.fits(plan[g][0], plan[g][1], g * tiles_per_rank) for g in range(len(plan)) for b in bufs):
        return None
    sends, recvs = exchange_plan(plan, tiles_per_rank, rank)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    ops, staged = [], []
    for b in bufs:
        for s, (first, cnt) in enumerate(recvs):
            if cnt and s != rank:
                dst = b.rows(first, cnt, tile0)
                if stage_through_host:
                    tmp = dst.cpu()
                    staged.append((dst, tmp))
                    ops.append(dist.P2POp(dist.irecv, tmp, peer(s), group))
                else:
                    ops.append(dist.P2POp(dist.irecv, dst, peer(s), group))
        for d, (first, cnt) in enumerate(sends):
            if cnt and d != rank:
                src = b.rows(first, cnt, tile0)
                ops.append(dist.P2POp(dist.isend, src.cpu() if stage_through_host else src, peer(d), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for dst, tmp in staged:
        dst.copy_(tmp)
    return lo, hi - lo + 1


class ShardedParticleFilter:
    """Bootstrap filter with its N particles sharded over the ranks of a torch.distributed group."""

    def __init__(self, model, n_particles, seed=0, resampler=capi.RESAMP_MULTINOMIAL, device=None, group=None, filter_id=0,
                 resamp_sched=1):
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.stage = self.backend != "nccl"                     # gloo rehearsal: collectives on host copies
        self.n, self.B, self.Bl, self.Bown, self.n_local = shard_layout(n_particles, self.rank, self.world)
        self.tile0 = self.rank * self.Bl
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.resamp_sched = int(resamp_sched)
        cfg = capi.Config(model=model, n_particles=n_particles, n_filters=1, dtype=capi.F64, resampler=resampler,
                          resamp_sched=self.resamp_sched, seed=seed, device=self.device.index or 0, first_filter_id=filter_id)
        self._h = C.c_void_p()
        capi.check(capi.lib().ssme_pf_shard_create(C.byref(cfg), self.rank, self.world, C.byref(self._h)))
        # one side stream for kernels, staging copies and (nccl) collectives: torch's default stream has handle 0,
        # which ssme_pf_set_stream reads as "use the handle's own stream"
        self._stream = torch.cuda.Stream(self.device)
        self._chk(capi.lib().ssme_pf_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)))
        f64 = dict(dtype=torch.float64, device=self.device)
        # particles and tile-local integer cdf (integers < 2^53 in fp64) after the last step: ping-pong halo buffers
        margin = min(self.B - self.Bl, max(2, self.Bl // 2)) if self.world > 1 else 0
        self._hx = [HaloBuffer(self.Bl, TILE, margin, self.device, torch.float64) for _ in range(2)]
        self._hc = [HaloBuffer(self.Bl, TILE, margin, self.device, torch.float64) for _ in range(2)]
        self._cur = 0
        self.tiles_loc = torch.zeros((2, self.Bl), **f64)       # row 0: tile sums, row 1: tile maxima
        self.tiles_all = torch.zeros((2, self.world * self.Bl), **f64)      # gathered: world x Bl entries, the first B are tiles
        self.anc = None
        self.exchanged_tiles = 0                                # tiles received from other ranks (statistics)
        self._T = 0
        torch.cuda.synchronize(self.device)

    def close(self):
        if getattr(self, "_comm", None) is not None:
            capi.lib().ssme_shard_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().ssme_pf_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _chk(self, status):
        if status != capi.OK:
            msg = capi.lib().ssme_pf_strerror(status).decode()
            if status in (capi.ERR_HIP, capi.ERR_STATE, capi.ERR_UNSUPPORTED):
                msg += " (" + capi.lib().ssme_pf_last_error(self._h).decode() + ")"
            raise SsmeError(status, msg)

    def set_params(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64).reshape(1, -1)
        self._chk(capi.lib().ssme_pf_set_params(self._h, capi.dptr(th), th.shape[1], 1))

    def record_ancestors(self, on=True):
        import torch
        self.anc = torch.zeros((self.Bl, TILE), dtype=torch.int32, device=self.device) if on else None

    # ---- collectives --------------------------------------------------------------------------------------------
    def _gather_tiles(self):
        """tiles_all[:, g*Bl:(g+1)*Bl] = rank g's tiles_loc."""
        G = gather_flat(self.tiles_loc.reshape(-1), self.world, self.group, self.stage)          # [world, 2 * Bl]
        self.tiles_all[0].view(self.world, self.Bl).copy_(G[:, :self.Bl])
        self.tiles_all[1].view(self.world, self.Bl).copy_(G[:, self.Bl:])

    def _ptr(self, t):
        return C.c_void_p(t.data_ptr())

    # ---- the series loop ------------------------------------------------------------------------------------------
    def run_series(self, y, z=None):
        """log p(y_{1:T}) of the sharded filter; identical on every rank."""
        import torch
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._stream):
            return self._run_series(y, z)

    def _run_series(self, y, z):
        import torch
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        T = yv.size
        L = capi.lib()
        self._chk(L.ssme_pf_shard_prepare(self._h, capi.dptr(yv), capi.dptr(zv), T))
        ts, tm = self.tiles_all[0], self.tiles_all[1]
        lo_hi = (C.c_int32 * (2 * self.world))()
        self.exchanged_tiles = 0
        for t in range(T):
            hx, hc = self._hx[self._cur], self._hc[self._cur]            # sources: the last step's outputs (+ halos)
            ox, oc = self._hx[self._cur ^ 1], self._hc[self._cur ^ 1]    # this step's outputs go to the other pair
            if t == 0:
                xw = cw = None
                win0 = 0
            elif t % self.resamp_sched != 0:
                # no resampling draw closes step t - 1: every particle continues itself with its carried log-weight; the
                # sources are this rank's own tiles, nothing travels but the tile sums (for the step's log-likelihood)
                self._gather_tiles()
                if self.B > 1024:
                    self._chk(L.ssme_pf_shard_plan(self._h, self._ptr(ts), self._ptr(tm), t, lo_hi))     # split level-2: accounts the step
                xw, cw, win0 = hx.own(), hc.own(), self.tile0
            else:
                self._gather_tiles()
                self._chk(L.ssme_pf_shard_plan(self._h, self._ptr(ts), self._ptr(tm), t, lo_hi))
                plan = [(lo_hi[2 * g], lo_hi[2 * g + 1]) for g in range(self.world)]
                got = exchange_halos([hc, hx], self.tile0, plan, self.Bl, self.rank, self.group, self.stage)
                if got is None:                                          # window wider than the margins: assemble it
                    sends, recvs = exchange_plan(plan, self.Bl, self.rank)
                    cw = exchange_tiles(hc.own(), self.tile0, sends, recvs, self.rank, self.group, self.stage)
                    xw = exchange_tiles(hx.own(), self.tile0, sends, recvs, self.rank, self.group, self.stage)
                    win0 = plan[self.rank][0]
                else:
                    win0, wt = got
                    cw, xw = hc.rows(win0, wt, self.tile0), hx.rows(win0, wt, self.tile0)
                lo_r, hi_r = plan[self.rank]
                self.exchanged_tiles += max(0, min(hi_r + 1, self.tile0) - lo_r) + max(0, hi_r - max(lo_r - 1, self.tile0 + self.Bl - 1))
            self._chk(L.ssme_pf_shard_step(
                self._h, t, None if xw is None else self._ptr(xw), None if cw is None else self._ptr(cw), win0,
                self._ptr(ts), self._ptr(tm), self._ptr(ox.own()), self._ptr(oc.own()), self._ptr(self.tiles_loc[0]),
                self._ptr(self.tiles_loc[1]), None if self.anc is None else self._ptr(self.anc)))
            self._cur ^= 1
        self._gather_tiles()
        self._chk(L.ssme_pf_shard_finalize(self._h, T - 1, self._ptr(ts), self._ptr(tm)))
        self._T = T
        out = np.empty(1)
        self._chk(L.ssme_pf_get_loglik(self._h, capi.dptr(out)))
        return float(out[0])

    def per_step(self):
        out = np.empty((1, self._T))
        self._chk(capi.lib().ssme_pf_get_per_step(self._h, capi.dptr(out), self._T))
        return out[0]

    def local_particles(self):
        return self._hx[self._cur].own().reshape(-1)[:self.n_local].cpu().numpy()

    def local_cdf(self):
        return self._hc[self._cur].own().reshape(-1)[:self.n_local].cpu().numpy()

    # ---- the same loop in C++ over RCCL (ssme_pf_shard_run_series): no Python, no host synchronisation per step ----
    def _native_comm(self):
        """An RCCL communicator for the C++ driver: rank 0 makes the unique id, torch.distributed ships its 128 bytes."""
        if getattr(self, "_comm", None) is None:
            self._comm = make_native_comm(self.rank, self.world, self.device.index or 0, self.group)
        return self._comm

    def run_series_native(self, y, z=None, mode=0):
        """log p(y_{1:T}) through the C++ driver.  mode 0: fixed-halo fast path with an exact rerun if a window ever left
        the halo; 1: fast path only; 2: exact (host-planned) path.  Needs one GPU per rank (RCCL)."""
        import torch
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        comm = self._native_comm()
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        out = np.empty(1)
        self._chk(capi.lib().ssme_pf_shard_run_series(self._h, comm, capi.dptr(yv), capi.dptr(zv), yv.size, int(mode), capi.dptr(out)))
        self._T = yv.size
        return float(out[0])

    def native_state(self):
        """(particles, integer cdf, path, tiles received) of this rank after run_series_native."""
        n = self.n_local
        x, cdf = np.empty(n), np.empty(n, dtype=np.uint64)
        path, ex = C.c_int32(), C.c_int64()
        self._chk(capi.lib().ssme_pf_shard_download(self._h, capi.dptr(x), capi.u64ptr(cdf), C.byref(path), C.byref(ex)))
        return x, cdf, path.value, ex.value


class ShardedLiuWest:
    """Liu-West filter (svol_lw_1_par: test/test_liu_west.cpp:22-157) with its particles sharded over the ranks of a
    torch.distributed group -- BASELINE.json configs[4].  Two exchanges per time step (module docstring scheme, twice):

        gather (tsumB, tmaxB)                -> plan 0 -> windows of (cdfB, x, theta)        -> stage 1 (resample, lw1, moments)
        gather (tsumA, tmaxA, 14 moments)    -> mid (theta-bar, Cholesky; every rank, same bits)
                                             -> plan 1 -> windows of (cdfA, lw1, x, theta)   -> stage 2 (k draw, jitter, fSamp)

    Bit-identical to the unsharded `svol_lw_1_par.run_series` with the same N and seed (tests/test_sharded_gpu.py).
    """

    def __init__(self, delta, phi_l, phi_u, mu_l, mu_u, sig_l, sig_u, rho_l, rho_u, nparts, seed=0, transforms=(2, 0, 3, 1),
                 device=None, group=None, filter_id=0, form=0, rs=1):
        """form 0: auxiliary-particle form (LWFilterWithCovs); 1: SISR form (LWFilter2WithCovs, svol_lw_2_par): no first-stage
        weights and no k draw, so a step has ONE window exchange (for the resampling draw) instead of two.
        rs: the resampling schedule m_rs (liu_west_filter.h:1139-1140): steps without a resampling draw exchange nothing for
        stage 1 (every particle continues itself with its carried second-stage weight)."""
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.stage = dist.get_backend(group) != "nccl"
        self.n, self.B, self.Bl, self.Bown, self.n_local = shard_layout(nparts, self.rank, self.world)
        self.tile0 = self.rank * self.Bl
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.form, self.rs = int(form), int(rs)
        cfg = capi.LwConfig(n_particles=nparts, n_filters=1, seed=seed, device=self.device.index or 0, first_filter_id=filter_id,
                            delta=delta, form=self.form, resamp_sched=self.rs)
        cfg.transforms[:] = list(transforms)
        cfg.prior_lo[:] = [phi_l, mu_l, sig_l, rho_l]
        cfg.prior_hi[:] = [phi_u, mu_u, sig_u, rho_u]
        self._h = C.c_void_p()
        capi.check(capi.lib().ssme_lw_shard_create(C.byref(cfg), self.rank, self.world, C.byref(self._h)))
        self._stream = torch.cuda.Stream(self.device)
        self._chk(capi.lib().ssme_lw_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)))
        f64 = dict(dtype=torch.float64, device=self.device)
        Bl = self.Bl
        margin = min(self.B - Bl, max(2, Bl // 2)) if self.world > 1 else 0
        self._rows = Bl + 2 * margin
        mk = lambda: HaloBuffer(Bl, TILE, margin, self.device, torch.float64)
        # stage-2 outputs / stage-1 sources, and stage-1 outputs / stage-2 sources; theta: one 32-byte record per particle,
        # i.e. rows of 2048 x 4 doubles that travel with their tile
        self.xB, self.cdfB, self.xr, self.lw1, self.cdfA = mk(), mk(), mk(), mk(), mk()
        self.thB = HaloBuffer(Bl, TILE * 4, margin, self.device, torch.float64)
        self.thr = HaloBuffer(Bl, TILE * 4, margin, self.device, torch.float64)
        self.tilesB = torch.zeros((2, Bl), **f64)            # rows: tile sums, tile maxima (second-stage weights)
        self._locA = torch.zeros(18 * Bl, **f64)             # stage-1 outputs in one buffer: tile sums | tile maxima | 16 moments per tile
        self.tilesA = self._locA[:2 * Bl].view(2, Bl)
        self.mom = self._locA[2 * Bl:].view(Bl, 16)
        self.allB = torch.zeros((2, self.world * Bl), **f64)   # gathered: world x Bl entries, the first B are tiles
        self.allA = torch.zeros((2, self.world * Bl), **f64)
        self.mom_all = torch.zeros((self.world * Bl, 16), **f64)
        self.exchanged_tiles = 0
        self._T = 0
        torch.cuda.synchronize(self.device)

    def close(self):
        if getattr(self, "_comm", None) is not None:
            capi.lib().ssme_shard_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().ssme_lw_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _chk(self, status):
        if status != capi.OK:
            msg = capi.lib().ssme_pf_strerror(status).decode()
            if status == capi.ERR_HIP:
                msg += " (" + capi.lib().ssme_lw_last_error(self._h).decode() + ")"
            raise SsmeError(status, msg)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    def _gather_B(self):
        G = gather_flat(self.tilesB.reshape(-1), self.world, self.group, self.stage)               # [world, 2 Bl]
        self.allB[0].view(self.world, self.Bl).copy_(G[:, :self.Bl])
        self.allB[1].view(self.world, self.Bl).copy_(G[:, self.Bl:])

    def _gather_A(self):
        Bl = self.Bl
        G = gather_flat(self._locA, self.world, self.group, self.stage)                             # [world, 18 Bl]: ONE collective
        self.allA[0].view(self.world, Bl).copy_(G[:, :Bl])
        self.allA[1].view(self.world, Bl).copy_(G[:, Bl:2 * Bl])
        self.mom_all.view(self.world, Bl * 16).copy_(G[:, 2 * Bl:])

    def _windows(self, which, t, tiles_all, halos2d, halo_th):
        """plan + exchange for one draw.  Returns (win_tile0, window tiles, [2-D source windows], theta source window).
        Own tiles stay where they are; only halo tiles travel."""
        import torch
        lo_hi = (C.c_int32 * (2 * self.world))()
        self._chk(capi.lib().ssme_lw_shard_plan(self._h, which, t, self._ptr(tiles_all[0]), self._ptr(tiles_all[1]), lo_hi))
        plan = [(lo_hi[2 * g], lo_hi[2 * g + 1]) for g in range(self.world)]
        lo_r, hi_r = plan[self.rank]
        self.exchanged_tiles += max(0, min(hi_r + 1, self.tile0) - lo_r) + max(0, hi_r - max(lo_r - 1, self.tile0 + self.Bl - 1))
        got = exchange_halos(halos2d + [halo_th], self.tile0, plan, self.Bl, self.rank, self.group, self.stage)
        if got is not None:
            w0, wt = got
            return w0, wt, [hb.rows(w0, wt, self.tile0) for hb in halos2d], halo_th.rows(w0, wt, self.tile0)
        sends, recvs = exchange_plan(plan, self.Bl, self.rank)             # window wider than the margins: assemble it
        wins = [exchange_tiles(hb.own(), self.tile0, sends, recvs, self.rank, self.group, self.stage) for hb in halos2d]
        wth = exchange_tiles(halo_th.own(), self.tile0, sends, recvs, self.rank, self.group, self.stage)
        self._keep = (wins, wth)                                           # keep the assembled windows alive until the launch
        return lo_r, hi_r - lo_r + 1, wins, wth

    def run_series(self, y, z=None):
        import torch
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._stream):
            return self._run_series(y, z)

    def _run_series(self, y, z):
        L, p = capi.lib(), self._ptr
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        T = yv.size
        self._chk(L.ssme_lw_shard_prepare(self._h, capi.dptr(yv), capi.dptr(zv), T))
        self.exchanged_tiles = 0
        thB_own, thr_own = p(self.thB.own()), p(self.thr.own())
        self._chk(L.ssme_lw_shard_init(self._h, p(self.xB.own()), thB_own, p(self.cdfB.own()), p(self.tilesB[0]), p(self.tilesB[1])))
        for t in range(1, T):
            self._gather_B()
            if t % self.rs == 0:
                w0, rows, (w_x, w_cdf), w_th = self._windows(0, t, self.allB, [self.xB, self.cdfB], self.thB)
            else:       # no resampling draw closes step t - 1: the sources of stage 1 are this rank's own particles
                if self.B > 1024:      # split level-2: the plan provides (m, S) of the second-stage weights for the accounting
                    lo_hi = (C.c_int32 * (2 * self.world))()
                    self._chk(L.ssme_lw_shard_plan(self._h, 0, t, p(self.allB[0]), p(self.allB[1]), lo_hi))
                w0, rows = self.tile0, self.Bl
                w_x, w_cdf, w_th = self.xB.own(), self.cdfB.own(), self.thB.own()
            self._chk(L.ssme_lw_shard_stage1(self._h, t, w0, rows, p(w_x), p(w_th), p(w_cdf), p(self.allB[0]), p(self.allB[1]),
                                             p(self.xr.own()), thr_own, p(self.lw1.own()), p(self.cdfA.own()), p(self.tilesA[0]),
                                             p(self.tilesA[1]), p(self.mom), None))
            self._gather_A()
            # the plan of the k draw first: with the split level-2 it also provides the (m, S) that mid turns into lse1
            if self.form == 0:
                w0, rows, (w_x, w_lw1, w_cdf), w_th = self._windows(1, t, self.allA, [self.xr, self.lw1, self.cdfA], self.thr)
            else:       # SISR form: every particle continues itself -- the sources of stage 2 are this rank's own stage-1 outputs
                w0, rows = self.tile0, self.Bl
                w_x, w_lw1, w_cdf, w_th = self.xr.own(), self.lw1.own(), self.cdfA.own(), self.thr.own()
            self._chk(L.ssme_lw_shard_mid(self._h, t, p(self.allA[0]), p(self.allA[1]), p(self.mom_all)))
            self._chk(L.ssme_lw_shard_stage2(self._h, t, w0, rows, p(w_x), p(w_th), p(w_lw1), p(w_cdf), p(self.allA[0]),
                                             p(self.allA[1]), p(self.xB.own()), thB_own, p(self.cdfB.own()), p(self.tilesB[0]),
                                             p(self.tilesB[1]), None))
        self._gather_B()
        self._chk(L.ssme_lw_shard_finalize(self._h, T - 1, p(self.allB[0]), p(self.allB[1])))
        self._T = T
        out = np.empty(1)
        self._chk(L.ssme_lw_get_loglik(self._h, capi.dptr(out)))
        return float(out[0])

    def per_step(self):
        out = np.empty((1, self._T))
        self._chk(capi.lib().ssme_lw_get_per_step(self._h, capi.dptr(out), self._T))
        return out[0]

    def local_particles(self):
        return self.xB.own().reshape(-1)[:self.n_local].cpu().numpy()

    def local_theta(self):
        """[4, n_local]: transformed parameters of this rank's particles (the device keeps [particle][4] records)."""
        return self.thB.own().reshape(-1, 4)[:self.n_local].t().contiguous().cpu().numpy()

    # ---- the same loop in C++ over RCCL (ssme_lw_shard_run_series) ----
    def run_series_native(self, y, z=None):
        """log p(y_{1:T}) through the C++ driver (fixed-halo exchange, no host synchronisation per step).  If a resampling
        window ever leaves the halo ON ANY RANK the driver says so ON EVERY RANK (the per-rank flags are reduced by one
        ncclAllReduce after the time loop) and the exact Python-driven loop runs instead -- on all ranks together, so the
        collectives stay matched.  One GPU per rank."""
        import torch
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        if getattr(self, "_comm", None) is None:
            self._comm = make_native_comm(self.rank, self.world, self.device.index or 0, self.group)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        out = np.empty(1)
        st = capi.lib().ssme_lw_shard_run_series(self._h, self._comm, capi.dptr(yv), capi.dptr(zv), yv.size, capi.dptr(out))
        self.native_path = "fixed halo"
        if st == capi.ERR_STATE:
            self.native_path = "exact (Python-driven) after a window left the halo"
            return self.run_series(y, z)
        self._chk(st)
        self._T = yv.size
        return float(out[0])

    def native_state(self):
        n = self.n_local
        x, th = np.empty(n), np.empty((4, n))
        ex = C.c_int64()
        self._chk(capi.lib().ssme_lw_shard_download(self._h, capi.dptr(x), capi.dptr(th), C.byref(ex)))
        return x, th, ex.value
