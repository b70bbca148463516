"""Host-side mirror of the reference's filter interface for the GPU core.

Names and argument meaning follow the reference so that parity tests read like its own:
  svol_bs(phi, beta, sigma) / svol_bs.from_pack(pp)      example/univ_svol_bootstrap_filter.h:34-35,46-61
  .filter(y) / .getLogCondLike()                         example/estimate_univ_svol.h:124-125
  svol_leverage(phi, mu, sigma, rho).filter(y, z)        test/test_pswarm.cpp:33-76
  .getExpectations()                                     include/ssme/pswarm_filter.h:87-89
  log_like_eval(theta, data)                             example/estimate_univ_svol.h:108-131
Every call goes through the C ABI (include/ssme_pf.h); there is no CPU path in this package.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import (MODEL_LIN_GAUSS, MODEL_SVOL, MODEL_SVOL_LEVERAGE, MODEL_USER0, RESAMP_MULTINOMIAL, RESAMP_MULTINOMIAL_IID,
                    RESAMP_STRATIFIED, RESAMP_SYSTEMATIC, SsmeError)

__all__ = ["default_tile", "ParticleFilterBank", "svol_bs", "svol_leverage", "lin_gauss_bs", "log_like_eval", "svol_lw_1_par", "svol_lw_2_par", "SwarmWithCovs", "Swarm", "svol_swarm_1",
           "TR_NULL", "TR_TWICE_FISHER", "TR_LOGIT", "TR_LOG",
           "MODEL_SVOL", "MODEL_SVOL_LEVERAGE", "MODEL_LIN_GAUSS", "MODEL_USER0", "RESAMP_MULTINOMIAL", "RESAMP_SYSTEMATIC",
           "RESAMP_STRATIFIED", "RESAMP_MULTINOMIAL_IID", "SsmeError"]


def default_tile(n_particles, bank_filters=1):
    """The tile size tile = 0 stands for (ssme_pf_default_tile): a function of N and the size of the whole bank of filters."""
    return int(capi.lib().ssme_pf_default_tile(int(n_particles), int(bank_filters)))


class ParticleFilterBank:
    """R independent bootstrap filters of N particles on one GPU (one C-ABI handle).

    R plays the role of thread_pool's num_pfilters (include/ssme/thread_pool.h:189-215) or of
    the swarm's nparamparts (include/ssme/pswarm_filter.h:280-304).
    """

    def __init__(self, model, n_particles, n_filters=1, seed=0, resampler=RESAMP_MULTINOMIAL, resamp_sched=1,
                 device=0, first_filter_id=0, tile=0, dtype=capi.F64, n_filters_total=0):
        """tile: particles per tile, 0 = chosen from (N, bank size), or 2048 / 1024 / 512.
        n_filters_total: size of the whole bank when this handle holds only a part of it (filters dealt to several GPUs
        or handles): the default tile then follows the bank, and a filter's results do not depend on the split.
        dtype: F64, or F32 = float at the boundary (inputs and outputs rounded to float, fp64 arithmetic; ssme_pf.h)."""
        self._h = C.c_void_p()
        self.model, self.n, self.r = int(model), int(n_particles), int(n_filters)
        self._last_T = 0
        cfg = capi.Config(model=model, n_particles=n_particles, n_filters=n_filters, dtype=dtype,
                          resampler=resampler, resamp_sched=resamp_sched, seed=seed, device=device,
                          first_filter_id=first_filter_id, tile_particles=tile, n_filters_total=n_filters_total)
        capi.check(capi.lib().ssme_pf_create(C.byref(cfg), C.byref(self._h)))
        t, b = C.c_int32(), C.c_int32()
        capi.check(capi.lib().ssme_pf_get_layout(self._h, C.byref(t), C.byref(b)))
        self.tile, self.n_tiles = t.value, b.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().ssme_pf_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _chk(self, status):
        capi.check(status, self._h)

    def set_params(self, theta):
        """theta: (n_theta,) shared by all filters or (R, n_theta); untransformed parameters."""
        th = capi.as_f64(theta)
        rows = 1 if th.ndim == 1 else th.shape[0]
        self._chk(capi.lib().ssme_pf_set_params(self._h, capi.dptr(th), th.shape[-1], rows))

    def reset(self):
        self._chk(capi.lib().ssme_pf_reset(self._h))

    def set_small_series(self, enable=True):
        """One-tile filters (N <= 2048): whole series in one launch (default) or the tiled per-step kernel."""
        self._chk(capi.lib().ssme_pf_set_small_series(self._h, 1 if enable else 0))

    def set_seed(self, seed):
        """New random stream for the next evaluation (a fresh likelihood estimate per PMMH proposal); resets the filters."""
        self._chk(capi.lib().ssme_pf_set_seed(self._h, int(seed)))

    def set_debug(self, record_ancestors=True, keep_logw=True, split_level2=None):
        """Parity/debug: record ancestor indices and/or keep the log-weights in device memory; split_level2 forces the
        one-launch-per-filter level-2 (True) or the in-kernel one (False, up to 2048 tiles); None chooses by size.
        "tables": the split level-2 with its tables and source ranges written by the level-2 kernels (what sharded and
        Liu-West filters run; two launches above 1024 tiles) instead of one launch + ranges found by the step kernel."""
        pol = 0 if split_level2 is None else (20 if split_level2 == "tables" else (4 if split_level2 else 8))   # None: by size
        self._chk(capi.lib().ssme_pf_set_debug(self._h, (1 if record_ancestors else 0) | (2 if keep_logw else 0) | pol))

    def set_graph_mode(self, on=True):
        self._chk(capi.lib().ssme_pf_set_graph_mode(self._h, int(on)))

    def set_tuning(self, threads_per_tile=512):
        self._chk(capi.lib().ssme_pf_set_tuning(self._h, int(threads_per_tile)))

    def _dims(self):
        """(dim_x, dim_y): 1, 1 unless this is a user model with a vector state / observation (csrc/model_api.h)."""
        if getattr(self, "_dxy", None) is None:
            self._dxy = (1, 1)
            if self.model == capi.MODEL_USER0:
                dx, dy = C.c_int32(), C.c_int32()
                self._chk(capi.lib().ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)))
                self._dxy = (dx.value, dy.value)
        return self._dxy

    def step(self, y, z=None):
        """One filter(y[, z]) on every filter; returns the R log conditional likelihoods.  y: dim_y values."""
        yv = np.ascontiguousarray(np.ravel(y), dtype=np.float64)
        assert yv.size == self._dims()[1]
        zv = None if z is None else np.array([z], dtype=np.float64)
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_pf_step(self._h, capi.dptr(yv), capi.dptr(zv), capi.dptr(out)))
        return out

    def run_series(self, y, z=None):
        """The log_like_eval loop for all R filters; returns R log-likelihoods.  y: [T], or [T, dim_y] for a vector observation."""
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        T = yv.size // self._dims()[1]
        assert T * self._dims()[1] == yv.size
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_pf_run_series(self._h, capi.dptr(yv), capi.dptr(zv), T, capi.dptr(out)))
        self._last_T = T
        return out

    def per_step(self):
        T = self._last_T
        out = np.empty((self.r, T))
        self._chk(capi.lib().ssme_pf_get_per_step(self._h, capi.dptr(out), T))
        return out

    def loglik(self):
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_pf_get_loglik(self._h, capi.dptr(out)))
        return out

    def log_mean_exp(self):
        out = np.empty(1)
        self._chk(capi.lib().ssme_pf_log_mean_exp(self._h, capi.dptr(out)))
        return float(out[0])

    def expectations(self, functional):
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_pf_get_expectations(self._h, functional, capi.dptr(out)))
        return out

    def expectations_multi(self, functionals):
        """E[h_i] of every filter for up to 4 built-in functionals in one pass: [n, R]."""
        fs = np.ascontiguousarray(functionals, dtype=np.int32)
        out = np.empty((fs.size, self.r))
        self._chk(capi.lib().ssme_pf_get_expectations_multi(self._h, fs.ctypes.data_as(C.POINTER(C.c_int32)), fs.size,
                                                            capi.dptr(out)))
        return out

    def swarm_aggregate(self, functionals=(), num_threads=0):
        """(mean over filters of the last log conditional likelihoods, [mean expectations]) reduced on the device.
        num_threads > 0: the reference's mean of per-thread means for a pool of that many workers (member i on thread
        i % num_threads, pswarm_filter.h:96-160) -- the plain mean unless num_threads does not divide the member count."""
        fs = np.ascontiguousarray(functionals, dtype=np.int32)
        ll, ex = np.empty(1), np.empty(max(fs.size, 1))
        self._chk(capi.lib().ssme_pf_swarm_aggregate_threads(self._h, fs.ctypes.data_as(C.POINTER(C.c_int32)), fs.size, int(num_threads),
                                                             capi.dptr(ll), capi.dptr(ex)))
        return float(ll[0]), ex[:fs.size].tolist()

    def weights(self, f=0):
        """(x, w) of filter f after the last step for host-side functionals: w = exp(logw - max logw); x: [dim_x, N] for a vector model."""
        dx = self._dims()[0]
        x, w = (np.empty(self.n) if dx == 1 else np.empty((dx, self.n))), np.empty(self.n)
        self._chk(capi.lib().ssme_pf_download_weights(self._h, f, capi.dptr(x), capi.dptr(w)))
        return x, w

    def state(self, f=0, ancestors=False, logw=True):
        """Parity/debug view of filter f after the last step (cdf and tile sums are exact uint64)."""
        n = self.n
        dx = self._dims()[0]
        x = np.empty(n) if dx == 1 else np.empty((dx, n))             # vector states: one row per component
        lw = np.empty(n) if logw else None
        cdf = np.empty(n, dtype=np.uint64)
        anc = np.empty(n, dtype=np.uint32) if ancestors else None
        self._chk(capi.lib().ssme_pf_download_state(self._h, f, capi.dptr(x), capi.dptr(lw), capi.u64ptr(cdf),
                                                    capi.u32ptr(anc)))
        nt = self.n_tiles
        m, s, rs = np.empty(1), np.zeros(1, dtype=np.uint64), C.c_int32()
        A, mb = np.empty(nt, dtype=np.uint64), np.empty(nt)
        self._chk(capi.lib().ssme_pf_download_scalars(self._h, f, capi.dptr(m), capi.u64ptr(s), capi.u64ptr(A),
                                                      capi.dptr(mb), C.byref(rs)))
        return dict(x=x, logw=lw, cdf=cdf, anc=anc, m=float(m[0]), S=int(s[0]), A=A, mb=mb, rshift=rs.value)

    def last_elapsed_ms(self):
        ms = C.c_float()
        self._chk(capi.lib().ssme_pf_last_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def profile_series(self, y, z=None):
        """Mean launch duration (us) of the step kernel, HIP events on the handle's stream."""
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        us = np.empty(1)
        cnt = np.zeros(1, dtype=np.int32)
        self._chk(capi.lib().ssme_pf_profile_series(self._h, capi.dptr(yv), capi.dptr(zv), yv.size, capi.dptr(us),
                                                    cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return {"filter_step_us": float(us[0]), "launches": int(cnt[0])}


class _SingleFilter:
    """One model object with the reference's call surface: filter(), getLogCondLike()."""
    _model = None

    def __init__(self, theta, nparts, seed=0, resampler=RESAMP_MULTINOMIAL, rs=1, device=0):
        self._bank = ParticleFilterBank(self._model, nparts, 1, seed, resampler, rs, device)
        self._bank.set_params(theta)
        self._last = 0.0
        self._fs = ()

    def filter(self, y, z=None, fs=()):
        self._fs = tuple(fs)
        self._last = float(self._bank.step(float(np.ravel(y)[0]), None if z is None else float(np.ravel(z)[0]))[0])

    def getLogCondLike(self):
        return self._last

    def getExpectations(self):
        """fs given to filter(): SSME_H_* ids run on the device in one pass; anything callable (the reference's
        std::function h, pswarm_filter.h:44) is evaluated on the host over the downloaded (x, weights)."""
        ids = [f for f in self._fs if not callable(f)]
        dev = iter(self._bank.expectations_multi(ids)[:, 0].tolist()) if ids else iter(())
        xw = None
        out = []
        for f in self._fs:
            if callable(f):
                if xw is None:
                    xw = self._bank.weights(0)
                hv = np.array([np.asarray(f(xi), dtype=np.float64) for xi in xw[0]])
                out.append(np.tensordot(xw[1], hv, axes=(0, 0)) / xw[1].sum())
            else:
                out.append(float(next(dev)))
        return out

    @property
    def bank(self):
        return self._bank


class svol_bs(_SingleFilter):
    """example/univ_svol_bootstrap_filter.h:17-103. Ctor order (phi, beta, sigma) as :34."""
    _model = MODEL_SVOL

    def __init__(self, phi, beta, sigma, nparts=500, **kw):
        super().__init__([beta, phi, sigma], nparts, **kw)

    @classmethod
    def from_pack(cls, untrans_params, nparts=500, **kw):
        """param::pack ctor, order beta, phi, ss with sigma = sqrt(ss) (:55-61)."""
        beta, phi, ss = (float(v) for v in untrans_params)
        return cls(phi, beta, float(np.sqrt(ss)), nparts, **kw)


class svol_leverage(_SingleFilter):
    """test/test_pswarm.cpp:32-141. Ctor order (phi, mu, sigma, rho)."""
    _model = MODEL_SVOL_LEVERAGE

    def __init__(self, phi, mu, sigma, rho, nparts=500, **kw):
        super().__init__([phi, mu, sigma, rho], nparts, **kw)


class lin_gauss_bs(_SingleFilter):
    """Linear-Gaussian anchor model (not in the reference): exact Kalman log-lik known."""
    _model = MODEL_LIN_GAUSS

    def __init__(self, phi, sigma, tau, nparts=500, **kw):
        super().__init__([phi, sigma, tau], nparts, **kw)


def log_like_eval(untrans_theta, data, nparts=500, num_pfilters=1, seed=0, resampler=RESAMP_MULTINOMIAL, device=0,
                  bank=None):
    """estimate_univ_svol.h:108-131 with thread_pool's replicate aggregation (thread_pool.h:263-268):
    num_pfilters independent filters at the same theta, log-mean-exp of their log-likelihoods."""
    data = capi.as_f64(data)
    if data.size == 0:
        raise ValueError("can't read in data")   # std::length_error in the reference (:112-113)
    beta, phi, ss = (float(v) for v in untrans_theta)
    own = bank is None
    if own:
        bank = ParticleFilterBank(MODEL_SVOL, nparts, num_pfilters, seed, resampler, 1, device)
    try:
        bank.set_params([beta, phi, float(np.sqrt(ss))])
        bank.run_series(data)
        return bank.log_mean_exp()
    finally:
        if own:
            bank.close()


# ---- particle swarm ---------------------------------------------------------------------------------------------
class SwarmWithCovs:
    """include/ssme/pswarm_filter.h:325-560: nparamparts bootstrap filters, one parameter draw each, advanced together.

    update(y, z) = filter(y, z, fs) on every member (comp_func :380-388), then the swarm's log conditional likelihood and
    expectations are the plain averages over members (intra/inter_agg_func :392-460: uniform weights, parameters come
    from the prior).  All members live in ONE handle (n_filters = nparamparts, one theta row each), so an update is one
    kernel launch.  Subclasses give samp_untrans_params() (the reference's pure virtual); fs are SSME_H_* ids.
    """
    _model = MODEL_SVOL_LEVERAGE

    def __init__(self, fs, nstateparts, nparamparts, seed=0, resampler=RESAMP_MULTINOMIAL, device=0, first_filter_id=0):
        self._fs = tuple(int(f) for f in fs)
        self._args = (int(nstateparts), int(nparamparts), seed, resampler, 1, device)
        self._first = first_filter_id
        self._bank = None
        self._lcl = 0.0
        self._exp = [0.0] * len(self._fs)
        self.num_obs = 0
        self.params = None

    def samp_untrans_params(self):
        raise NotImplementedError("subclasses sample one untransformed parameter vector, as the reference's pure virtual")

    def _finish_construction(self):                         # pswarm_filter.h:280-304
        n, r = self._args[0], self._args[1]
        self.params = np.array([self.samp_untrans_params() for _ in range(r)], dtype=np.float64)
        self._bank = ParticleFilterBank(self._model, n, r, *self._args[2:], first_filter_id=self._first)
        self._bank.set_params(self.params)

    def update(self, y, z=0.0):
        if self._bank is None:
            self._finish_construction()
        self._member_lcl = self._bank.step(float(np.ravel(y)[0]), float(np.ravel(z)[0]))
        self._lcl, self._exp = self._bank.swarm_aggregate(self._fs)      # means over members, reduced on the device
        self.num_obs += 1

    def getLogCondLike(self):
        return self._lcl

    def getExpectations(self):
        return list(self._exp)

    def close(self):
        if self._bank is not None:
            self._bank.close()
            self._bank = None


class Swarm(SwarmWithCovs):
    """include/ssme/pswarm_filter.h:23-320: the swarm without covariates, over univariate-SVOL members.
    samp_untrans_params() returns (beta, phi, sigma), the C ABI's order for MODEL_SVOL."""
    _model = MODEL_SVOL

    def update(self, y):
        if self._bank is None:
            self._finish_construction()
        self._member_lcl = self._bank.step(float(np.ravel(y)[0]), None)
        self._lcl, self._exp = self._bank.swarm_aggregate(self._fs)
        self.num_obs += 1


class svol_swarm_1(SwarmWithCovs):
    """test/test_pswarm.cpp:146-208: SVOL-leverage members, parameters (phi, mu, sigma, rho) drawn from uniform priors."""

    def __init__(self, fs, phi_l, phi_u, mu_l, mu_u, sig_l, sig_u, rho_l, rho_u, dte=0, nstateparts=500, nparamparts=100,
                 prior_seed=0, **kw):
        super().__init__(fs, nstateparts, nparamparts, **kw)
        self._lo = np.array([phi_l, mu_l, sig_l, rho_l], dtype=np.float64)
        self._hi = np.array([phi_u, mu_u, sig_u, rho_u], dtype=np.float64)
        self._rng = np.random.default_rng(prior_seed)

    def samp_untrans_params(self):
        return self._lo + (self._hi - self._lo) * self._rng.random(4)


# ---- Liu-West -------------------------------------------------------------------------------------------------
TR_NULL, TR_TWICE_FISHER, TR_LOGIT, TR_LOG = 0, 1, 2, 3        # include/ssme/parameters.h:27


class svol_lw_1_par:
    """test/test_liu_west.cpp:22-157: Liu-West filter (auxiliary form, with covariates) for the SVOL-leverage model.

    Ctor arguments as the reference's: delta, then the uniform prior bounds of phi, mu, sigma, rho.  Methods:
    filter(y, z), getLogCondLike(), getParamSamples() (transformed parameters, as param::pack::get_trans_params),
    plus run_series / param_means for whole-series use.  n_filters independent filters share one handle.
    """

    _form = 0            # auxiliary-particle form (LWFilterWithCovs); svol_lw_2_par overrides

    def __init__(self, delta, phi_l, phi_u, mu_l, mu_u, sig_l, sig_u, rho_l, rho_u, dte=0, nparts=10, n_filters=1, seed=0,
                 device=0, first_filter_id=0, transforms=(TR_LOGIT, TR_NULL, TR_LOG, TR_TWICE_FISHER), rs=1):
        """rs: the reference's resampling schedule m_rs (resample when (t + 1) % rs == 0; liu_west_filter.h:1139-1140)."""
        self._h = C.c_void_p()
        self.n, self.r = int(nparts), int(n_filters)
        cfg = capi.LwConfig(n_particles=nparts, n_filters=n_filters, seed=seed, device=device, first_filter_id=first_filter_id,
                            delta=delta, form=self._form, resamp_sched=rs)
        cfg.transforms[:] = list(transforms)
        cfg.prior_lo[:] = [phi_l, mu_l, sig_l, rho_l]
        cfg.prior_hi[:] = [phi_u, mu_u, sig_u, rho_u]
        capi.check(capi.lib().ssme_lw_create(C.byref(cfg), C.byref(self._h)))
        self._last = np.zeros(self.r)
        self._T = 0

    def close(self):
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

    def set_debug(self, on=True, split_level2=None):
        """Record k / ancestor indices; split_level2: True / False force the level-2 policy (None: split above 1024 tiles)."""
        pol = 0 if split_level2 is None else (4 if split_level2 else 8)
        self._chk(capi.lib().ssme_lw_set_debug(self._h, (1 if on else 0) | pol))

    def reset(self):
        self._chk(capi.lib().ssme_lw_reset(self._h))

    def filter(self, y, z=0.0, fs=()):
        """filter(obs, cov, fs) of LWFilterWithCovs / LWFilter2WithCovs (liu_west_filter.h:840, :2050); without z it is
        the no-covariate LWFilter / LWFilter2 call (:238, :1447; the same model with the covariate term at zero).
        fs: functionals for getExpectations(): ids 0-7 (see expectations()) run on the device; callables
        h(x, z, theta) -- the reference's std::function (state, covariate, UNTRANSFORMED parameters) -- on the host."""
        yv = np.array([float(np.ravel(y)[0])])
        zv = np.array([float(np.ravel(z)[0])])
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_lw_step(self._h, capi.dptr(yv), capi.dptr(zv), capi.dptr(out)))
        self._last = out
        self._fs, self._z = tuple(fs), zv[0]

    def getLogCondLike(self):
        return float(self._last[0]) if self.r == 1 else self._last.copy()

    def getExpectations(self):
        """E[h(x_t, z_t, theta) | y_{1:t}] for the fs given to the last filter() call (:1054-1075, :2267-2290), filter 0:
        a list with one entry per functional, shaped as h returns (scalars for the built-in ids)."""
        fs = getattr(self, "_fs", ())
        ids = [int(f) for f in fs if not callable(f)]
        dev = self.expectations(ids)[:, 0] if ids else []
        out, d, host = [], 0, None
        for f in fs:
            if not callable(f):
                out.append(float(dev[d]))
                d += 1
                continue
            if host is None:
                host = self.weights(0)
            x, th, w = host
            acc = None
            for i in range(self.n):
                hv = np.asarray(f(x[i], self._z, th[:, i]), dtype=np.float64) * w[i]
                acc = hv if acc is None else acc + hv
            out.append(acc / w.sum())
        return out

    def run_series(self, y, z=None):
        yv = capi.as_f64(y)
        zv = None if z is None else capi.as_f64(z)
        out = np.empty(self.r)
        self._chk(capi.lib().ssme_lw_run_series(self._h, capi.dptr(yv), capi.dptr(zv), yv.size, capi.dptr(out)))
        self._T = yv.size
        return out

    def per_step(self):
        out = np.empty((self.r, self._T))
        self._chk(capi.lib().ssme_lw_get_per_step(self._h, capi.dptr(out), self._T))
        return out

    def param_means(self):
        out = np.empty((self.r, 4))
        self._chk(capi.lib().ssme_lw_get_param_means(self._h, capi.dptr(out)))
        return out

    def expectations(self, functionals):
        """E[h | y_{1:t}] under the last step's weights, [n, R]: ids 0-3 = x, x^2, exp(x/2), 42 (the SSME_H_* ids);
        4-7 = untransformed phi, mu, sigma, rho (getExpectations(), liu_west_filter.h:1054-1075)."""
        fs = np.ascontiguousarray(functionals, dtype=np.int32)
        out = np.empty((fs.size, self.r))
        self._chk(capi.lib().ssme_lw_get_expectations(self._h, fs.ctypes.data_as(C.POINTER(C.c_int32)), fs.size, capi.dptr(out)))
        return out

    def weights(self, f=0):
        """(x, untransformed theta [4, N], w) of filter f after the last step, for host-side functionals h(x, z, theta)."""
        x, th, w = np.empty(self.n), np.empty((4, self.n)), np.empty(self.n)
        self._chk(capi.lib().ssme_lw_download_weights(self._h, f, capi.dptr(x), capi.dptr(th), capi.dptr(w)))
        return x, th, w

    def state(self, f=0, indices=False):
        n = self.n
        x, th = np.empty(n), np.empty((4, n))
        k = np.empty(n, dtype=np.uint32) if indices else None
        a = np.empty(n, dtype=np.uint32) if indices else None
        tb, L = np.empty(4), np.empty((4, 4))
        self._chk(capi.lib().ssme_lw_download_state(self._h, f, capi.dptr(x), capi.dptr(th), capi.u32ptr(k), capi.u32ptr(a),
                                                    capi.dptr(tb), capi.dptr(L)))
        return dict(x=x, theta=th, kidx=k, anc=a, thetabar=tb, L=L)

    def getParamSamples(self, f=0):
        return self.state(f)["theta"]

    def last_elapsed_ms(self):
        ms = C.c_float()
        self._chk(capi.lib().ssme_lw_last_elapsed_ms(self._h, C.byref(ms)))
        return ms.value


class svol_lw_2_par(svol_lw_1_par):
    """test/test_liu_west.cpp:214-358: the "alternative" Liu-West filter -- LWFilter2WithCovs, plain SISR form
    (liu_west_filter.h:2191-2343): jitter, qSamp (= the transition), weight += logFEv + logGEv - logQEv = logGEv.
    Same ctor arguments and methods as svol_lw_1_par."""
    _form = 1
