"""ctypes wrapper over oracle/libssme_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (ssme_amd) never does.  See oracle/ssme_oracle.cpp for the reference
file:line each function follows and for the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libssme_oracle.so")
_SO_O3 = os.path.join(_HERE, "libssme_oracle_o3.so")          # -O3 build, timed by bench.py's cpu_baseline only

MODEL_SVOL, MODEL_SVOL_LEVERAGE, MODEL_LIN_GAUSS = 0, 1, 2
RESAMP_MULTINOMIAL, RESAMP_SYSTEMATIC, RESAMP_STRATIFIED, RESAMP_MULTINOMIAL_IID = 0, 1, 2, 3
TILE = 2048


def build(force=False):
    src = os.path.join(_HERE, "ssme_oracle.cpp")
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO_O3)):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_REF_SO = os.path.join(_HERE, "_ref", "libref_thread_pool.so")


def build_ref(reference="/root/reference"):
    """oracle/_ref: the reference's own thread_pool.h behind a driver (only where /root/reference exists; the GPU box
    uses the prebuilt file).  Returns the path or None."""
    hdr = os.path.join(reference, "include", "ssme", "thread_pool.h")
    if os.path.exists(hdr):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref", "REF=" + reference], stdout=subprocess.DEVNULL)
    return _REF_SO if os.path.exists(_REF_SO) else None


def ref_log_mean_exp(v):
    """log-mean-exp computed BY THE REFERENCE (thread_pool<>::work); None if oracle/_ref is not built."""
    so = build_ref()
    if so is None:
        return None
    L = C.CDLL(so)
    L.ref_thread_pool_log_mean_exp.restype = C.c_double
    L.ref_thread_pool_log_mean_exp.argtypes = [C.POINTER(C.c_double), C.c_int]
    v = np.ascontiguousarray(v, dtype=np.float64)
    return float(L.ref_thread_pool_log_mean_exp(v.ctypes.data_as(C.POINTER(C.c_double)), v.size))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, u32p = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
        L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
        for f in (L.orc_exp, L.orc_log, L.orc_log_u, L.orc_log_u32, L.orc_exp_t):
            f.argtypes = [dp, dp, C.c_long]
        L.orc_sincos2pi.argtypes = [dp, dp, dp, C.c_long]
        L.orc_sincos_k24.argtypes = [dp, dp, dp, C.c_long]
        L.orc_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, dp]
        u64p = C.POINTER(C.c_uint64)
        L.orc_exp_scaled.argtypes = [dp, C.c_int, dp, C.c_long]
        L.orc_gamma.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_int, dp]
        L.orc_quantize.argtypes = [dp, C.c_int, u64p, C.c_long]
        L.orc_pf_sum_int.restype = C.c_uint64
        L.orc_pf_sum_int.argtypes = [C.c_void_p]
        L.orc_pf_create.restype = C.c_void_p
        L.orc_pf_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, dp, C.c_int]
        L.orc_pf_destroy.argtypes = [C.c_void_p]
        L.orc_pf_reset.argtypes = [C.c_void_p]
        L.orc_pf_step.restype = C.c_double
        L.orc_pf_step.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_pf_loglik.restype = C.c_double
        L.orc_pf_loglik.argtypes = [C.c_void_p]
        L.orc_pf_run_series.restype = C.c_double
        L.orc_pf_run_series.argtypes = [C.c_void_p, dp, dp, C.c_int, dp]
        L.orc_pf_state.argtypes = [C.c_void_p, dp, dp, u64p, u32p, u64p, dp, dp]
        L.orc_rescale.argtypes = [u64p, dp, C.c_int, u64p, C.c_long]
        L.orc_pf_expectation.restype = C.c_double
        L.orc_pf_expectation.argtypes = [C.c_void_p, C.c_int]
        L.orc_ref_run_series.restype = C.c_double
        L.orc_ref_run_series.argtypes = [C.c_int, dp, C.c_int, dp, dp, C.c_int, C.c_uint32, C.c_int, C.c_int, dp]
        L.orc_log_mean_exp.restype = C.c_double
        L.orc_log_mean_exp.argtypes = [dp, C.c_int]
        L.orc_inv_transform.restype = C.c_double
        L.orc_inv_transform.argtypes = [C.c_int, C.c_double]
        L.orc_log_jacobian.restype = C.c_double
        L.orc_log_jacobian.argtypes = [C.c_int, C.c_double]
        L.orc_kalman_loglik.restype = C.c_double
        L.orc_kalman_loglik.argtypes = [C.c_double, C.c_double, C.c_double, dp, C.c_int, dp]
        i32p = C.POINTER(C.c_int32)
        L.orc_lw_create.restype = C.c_void_p
        L.orc_lw_create.argtypes = [C.c_int, C.c_uint64, C.c_uint32, i32p, dp, dp, C.c_double, C.c_int, C.c_int]
        L.orc_lw_expectation.restype = C.c_double
        L.orc_lw_expectation.argtypes = [C.c_void_p, C.c_int]
        L.orc_lw_destroy.argtypes = [C.c_void_p]
        L.orc_lw_step.restype = C.c_double
        L.orc_lw_step.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_lw_loglik.restype = C.c_double
        L.orc_lw_loglik.argtypes = [C.c_void_p]
        L.orc_lw_param_means.argtypes = [C.c_void_p, dp]
        L.orc_lw_state.argtypes = [C.c_void_p, dp, dp, dp, u32p, u32p, dp, dp]
        L.orc_lw_ref_run.restype = C.c_double
        L.orc_lw_ref_run.argtypes = [C.c_int, i32p, dp, dp, C.c_double, dp, dp, C.c_int, C.c_uint32, dp, dp, C.c_int, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_u32p(c), _u32p(k), _u32p(o))
    return o


def _map1(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    fn(_dp(x), _dp(y), x.size)
    return y


def exp(x):
    return _map1(lib().orc_exp, x)


def log(x):
    return _map1(lib().orc_log, x)


def exp_t(x):
    """The bootstrap filter's exp (256-entry table + degree-5 series)."""
    return _map1(lib().orc_exp_t, x)


def log_u(x):
    """The table-based logarithm of the bootstrap filter's uniform draws (absolute error < 2^-51)."""
    return _map1(lib().orc_log_u, x)


def log_u32(x):
    """The logarithm behind the exponential spacings (32-bit uniforms, result quantised to 2^-35: absolute error < 2^-43)."""
    return _map1(lib().orc_log_u32, x)


def sincos_k24(k):
    """sin, cos of 2 pi k / 2^24 for integer k in [0, 2^24): the table-driven Box-Muller angle of the hot loops."""
    k = np.ascontiguousarray(k, dtype=np.float64)
    s, c = np.empty_like(k), np.empty_like(k)
    lib().orc_sincos_k24(_dp(k), _dp(s), _dp(c), k.size)
    return s, c


def sincos2pi(u):
    u = np.ascontiguousarray(u, dtype=np.float64)
    s, c = np.empty_like(u), np.empty_like(u)
    lib().orc_sincos2pi(_dp(u), _dp(s), _dp(c), u.size)
    return s, c


def normals(seed, rep, t, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_normals(seed, rep, t, n, _dp(out))
    return out


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64)) if a is not None else None


def exp_scaled(x, sc):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().orc_exp_scaled(_dp(x), int(sc), _dp(y), x.size)
    return y


def gamma_draws(seed, rep, t, shape, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_gamma(seed, rep, t, float(shape), n, _dp(out))
    return out


def rescale(A, dm, sc):
    A = np.ascontiguousarray(A, dtype=np.uint64)
    dm = np.ascontiguousarray(dm, dtype=np.float64)
    out = np.empty(A.size, dtype=np.uint64)
    lib().orc_rescale(_u64p(A), _dp(dm), int(sc), _u64p(out), A.size)
    return out


def quantize(x, sc):
    x = np.ascontiguousarray(x, dtype=np.float64)
    q = np.empty(x.size, dtype=np.uint64)
    lib().orc_quantize(_dp(x), int(sc), _u64p(q), x.size)
    return q


def default_tile(n, n_filters=1):
    """The device's rule when the caller does not choose a tile size (pf_api.hip: default_tile): the smallest of
    512 / 1024 / 2048 particles per tile that leaves at most 256 / 512 workgroups; 2048 for N <= 2048 and beyond."""
    if n <= 2048:
        return 2048
    if n_filters * ((n + 511) // 512) <= 256:
        return 512
    if n_filters * ((n + 1023) // 1024) <= 512:
        return 1024
    return 2048


class UserModelFilter:
    """Kernel-matched oracle filter (mode B) whose MODEL is given by Python callbacks -- the mirror of the device's model
    extension point (ssme_amd/csrc/model_api.h, SSME_MODEL_USER0): prop(x, zn, zcov) -> x', logg(y, x) -> log g, init_sd = the
    standard deviation of the t = 0 draw.  The callbacks restate the user model's operation sequence with THIS module's
    libm-free functions (exp_t, log, ...), independently of the device header."""

    def __init__(self, n, seed, init_sd, prop, logg, rep=0, resampler=RESAMP_MULTINOMIAL, resamp_sched=1, tile=None, bad=False):
        self.n = int(n)
        self.tile = default_tile(self.n) if tile is None else int(tile)
        self.nt = (self.n + self.tile - 1) // self.tile
        L = lib()
        self._prop = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double)(prop)      # kept alive with the object
        self._logg = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double)(logg)
        L.orc_pf_create_user.restype = C.c_void_p
        L.orc_pf_create_user.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int]
        self._h = L.orc_pf_create_user(self.n, resampler, resamp_sched, seed, rep, float(init_sd), int(bool(bad)),
                                       C.cast(self._prop, C.c_void_p), C.cast(self._logg, C.c_void_p), self.tile)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pf_destroy(self._h)
            self._h = None

    step = None      # filled in below from Filter (same handle type)


class UserVectorModelFilter:
    """The same for a user model with a VECTOR state / observation (model_api.h: dim_x, dim_y <= 4).  Callbacks on numpy views:
    init(zn[dx]) -> x0[dx];  prop(x[dx], zn[dx], zcov) -> x'[dx];  logg(y[dy], x[dx]) -> log g.  Component 0 of the normals is the
    pair's draw of the scalar filter, component d >= 1 one more Philox call per pair on counter stream 96 + d."""

    def __init__(self, n, seed, dx, dy, init, prop, logg, rep=0, resampler=RESAMP_MULTINOMIAL, resamp_sched=1, tile=None, bad=False):
        self.n, self.dx, self.dy = int(n), int(dx), int(dy)
        self.tile = default_tile(self.n) if tile is None else int(tile)
        self.nt = (self.n + self.tile - 1) // self.tile
        L = lib()
        dp = C.POINTER(C.c_double)
        arr = lambda p, k: np.ctypeslib.as_array(p, shape=(k,))

        def _init(zn, x0):
            arr(x0, self.dx)[:] = init(arr(zn, self.dx))

        def _prop(x, zn, zcov, xn):
            arr(xn, self.dx)[:] = prop(arr(x, self.dx), arr(zn, self.dx), zcov)

        def _logg(y, x):
            return float(logg(arr(y, self.dy), arr(x, self.dx)))

        self._cb = (C.CFUNCTYPE(None, dp, dp)(_init), C.CFUNCTYPE(None, dp, dp, C.c_double, dp)(_prop), C.CFUNCTYPE(C.c_double, dp, dp)(_logg))
        L.orc_pf_create_user_vec.restype = C.c_void_p
        L.orc_pf_create_user_vec.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_pf_run_series_vec.restype = C.c_double
        L.orc_pf_run_series_vec.argtypes = [C.c_void_p, dp, dp, C.c_int, dp]
        L.orc_pf_state_dim.argtypes = [C.c_void_p, C.c_int, dp]
        self._h = L.orc_pf_create_user_vec(self.n, resampler, resamp_sched, seed, rep, self.dx, self.dy, int(bool(bad)),
                                           C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), C.cast(self._cb[2], C.c_void_p),
                                           self.tile)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pf_destroy(self._h)
            self._h = None

    def run_series(self, y, z=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        z = None if z is None else np.ascontiguousarray(z, dtype=np.float64)
        T = y.size // self.dy
        per = np.empty(T)
        ll = lib().orc_pf_run_series_vec(self._h, _dp(y), _dp(z), T, _dp(per))
        return ll, per

    def state(self):
        """x: [dx, n] (component 0 first), cdf, ancestors, ... as Filter.state()."""
        st = Filter.state(self)
        x = np.empty((self.dx, self.n))
        x[0] = st["x"]
        for d in range(1, self.dx):
            lib().orc_pf_state_dim(self._h, d, _dp(x[d]))
        st["x"] = x
        return st


class Filter:
    """Kernel-matched oracle filter (mode B), one replicate.  tile: particles per tile (None: the device's default by N)."""

    def __init__(self, model, n, theta, seed, rep=0, resampler=RESAMP_MULTINOMIAL, resamp_sched=1, tile=None):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        self.n = int(n)
        self.tile = default_tile(self.n) if tile is None else int(tile)
        self.nt = (self.n + self.tile - 1) // self.tile
        self._h = lib().orc_pf_create(model, n, resampler, resamp_sched, seed, rep, _dp(th), self.tile)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pf_destroy(self._h)
            self._h = None

    def reset(self):
        lib().orc_pf_reset(self._h)

    def step(self, y, z=0.0):
        return lib().orc_pf_step(self._h, float(y), float(z))

    @property
    def loglik(self):
        return lib().orc_pf_loglik(self._h)

    def run_series(self, y, z=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        z = None if z is None else np.ascontiguousarray(z, dtype=np.float64)
        per = np.empty(y.size)
        ll = lib().orc_pf_run_series(self._h, _dp(y), _dp(z), y.size, _dp(per))
        return ll, per

    def state(self):
        n = self.n
        x, lw = np.empty(n), np.empty(n)
        loc = np.empty(n, dtype=np.uint64)
        anc = np.empty(n, dtype=np.uint32)
        A, mb, sc = np.empty(self.nt, dtype=np.uint64), np.empty(self.nt), np.empty(3)
        lib().orc_pf_state(self._h, _dp(x), _dp(lw), _u64p(loc), _u32p(anc), _u64p(A), _dp(mb), _dp(sc))
        return dict(x=x, logw=lw, cdf=loc, anc=anc, A=A, mb=mb, m=sc[0], S=int(lib().orc_pf_sum_int(self._h)),
                    rshift=int(sc[2]))

    def expectation(self, kind):
        return lib().orc_pf_expectation(self._h, kind)


_lib_o3 = None


def _baseline_lib():
    global _lib_o3
    if _lib_o3 is None:
        build()
        L = C.CDLL(_SO_O3)
        dp = C.POINTER(C.c_double)
        L.orc_ref_run_series.restype = C.c_double
        L.orc_ref_run_series.argtypes = [C.c_int, dp, C.c_int, dp, dp, C.c_int, C.c_uint32, C.c_int, C.c_int, dp]
        _lib_o3 = L
    return _lib_o3


def ref_run_series(model, theta, n, y, z=None, seed=1, use_float=False, fast_resampler=False, o3=False):
    """Mode A: mt19937 / <random> reference-faithful filter. Returns (loglik, per-step).
    o3: run the -O3 build (bench.py's cpu_baseline; same source, the reference example's optimisation level)."""
    th = np.ascontiguousarray(theta, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = None if z is None else np.ascontiguousarray(z, dtype=np.float64)
    per = np.empty(y.size)
    L = _baseline_lib() if o3 else lib()
    ll = L.orc_ref_run_series(model, _dp(th), n, _dp(y), _dp(z), y.size, seed, int(use_float),
                              int(fast_resampler), _dp(per))
    return ll, per


def log_mean_exp(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    return lib().orc_log_mean_exp(_dp(v), v.size)


def inv_transform(kind, tp):
    return lib().orc_inv_transform(kind, tp)


def log_jacobian(kind, tp):
    return lib().orc_log_jacobian(kind, tp)


def kalman_loglik(phi, sigma, tau, y):
    y = np.ascontiguousarray(y, dtype=np.float64)
    per = np.empty(y.size)
    return lib().orc_kalman_loglik(phi, sigma, tau, _dp(y), y.size, _dp(per)), per


# ---- Liu-West (include/ssme/liu_west_filter.h:971-1159; model test/test_liu_west.cpp:82-157) ----
TR_NULL, TR_TWICE_FISHER, TR_LOGIT, TR_LOG = 0, 1, 2, 3
LW_TRANSFORMS = (TR_LOGIT, TR_NULL, TR_LOG, TR_TWICE_FISHER)          # phi, mu, sigma, rho (test_liu_west.cpp:70)
LW_PRIOR_LO = (0.8, -0.1, 0.01, -0.5)                                  # test_liu_west.cpp:165
LW_PRIOR_HI = (0.99, 0.1, 0.1, -0.01)


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class LWFilter:
    """Kernel-matched Liu-West oracle with covariates, one filter.  form 0: auxiliary form (LWFilterWithCovs), 1: SISR form
    (LWFilter2WithCovs); resamp_sched = m_rs."""

    def __init__(self, n, seed, rep=0, delta=0.99, transforms=LW_TRANSFORMS, lo=LW_PRIOR_LO, hi=LW_PRIOR_HI, form=0, resamp_sched=1):
        self.n = int(n)
        tr = np.ascontiguousarray(transforms, dtype=np.int32)
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self._h = lib().orc_lw_create(n, seed, rep, _i32p(tr), _dp(lo), _dp(hi), float(delta), int(form), int(resamp_sched))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_lw_destroy(self._h)
            self._h = None

    def step(self, y, z=0.0):
        return lib().orc_lw_step(self._h, float(y), float(z))

    @property
    def loglik(self):
        return lib().orc_lw_loglik(self._h)

    def param_means(self):
        out = np.empty(4)
        lib().orc_lw_param_means(self._h, _dp(out))
        return out

    def expectation(self, functional):
        """ids 0-3: x, x^2, exp(x/2), 42 of the state; 4-7: untransformed phi, mu, sigma, rho."""
        return lib().orc_lw_expectation(self._h, int(functional))

    def state(self):
        n = self.n
        x, th, lw = np.empty(n), np.empty((4, n)), np.empty(n)
        k, a = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
        tb, L = np.empty(4), np.empty((4, 4))
        lib().orc_lw_state(self._h, _dp(x), _dp(th), _dp(lw), _u32p(k), _u32p(a), _dp(tb), _dp(L))
        return dict(x=x, theta=th, logw=lw, kidx=k, anc=a, thetabar=tb, L=L)


def lw_ref_run(n, y, z, seed=1, delta=0.99, transforms=LW_TRANSFORMS, lo=LW_PRIOR_LO, hi=LW_PRIOR_HI, form=0, resamp_sched=1):
    """Mode A: reference-faithful Liu-West (mt19937), auxiliary (form 0) or SISR (form 1) form, resampling schedule m_rs.
    Returns (loglik, per-step, posterior means of phi, mu, sigma, rho)."""
    tr = np.ascontiguousarray(transforms, dtype=np.int32)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    per, means = np.empty(y.size), np.empty(4)
    ll = lib().orc_lw_ref_run(n, _i32p(tr), _dp(lo), _dp(hi), float(delta), _dp(y), _dp(z), y.size, seed, _dp(per), _dp(means),
                              int(form), int(resamp_sched))
    return ll, per, means


for _name in ("reset", "step", "loglik", "run_series", "state", "expectation"):
    if hasattr(Filter, _name):
        setattr(UserModelFilter, _name, getattr(Filter, _name))
