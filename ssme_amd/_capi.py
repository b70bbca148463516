"""ctypes binding of the C ABI in include/ssme_pf.h (ssme_amd/libssme_pf.so).

The library is the product; there is no CPU fallback.  If the shared object is missing or a
HIP device is absent, calls raise (SsmeError / OSError) -- nothing routes through oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SSME_PF_LIB") or os.path.join(HERE, "libssme_pf.so")   # SSME_PF_LIB: measurement builds

# enums of include/ssme_pf.h
OK, ERR_INVALID_ARG, ERR_LENGTH, ERR_UNSUPPORTED, ERR_HIP, ERR_STATE = range(6)
MODEL_SVOL, MODEL_SVOL_LEVERAGE, MODEL_LIN_GAUSS, MODEL_USER0 = 0, 1, 2, 3
RESAMP_MULTINOMIAL, RESAMP_SYSTEMATIC, RESAMP_STRATIFIED, RESAMP_MULTINOMIAL_IID = 0, 1, 2, 3
F64, F32 = 0, 1
H_X, H_X2, H_VOL, H_CONST42 = 0, 1, 2, 3

EXPORTS = [
    "ssme_pf_create", "ssme_pf_destroy", "ssme_pf_default_tile", "ssme_pf_user_model_n_theta", "ssme_pf_user_model_dims", "ssme_pf_set_params", "ssme_pf_reset", "ssme_pf_set_seed", "ssme_pf_set_small_series", "ssme_pf_shard_create", "ssme_pf_set_stream",
    "ssme_pf_shard_prepare", "ssme_shard_comm_get_unique_id", "ssme_shard_comm_init", "ssme_shard_comm_destroy", "ssme_pf_shard_run_series",
    "ssme_pf_shard_download", "ssme_pf_shard_stats", "ssme_pf_shard_layout", "ssme_pf_shard_plan", "ssme_pf_shard_step", "ssme_pf_shard_finalize", "ssme_pf_step",
    "ssme_pf_run_series", "ssme_pf_get_per_step", "ssme_pf_get_loglik", "ssme_pf_get_expectations", "ssme_pf_get_expectations_multi", "ssme_pf_swarm_aggregate", "ssme_pf_swarm_aggregate_threads", "ssme_pf_download_weights", "ssme_pf_get_layout",
    "ssme_pf_log_mean_exp", "ssme_pf_download_state", "ssme_pf_download_scalars", "ssme_pf_set_debug",
    "ssme_pf_set_graph_mode", "ssme_pf_set_tuning", "ssme_pf_last_elapsed_ms", "ssme_pf_profile_series", "ssme_pf_test_math",
    "ssme_pf_test_philox", "ssme_pf_test_quantize", "ssme_pf_test_rescale", "ssme_pf_test_block_scan",
    "ssme_pf_test_copy", "ssme_pf_test_gamma",
    "ssme_pf_strerror", "ssme_pf_last_error",
    "ssme_pf_version",
    "ssme_lw_create", "ssme_lw_destroy", "ssme_lw_reset", "ssme_lw_step", "ssme_lw_run_series", "ssme_lw_get_per_step",
    "ssme_lw_get_param_means", "ssme_lw_get_expectations", "ssme_lw_download_weights", "ssme_lw_download_state", "ssme_lw_set_debug", "ssme_lw_last_elapsed_ms",
    "ssme_lw_last_error",
    "ssme_lw_shard_create", "ssme_lw_set_stream", "ssme_lw_shard_set_plane_tiles", "ssme_lw_shard_prepare", "ssme_lw_shard_init", "ssme_lw_shard_plan",
    "ssme_lw_shard_stage1", "ssme_lw_shard_mid", "ssme_lw_shard_stage2", "ssme_lw_shard_finalize", "ssme_lw_get_loglik", "ssme_lw_shard_run_series", "ssme_lw_shard_download", "ssme_lw_shard_stats", "ssme_lw_shard_layout",
]


class Config(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("n_particles", C.c_int32), ("n_filters", C.c_int32), ("dtype", C.c_int32),
        ("resampler", C.c_int32), ("resamp_sched", C.c_int32), ("seed", C.c_uint64), ("device", C.c_int32),
        ("first_filter_id", C.c_uint32), ("tile_particles", C.c_int32), ("n_filters_total", C.c_int32),
    ]


class LwConfig(C.Structure):
    _fields_ = [
        ("n_particles", C.c_int32), ("n_filters", C.c_int32), ("seed", C.c_uint64), ("device", C.c_int32),
        ("first_filter_id", C.c_uint32), ("delta", C.c_double), ("transforms", C.c_int32 * 4),
        ("prior_lo", C.c_double * 4), ("prior_hi", C.c_double * 4), ("form", C.c_int32), ("resamp_sched", C.c_int32),
    ]


class SsmeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"ssme_pf status {status}: {msg}")
        self.status = status


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("SSME_PF_LIB"):
            # never load a library built from other sources than the ones on disk (content hash, ssme_amd/build.py):
            # a stale .so silently tests / benches yesterday's kernels
            from . import build as _build
            if _build.needs_build():
                _build.build()
        if not os.path.exists(SO_PATH):
            raise OSError(f"{SO_PATH} not built: run `python -m ssme_amd.build` (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback")
        # PyTorch-ROCm wheels bundle their own libamdhip64; when this process also uses torch (bench.py, tests),
        # torch must load its HIP runtime first so that one runtime serves both (the loader then resolves our
        # libamdhip64.so.7 dependency to the copy already mapped).  C/C++ callers without torch use /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        dp, u32p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        u64p = C.POINTER(C.c_uint64)
        H = C.c_void_p
        L.ssme_pf_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
        L.ssme_pf_destroy.argtypes = [H]
        L.ssme_pf_set_params.argtypes = [H, dp, C.c_int32, C.c_int32]
        L.ssme_pf_reset.argtypes = [H]
        L.ssme_pf_step.argtypes = [H, dp, dp, dp]
        L.ssme_pf_run_series.argtypes = [H, dp, dp, C.c_int32, dp]
        L.ssme_pf_get_per_step.argtypes = [H, dp, C.c_int32]
        L.ssme_pf_get_loglik.argtypes = [H, dp]
        L.ssme_pf_get_expectations.argtypes = [H, C.c_int32, dp]
        L.ssme_pf_log_mean_exp.argtypes = [H, dp]
        L.ssme_pf_get_expectations_multi.argtypes = [H, i32p, C.c_int32, dp]
        L.ssme_pf_swarm_aggregate.argtypes = [H, i32p, C.c_int32, dp, dp]
        L.ssme_pf_swarm_aggregate_threads.argtypes = [H, i32p, C.c_int32, C.c_int32, dp, dp]
        L.ssme_pf_download_weights.argtypes = [H, C.c_int32, dp, dp]
        L.ssme_pf_get_layout.argtypes = [H, i32p, i32p]
        L.ssme_pf_default_tile.argtypes = [C.c_int32, C.c_int32]
        L.ssme_pf_user_model_n_theta.argtypes = []
        L.ssme_pf_user_model_dims.argtypes = [i32p, i32p]
        L.ssme_pf_download_state.argtypes = [H, C.c_int32, dp, dp, u64p, u32p]
        L.ssme_pf_download_scalars.argtypes = [H, C.c_int32, dp, u64p, u64p, dp, i32p]
        L.ssme_pf_set_debug.argtypes = [H, C.c_int32]
        L.ssme_pf_set_graph_mode.argtypes = [H, C.c_int32]
        L.ssme_pf_set_tuning.argtypes = [H, C.c_int32]
        L.ssme_pf_last_elapsed_ms.argtypes = [H, C.POINTER(C.c_float)]
        L.ssme_pf_profile_series.argtypes = [H, dp, dp, C.c_int32, dp, i32p]
        L.ssme_pf_test_math.argtypes = [C.c_int32, C.c_int32, dp, dp, C.c_int64]
        L.ssme_pf_test_philox.argtypes = [C.c_int32, u32p, u32p, u32p]
        L.ssme_pf_test_quantize.argtypes = [C.c_int32, dp, C.c_int32, u64p, C.c_int64]
        L.ssme_pf_test_block_scan.argtypes = [C.c_int32, C.c_int32, u64p, u64p, u64p]
        L.ssme_pf_test_rescale.argtypes = [C.c_int32, u64p, dp, C.c_int32, u64p, C.c_int64]
        L.ssme_pf_test_copy.argtypes = [C.c_int32, C.c_int64, C.c_int32]
        L.ssme_pf_test_gamma.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, C.c_int32, C.c_double, C.c_int32, dp]
        L.ssme_pf_shard_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.POINTER(H)]
        L.ssme_pf_set_stream.argtypes = [H, C.c_void_p]
        L.ssme_pf_shard_prepare.argtypes = [H, dp, dp, C.c_int32]
        L.ssme_pf_shard_plan.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.ssme_pf_shard_step.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ssme_pf_shard_finalize.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
        L.ssme_shard_comm_get_unique_id.argtypes = [C.c_void_p]
        L.ssme_shard_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.ssme_shard_comm_destroy.argtypes = [C.c_void_p]
        L.ssme_pf_shard_run_series.argtypes = [H, C.c_void_p, dp, dp, C.c_int32, C.c_int32, dp]
        L.ssme_pf_shard_download.argtypes = [H, dp, u64p, i32p, C.POINTER(C.c_int64)]
        L.ssme_pf_shard_stats.argtypes = [H, i32p]
        L.ssme_pf_shard_layout.argtypes = [H, i32p]
        L.ssme_pf_set_seed.argtypes = [H, C.c_uint64]
        L.ssme_pf_set_small_series.argtypes = [H, C.c_int32]
        L.ssme_lw_create.argtypes = [C.POINTER(LwConfig), C.POINTER(H)]
        L.ssme_lw_destroy.argtypes = [H]
        L.ssme_lw_reset.argtypes = [H]
        L.ssme_lw_step.argtypes = [H, dp, dp, dp]
        L.ssme_lw_run_series.argtypes = [H, dp, dp, C.c_int32, dp]
        L.ssme_lw_get_per_step.argtypes = [H, dp, C.c_int32]
        L.ssme_lw_get_param_means.argtypes = [H, dp]
        L.ssme_lw_get_expectations.argtypes = [H, i32p, C.c_int32, dp]
        L.ssme_lw_download_weights.argtypes = [H, C.c_int32, dp, dp, dp]
        L.ssme_lw_download_state.argtypes = [H, C.c_int32, dp, dp, u32p, u32p, dp, dp]
        L.ssme_lw_set_debug.argtypes = [H, C.c_int32]
        L.ssme_lw_last_elapsed_ms.argtypes = [H, C.POINTER(C.c_float)]
        vp = C.c_void_p
        L.ssme_lw_shard_create.argtypes = [C.POINTER(LwConfig), C.c_int32, C.c_int32, C.POINTER(H)]
        L.ssme_lw_set_stream.argtypes = [H, vp]
        L.ssme_lw_shard_set_plane_tiles.argtypes = [H, C.c_int32]
        L.ssme_lw_shard_prepare.argtypes = [H, dp, dp, C.c_int32]
        L.ssme_lw_shard_init.argtypes = [H, vp, vp, vp, vp, vp]
        L.ssme_lw_shard_plan.argtypes = [H, C.c_int32, C.c_int32, vp, vp, C.POINTER(C.c_int32)]
        L.ssme_lw_shard_stage1.argtypes = [H, C.c_int32, C.c_int32, C.c_int32] + [vp] * 13
        L.ssme_lw_shard_mid.argtypes = [H, C.c_int32, vp, vp, vp]
        L.ssme_lw_shard_stage2.argtypes = [H, C.c_int32, C.c_int32, C.c_int32] + [vp] * 12
        L.ssme_lw_shard_finalize.argtypes = [H, C.c_int32, vp, vp]
        L.ssme_lw_get_loglik.argtypes = [H, dp]
        L.ssme_lw_shard_run_series.argtypes = [H, vp, dp, dp, C.c_int32, dp]
        L.ssme_lw_shard_download.argtypes = [H, dp, dp, C.POINTER(C.c_int64)]
        L.ssme_lw_shard_stats.argtypes = [H, i32p]
        L.ssme_lw_shard_layout.argtypes = [H, i32p]
        L.ssme_lw_last_error.restype = C.c_char_p
        L.ssme_lw_last_error.argtypes = [H]
        L.ssme_pf_strerror.restype = C.c_char_p
        L.ssme_pf_strerror.argtypes = [C.c_int]
        L.ssme_pf_last_error.restype = C.c_char_p
        L.ssme_pf_last_error.argtypes = [H]
        L.ssme_pf_version.restype = C.c_int
        for name in EXPORTS:
            if name not in ("ssme_pf_strerror", "ssme_pf_last_error", "ssme_pf_version", "ssme_lw_last_error"):
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def u32ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint32))


def u64ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint64))


def check(status, handle=None):
    if status != OK:
        L = lib()
        msg = L.ssme_pf_strerror(status).decode()
        if handle is not None and status == ERR_HIP:
            msg += " (" + L.ssme_pf_last_error(handle).decode() + ")"
        raise SsmeError(status, msg)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
