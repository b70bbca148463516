"""CPU suite: the C-ABI library builds, loads and exports every symbol include/ssme_pf.h declares;
host-side argument validation; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ssme_pf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssme_(?:pf|lw|shard)_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from ssme_amd import build, _capi
    so = build.build()
    assert os.path.exists(so)
    L = C.CDLL(so)
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ssme_pf.h but not exported"
    assert sorted(_capi.EXPORTS) == names
    assert _capi.lib().ssme_pf_version() >= 100


def test_strerror_and_null_handling():
    from ssme_amd import _capi
    L = _capi.lib()
    assert L.ssme_pf_strerror(0) == b"ok"
    assert b"invalid" in L.ssme_pf_strerror(_capi.ERR_INVALID_ARG)
    assert L.ssme_pf_create(None, None) == _capi.ERR_INVALID_ARG
    assert L.ssme_pf_destroy(None) == _capi.ERR_INVALID_ARG
    assert L.ssme_pf_reset(None) == _capi.ERR_INVALID_ARG


@pytest.mark.parametrize("field,value,status", [
    ("n_particles", 0, 1), ("n_filters", 0, 1), ("model", 9, 1), ("model", 3, 3), ("resampler", 7, 1), ("resamp_sched", 0, 1),
    ("dtype", 2, 1), ("n_particles", (1 << 25) + 1, 3),
])
def test_create_validates_config(field, value, status):
    from ssme_amd import _capi
    cfg = _capi.Config(model=0, n_particles=100, n_filters=1, dtype=0, resampler=0, resamp_sched=1, seed=1,
                       device=0, first_filter_id=0)
    setattr(cfg, field, value)
    h = C.c_void_p()
    assert _capi.lib().ssme_pf_create(C.byref(cfg), C.byref(h)) == status
    assert not h.value


def test_user_model_extension_point_builds():
    """ssme_amd/csrc/model_api.h: a library with a user model compiled in (tests/models/svol_student_t.h) builds with the same
    resource checks as the stock one (no scratch, no spills), exports the whole C ABI and reports the model's theta length;
    the stock library reports none and refuses the model id (no GPU needed for either)."""
    from ssme_amd import build, _capi
    so = build.build_user_model(os.path.join(ROOT, "tests", "models", "svol_student_t.h"), "student_t")
    L = C.CDLL(so)
    for n in _header_functions():
        assert hasattr(L, n)
    assert L.ssme_pf_user_model_n_theta() == 4
    assert _capi.lib().ssme_pf_user_model_n_theta() == 0
    dx, dy = C.c_int32(-1), C.c_int32(-1)
    assert L.ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)) == _capi.OK and (dx.value, dy.value) == (1, 1)
    assert _capi.lib().ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)) == _capi.ERR_UNSUPPORTED and (dx.value, dy.value) == (0, 0)
    # a model with a vector state and observation (tests/models/svol_two_factor.h: dim_x = dim_y = 2)
    so2 = build.build_user_model(os.path.join(ROOT, "tests", "models", "svol_two_factor.h"), "two_factor")
    L2 = C.CDLL(so2)
    for n in _header_functions():
        assert hasattr(L2, n)
    assert L2.ssme_pf_user_model_n_theta() == 6
    assert L2.ssme_pf_user_model_dims(C.byref(dx), C.byref(dy)) == _capi.OK and (dx.value, dy.value) == (2, 2)
    cfgv = _capi.Config(model=_capi.MODEL_USER0, n_particles=4 * 2048, n_filters=1, dtype=0, resampler=0, resamp_sched=1, seed=1, device=0)
    hv = C.c_void_p()
    assert L2.ssme_pf_shard_create(C.byref(cfgv), 0, 2, C.byref(hv)) == _capi.ERR_UNSUPPORTED       # vector models do not shard
    cfgv.dtype = _capi.F32
    assert L2.ssme_pf_create(C.byref(cfgv), C.byref(hv)) == _capi.ERR_UNSUPPORTED                   # ... and are fp64 at the boundary
    cfg = _capi.Config(model=_capi.MODEL_USER0, n_particles=100, n_filters=1, dtype=0, resampler=0, resamp_sched=1, seed=1, device=0)
    h = C.c_void_p()
    assert _capi.lib().ssme_pf_create(C.byref(cfg), C.byref(h)) == _capi.ERR_UNSUPPORTED


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ssme_amd
    with pytest.raises(ssme_amd.SsmeError) as e:
        ssme_amd.ParticleFilterBank(ssme_amd.MODEL_SVOL, 128)
    assert e.value.status == 4
    with pytest.raises(ssme_amd.SsmeError):
        ssme_amd.log_like_eval([1.0, 0.5, 2e-4], np.ones(4), nparts=64)


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = "import sys, ssme_amd, ssme_amd._capi; assert not any(m.startswith('oracle') for m in sys.modules), sys.modules.keys()"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for fn in ("_capi.py", "filters.py", "__init__.py", "build.py", "parallel.py", "sharded.py", "csrc/pf_api.hip",
               "csrc/pf_kernels.h", "csrc/pf_small.h", "csrc/lw_kernels.h", "csrc/ssme_math.h"):
        text = open(os.path.join(ROOT, "ssme_amd", fn)).read()
        assert "import oracle" not in text and "from oracle" not in text and "libssme_oracle" not in text


def test_shard_and_liu_west_entry_points_validate_arguments():
    """Argument checks happen before any HIP call, so they are testable without a device."""
    from ssme_amd import _capi
    L = _capi.lib()
    h = C.c_void_p()
    cfg = _capi.Config(model=0, n_particles=4 * 2048, n_filters=1, dtype=0, resampler=0, resamp_sched=1, seed=1, device=0,
                       first_filter_id=0)
    assert L.ssme_pf_shard_create(C.byref(cfg), 0, 0, C.byref(h)) == _capi.ERR_INVALID_ARG        # world < 1
    assert L.ssme_pf_shard_create(C.byref(cfg), 2, 2, C.byref(h)) == _capi.ERR_INVALID_ARG        # rank >= world
    assert L.ssme_pf_shard_create(C.byref(cfg), 0, 3, C.byref(h)) == _capi.ERR_UNSUPPORTED        # 4 tiles over 3 ranks
    cfg.n_filters = 2
    assert L.ssme_pf_shard_create(C.byref(cfg), 0, 2, C.byref(h)) == _capi.ERR_UNSUPPORTED        # one filter only
    cfg.n_filters, cfg.resamp_sched = 1, 0
    assert L.ssme_pf_shard_create(C.byref(cfg), 0, 2, C.byref(h)) == _capi.ERR_UNSUPPORTED        # resamp_sched < 1
    # accepted shapes get past the argument checks (and fail at the first HIP call where there is no device):
    # a resampling schedule; N that is not whole tiles per rank (5 tiles + 1 particle over 2 ranks: 3 + 3, the last one ragged)
    for n, sched in ((4 * 2048, 2), (5 * 2048 + 1, 1)):
        cfg.n_particles, cfg.resamp_sched = n, sched
        rc = L.ssme_pf_shard_create(C.byref(cfg), 1, 2, C.byref(h))
        assert rc in (_capi.OK, _capi.ERR_HIP)
        if rc == _capi.OK:
            lay = (C.c_int32 * 4)()
            assert L.ssme_pf_shard_layout(h, lay) == _capi.OK
            assert list(lay) == ([4, 2, 2, 4096] if n == 4 * 2048 else [6, 3, 3, 2 * 2048 + 1])
            L.ssme_pf_destroy(h)
            h = C.c_void_p()
    cfg.n_particles = 4 * 2048
    cfg.resamp_sched, cfg.dtype = 1, _capi.F32
    assert L.ssme_pf_shard_create(C.byref(cfg), 0, 2, C.byref(h)) == _capi.ERR_UNSUPPORTED        # float-at-the-boundary handles do not shard
    assert not h.value
    for fn in (L.ssme_pf_set_stream, ):
        assert fn(None, None) == _capi.ERR_INVALID_ARG
    assert L.ssme_pf_set_seed(None, 1) == _capi.ERR_INVALID_ARG
    assert L.ssme_pf_set_small_series(None, 1) == _capi.ERR_INVALID_ARG
    lw = _capi.LwConfig(n_particles=100, n_filters=1, seed=1, device=0, first_filter_id=0, delta=0.99)
    lw.transforms[:] = [2, 0, 3, 1]
    lw.prior_lo[:] = [0.8, -0.1, 0.01, -0.5]
    lw.prior_hi[:] = [0.99, 0.1, 0.1, -0.01]
    for field, value in (("delta", 0.0), ("delta", 1.5), ("n_particles", 0), ("n_filters", 0)):
        bad = _capi.LwConfig.from_buffer_copy(lw)
        setattr(bad, field, value)
        assert L.ssme_lw_create(C.byref(bad), C.byref(h)) == _capi.ERR_INVALID_ARG
    bad = _capi.LwConfig.from_buffer_copy(lw)
    bad.transforms[1] = 9
    assert L.ssme_lw_create(C.byref(bad), C.byref(h)) == _capi.ERR_INVALID_ARG
    bad = _capi.LwConfig.from_buffer_copy(lw)
    bad.prior_lo[0], bad.prior_hi[0] = 0.9, 0.8                                                  # empty prior interval
    assert L.ssme_lw_create(C.byref(bad), C.byref(h)) == _capi.ERR_INVALID_ARG
    assert L.ssme_lw_destroy(None) == _capi.ERR_INVALID_ARG
    assert L.ssme_lw_shard_create(C.byref(lw), 0, 0, C.byref(h)) == _capi.ERR_INVALID_ARG      # world < 1
    assert L.ssme_lw_shard_create(C.byref(lw), 0, 2, C.byref(h)) == _capi.ERR_UNSUPPORTED      # 100 particles: not whole tiles per rank
    assert L.ssme_lw_set_stream(None, None) == _capi.ERR_INVALID_ARG
    assert not h.value


def test_sharded_python_class_requires_process_group():
    from ssme_amd.sharded import ShardedParticleFilter
    with pytest.raises(AssertionError):
        ShardedParticleFilter(0, 4096)


def test_mock_rccl_and_thread_harness_compile():
    """The test-only RCCL stand-in (ranks = host threads on one GPU) and the multi-rank harness build against the C ABI."""
    import subprocess
    from ssme_amd import build
    so = build.build()
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    mock = os.path.join(cpp, "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wno-unused-result", os.path.join(cpp, "mock_rccl.cpp"), "-o", mock])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(cpp, "test_shard_threads.cpp"), "-o", os.path.join(cpp, "test_shard_threads"),
                           "-Wl,--no-as-needed", mock, "-Wl,--as-needed", so, "-Wl,-rpath," + cpp, "-Wl,-rpath," + os.path.dirname(so)])
