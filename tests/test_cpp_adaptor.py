"""The header-only C++ adaptor (include/ssme_gpu/bsfilter_gpu.hpp): compiles with g++ against the C ABI on CPU;
on the GPU it reproduces the oracle bit for bit when driven like the reference's callers."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adaptor")


def _build():
    from ssme_amd import build
    so = build.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_adaptor.cpp"),
                           "-o", EXE, so, "-Wl,-rpath," + os.path.dirname(so)])
    return EXE


def test_adaptor_compiles_and_links():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_adaptor_matches_oracle(oracle, spy):
    exe = _build()                       # always from the sources on disk (a stale binary has the old config layout)
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "spy_returns.csv")], text=True)
    vals = dict(line.split(" ", 1) for line in out.strip().splitlines())
    th = [1.0, 0.95, 0.25]
    y = spy[:64]
    of = oracle.Filter(oracle.MODEL_SVOL, 500, th, 77)
    assert float(vals["svol_bs"]) == of.run_series(y)[0]
    lls = [oracle.Filter(oracle.MODEL_SVOL, 500, th, 77, rep=r).run_series(y)[0] for r in range(4)]
    assert abs(float(vals["log_like_eval_gpu"]) - oracle.log_mean_exp(np.array(lls))) < 1e-12
    ol = oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, 1000, [0.9, 0.0, 1.0, -0.1], 77, rep=3)
    z = np.concatenate([[0.0], y[:-1]])
    assert float(vals["svol_leverage"]) == sum(ol.step(y[t], z[t]) for t in range(8))
    assert abs(float(vals["expect42"]) - 42.0) < 1e-4
    # probed functionals (device) and the host path over the downloaded (x, weights) agree with the oracle / each other
    assert abs(float(vals["expectx"]) - ol.expectation(0)) <= 1e-12 * abs(ol.expectation(0))
    assert abs(float(vals["host_x"]) - float(vals["expectx"])) <= 1e-12 * abs(float(vals["expectx"]))
    assert abs(float(vals["host_x2"]) - float(vals["dev_x2"])) <= 1e-12 * abs(float(vals["dev_x2"]))
    assert abs(float(vals["host_z"]) - z[7]) <= 1e-12 * abs(z[7]) and abs(float(vals["host_sin"])) <= 1.0
    so = ol.state()
    wo = np.exp(so["logw"] - so["logw"].max())
    assert abs(float(vals["host_sin"]) - (np.sin(so["x"]) * wo).sum() / wo.sum()) < 1e-9
    assert vals["length_error"].strip() == "ok"
    assert abs(float(vals["evaluator"]) - oracle.log_mean_exp(np.array(lls))) < 1e-12
    lw = oracle.LWFilter(800, 77)
    assert float(vals["liu_west"]) == sum(lw.step(y[t], z[t]) for t in range(6))
    lw2 = oracle.LWFilter(800, 77, form=1)
    assert float(vals["liu_west2"]) == sum(lw2.step(y[t], z[t]) for t in range(6))
    assert abs(float(vals["liu_west2_42"]) - 42.0) < 1e-9
    # (6c) std::function functionals of the Liu-West filters: device-recognised and host-summed ones, the no-covariate call
    lwf = oracle.LWFilter(800, 77)
    for t in range(6):
        ll_t = lwf.step(y[t], z[t])
    assert float(vals["lwf_ll"]) == ll_t
    assert abs(float(vals["lwf_42"]) - 42.0) < 1e-9
    assert abs(float(vals["lwf_phi"]) - lwf.expectation(4)) <= 1e-10 * abs(lwf.expectation(4))
    assert abs(float(vals["lwf_host_x"]) - float(vals["lwf_dev_x"])) <= 1e-10 * abs(float(vals["lwf_dev_x"]))
    assert abs(float(vals["lwf_host_rho"]) - float(vals["lwf_dev_rho"])) <= 1e-10 * abs(float(vals["lwf_dev_rho"]))
    assert abs(float(vals["lwf_host_z"]) - z[5]) <= 1e-12 * abs(z[5])
    st = lwf.state()
    wts = np.exp(st["logw"] - st["logw"].max())
    sig = np.exp(st["theta"][2])                          # sigma = exp(theta_2): transforms logit, null, log, twice_fisher
    want = (st["x"] * sig * wts).sum() / wts.sum()
    assert abs(float(vals["lwf_host_xsig"]) - want) <= 1e-9 * abs(want)
    lwn = oracle.LWFilter(800, 77, form=1)
    assert float(vals["lwn_ll"]) == sum(lwn.step(y[t], 0.0) for t in range(4))
    assert abs(float(vals["lwn_x2"]) - lwn.expectation(1)) <= 1e-10 * abs(lwn.expectation(1))
    # swarm: 5 members, theta rows as test_swarm::samp_untrans_params, plain averages over members
    mem = []
    for k in range(5):
        u = 0.1 + 0.2 * k
        th = [0.8 + 0.19 * u, -0.1 + 0.2 * u, 0.01 + 0.09 * u, -0.5 + 0.49 * u]
        mem.append(oracle.Filter(oracle.MODEL_SVOL_LEVERAGE, 600, th, 77, rep=k))
    tot, ex = 0.0, 0.0
    for t in range(5):
        lls = [m.step(y[t], z[t]) for m in mem]
        tot += sum(lls) / 5
        ex = sum(m.expectation(0) for m in mem) / 5
    assert abs(float(vals["swarm"]) - tot) < 1e-12
    assert abs(float(vals["swarm42"]) - 42.0) < 1e-4
    assert abs(float(vals["swarmx"]) - ex) < 1e-9
    mem = []
    for k in range(3):
        u = 0.2 + 0.3 * k
        mem.append(oracle.Filter(oracle.MODEL_SVOL, 400, [0.8 + 0.4 * u, 0.9 + 0.05 * u, 0.2 + 0.1 * u], 77, rep=k))   # beta, phi, sigma
    tot = sum(sum(m.step(y[t]) for m in mem) / 3 for t in range(4))
    assert abs(float(vals["swarm_nocov"]) - tot) < 1e-12
    # (10) the unmodified swarm templates (structural stand-in): one handle per member, filter ids 0..4, same members as (8)
    assert abs(float(vals["uswarm"]) - float(vals["swarm"])) < 1e-12
    assert abs(float(vals["uswarm42"]) - 42.0) < 1e-4
    assert abs(float(vals["uswarmx"]) - float(vals["swarmx"])) < 1e-9
    mus = [-0.1 + 0.2 * (0.1 + 0.2 * k) for k in range(5)]
    assert abs(float(vals["uswarmmux"]) - (float(vals["swarmx"]) + sum(mus) / 5)) < 1e-9        # E[mu_i + x] per member
    assert abs(float(vals["uswarm_nocov"]) - float(vals["swarm_nocov"])) < 1e-12
    # (10b) the same swarms with their members in ONE handle behind the unmodified templates (swarm_context): one launch per
    # update, the numbers of the one-handle-per-member path -- log-likelihoods and device functionals to the bit, the
    # host-summed functional (uses the member's parameters) likewise (same particles, same weights, same host loop)
    for k in ("", "42", "x", "mux"):
        assert float(vals["cswarm" + k]) == float(vals["uswarm" + k]), k
    assert float(vals["cswarm_nocov"]) == float(vals["uswarm_nocov"]) and float(vals["cswarm_nocov_x"]) == float(vals["uswarm_nocov_x"])
    assert vals["late_member"].strip() == "rejected" and vals["mismatch"].strip() == "rejected"
    # (10c) members driven from three concurrent threads, as split_data_thread_pool does
    assert abs(float(vals["tswarm"]) - float(vals["uswarm"])) < 1e-12 and abs(float(vals["tswarmx"]) - float(vals["uswarmx"])) < 1e-12
    # functionals are summed on the host unless DECLARED (or the opt-in probe classifies them): the default must be right
    # for functions the fixed probe points cannot tell from a built-in
    assert abs(float(vals["hostdefault_42"]) - 42.0) < 1e-9
    assert abs(float(vals["hostdefault_x"]) - float(vals["expectx"])) <= 1e-12 * abs(float(vals["expectx"]))
    assert abs(float(vals["probe_x"]) - float(vals["expectx"])) <= 1e-12 * abs(float(vals["expectx"]))
    tail = ((so["x"] < -3.0) * wo).sum() / wo.sum()
    assert tail > 0 and abs(float(vals["tail_host"]) - tail) < 1e-12
    assert float(vals["tail_probe"]) == 0.0               # what the probe makes of it: "constant 0" -- hence opt-in only
    n, first = vals["read_data"].split()
    assert int(n) == spy.size and float(first) == spy[0]


def test_pmmh_harness_compiles():
    from ssme_amd import build
    so = build.build()
    exe = os.path.join(ROOT, "examples", "estimate_univ_svol_gpu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "estimate_univ_svol_gpu.cpp"), "-o", exe, so,
                           "-Wl,-rpath," + os.path.dirname(so)])
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_pmmh_harness_runs(tmp_path):
    """The shipped example's CLI on the device: 30 iterations, 4 replicate filters of 2000 particles."""
    exe = os.path.join(ROOT, "examples", "estimate_univ_svol_gpu")
    test_pmmh_harness_compiles()         # always from the sources on disk
    import json
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "spy_returns.csv"), str(tmp_path / "samples"),
                        str(tmp_path / "messages"), "30", "4", "2000", "7"], capture_output=True, text=True, check=True)
    info = json.loads(p.stderr.strip().splitlines()[-1])
    assert info["iters"] == 30 and info["T"] == 3084
    samples = [f for f in os.listdir(tmp_path) if f.startswith("samples_")]
    messages = [f for f in os.listdir(tmp_path) if f.startswith("messages_")]
    assert len(samples) == 1 and len(messages) == 1
    rows = np.loadtxt(tmp_path / samples[0], delimiter=",")
    assert rows.shape == (30, 3) and np.all(np.isfinite(rows)) and np.all((rows[:, 1] > -1) & (rows[:, 1] < 1)) and np.all(rows[:, 2] > 0)
    lines = open(tmp_path / messages[0]).read().splitlines()
    assert lines[0].startswith("iter number, accept rate, old_ll") and len(lines) == 31
    assert np.isfinite(float(lines[1].split(",")[2]))


def _build_user():
    from ssme_amd import build
    so = build.build_user_model(os.path.join(ROOT, "tests", "models", "svol_two_factor.h"), "two_factor")
    exe = os.path.join(ROOT, "tests", "cpp", "test_user_adaptor")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_user_adaptor.cpp"),
                           "-o", exe, so, "-Wl,-rpath," + os.path.dirname(so)])
    return exe


def test_user_model_adaptor_compiles_and_links():
    assert os.path.exists(_build_user())


@pytest.mark.gpu
def test_user_model_adaptor_matches_oracle(oracle, spy):
    """user_bs_gpu<nparts, 2, 2>: the caller-visible surface of a BSFilter<nparts, dimx, dimy, ...> model whose callbacks are the header
    compiled into the linked library (tests/models/svol_two_factor.h) -- filter(y), getLogCondLike(), expectations of functions of the
    whole state -- against the oracle's restatement of the same model."""
    from test_parity_gpu import _two_factor_oracle
    exe = _build_user()
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "spy_returns.csv")], text=True)
    vals = dict(line.split(" ", 1) for line in out.strip().splitlines())
    T = 8
    y = np.stack([spy[:T], spy[100:100 + T]], axis=1)
    of = _two_factor_oracle(oracle, 3000, 21, 1, 0, None, 1)
    ll, per = of.run_series(y)
    assert float(vals["user_vec_ll"]) == float(np.add.reduce(per)) or abs(float(vals["user_vec_ll"]) - ll) < 1e-9
    assert float(vals["user_vec_last"]) == per[-1]
    st = of.state()
    w = np.exp(st["logw"] - st["logw"].max())
    assert abs(float(vals["user_vec_sum"]) - ((st["x"][0] + st["x"][1]) * w).sum() / w.sum()) < 1e-9
    assert abs(float(vals["user_vec_prod"]) - ((st["x"][0] * st["x"][1]) * w).sum() / w.sum()) < 1e-9
    assert abs(float(vals["user_vec_42"]) - 42.0) < 1e-9
    assert vals["dims_check"].strip() == "ok"
