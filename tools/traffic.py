#!/usr/bin/env python3
"""HBM traffic of k_filter_step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes).

Usage (on the GPU box, from the repo root):
    python3 tools/traffic.py collect gpurun_out/traffic      # runs the two PMC passes (+ calibration)
    python3 tools/traffic.py summarize gpurun_out/traffic profiles/traffic_latest.json

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of a wide (16 B/lane) coalesced streaming read; other shapes are uncalibrated, so each pass also runs a streaming copy
of known size with the step kernel's access shape (k_calib_copy) and the measured bytes-per-counter factors are applied.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAL_N = 1 << 24          # doubles per calibration copy = 128 MiB read + 128 MiB written
CAL_REP = 4


def driver():
    sys.path.insert(0, ROOT)
    import numpy as np
    import ssme_amd
    from ssme_amd import _capi
    _capi.check(_capi.lib().ssme_pf_test_copy(0, CAL_N, CAL_REP))
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:64]
    bank = ssme_amd.ParticleFilterBank(0, 1 << 20, 1, 20260101, 0)
    bank.set_graph_mode(False)
    bank.set_params([1.0, 0.95, 0.25])
    bank.run_series(y)
    bank.close()


def collect(outdir, env=None, cwd=None, quiet=False):
    """Two PMC passes, one counter each (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass).
    The program after `--` is the python interpreter itself (no env/bash hop under the profiler)."""
    os.makedirs(outdir, exist_ok=True)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(outdir, ctr), "--",
               sys.executable, os.path.abspath(__file__), "driver"]
        if not quiet:
            print(" ".join(cmd), flush=True)
        with open(os.path.join(outdir, ctr + ".log"), "w") as log:
            subprocess.check_call(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd=cwd, timeout=600)


def mean_counter(outdir, ctr, kernel_substr, skip_first=0):
    vals = []
    for f in glob.glob(os.path.join(outdir, ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    vals = vals[skip_first:]
    return sum(vals) / len(vals), len(vals)


def summarize(outdir, outjson, quiet=False):
    res = {}
    cal_bytes = CAL_N * 8.0
    f_cal, _ = mean_counter(outdir, "FETCH_SIZE", "k_calib_copy")
    w_cal, _ = mean_counter(outdir, "WRITE_SIZE", "k_calib_copy")
    f_fac = cal_bytes / (f_cal * 1024.0)       # bytes per reported KiB-byte of FETCH_SIZE
    w_fac = cal_bytes / (w_cal * 1024.0)
    f_k, n = mean_counter(outdir, "FETCH_SIZE", "k_filter_step", skip_first=1)   # t = 0 reads nothing
    w_k, _ = mean_counter(outdir, "WRITE_SIZE", "k_filter_step", skip_first=1)
    res["calibration"] = {"copy_bytes_each_way": cal_bytes, "FETCH_SIZE_KiB": f_cal, "WRITE_SIZE_KiB": w_cal,
                          "fetch_factor": f_fac, "write_factor": w_fac,
                          "note": "factor = true bytes / (counter * 1024) on a 16-B-per-lane streaming copy"}
    res["k_filter_step"] = {"launches": n, "FETCH_SIZE_KiB": f_k, "WRITE_SIZE_KiB": w_k,
                            "read_bytes": f_k * 1024 * f_fac, "write_bytes": w_k * 1024 * w_fac}
    res["filter_step_bytes_per_launch"] = res["k_filter_step"]["read_bytes"] + res["k_filter_step"]["write_bytes"]
    res["algorithmic_bytes_per_launch"] = 32.0 * (1 << 20)
    if outjson:
        json.dump(res, open(outjson, "w"), indent=1)
    if not quiet:
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    if sys.argv[1] == "driver":
        driver()
    elif sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        summarize(sys.argv[2], sys.argv[3])
