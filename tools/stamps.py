#!/usr/bin/env python3
"""Diagnostic (measurement build only): per-phase timing of k_filter_step from s_memrealtime stamps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SSME_PF_LIB"] = os.path.join(ROOT, "ssme_amd", "libssme_pf_ablate.so")
os.environ["SSME_STAMPS"] = "1"
import ssme_amd
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "spy_returns.csv"))[:40]
bank = ssme_amd.ParticleFilterBank(0, 1 << 20, 1, 20260101, rs)
bank.set_tuning(nt); bank.set_graph_mode(False)
bank.set_params([1.0, 0.95, 0.25])
bank.run_series(y)
os.environ["SSME_STAMPS_DUMP"] = "/tmp/stamps.txt"
bank.step(0.1)          # step_args() dumps the stamps of the previous launch (last step of the series)
s = np.loadtxt("/tmp/stamps.txt")
names = {0: "start", 1: "l2 loads issued", 2: "level-2 scan+range", 3: "barrier+stage issue", 4: "E-gen+scan",
         5: "targets+LDS fill+barrier", 6: "search+gather issue", 7: "Box-Muller", 8: "prop+logG", 9: "block max", 12: "exp+scan",
         10: "cdf store", 11: "end"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10, 11]
t0 = s[:, 0].min()
print(f"blocks {s.shape[0]}  kernel span {(s[:, 11].max() - t0) / 100:.2f} us   start spread {(s[:, 0].max() - t0) / 100:.2f} us")
prev = 0
for k in order[1:]:
    d = (s[:, k] - s[:, prev]) / 100.0
    print(f"  {names[k]:28s} mean {d.mean():6.2f} us  max {d.max():6.2f}   (ends at mean {((s[:, k] - t0) / 100).mean():6.2f} us)")
    prev = k

dur = (s[:, 11] - s[:, 0]) / 100.0
print("block duration percentiles (us): ", {p: round(float(np.percentile(dur, p)), 2) for p in (0, 10, 50, 90, 99, 100)})
endt = (s[:, 11] - t0) / 100.0
print("block end-time percentiles (us): ", {p: round(float(np.percentile(endt, p)), 2) for p in (0, 10, 50, 90, 99, 100)})
# slowest blocks: which phases are slow?
idx = np.argsort(dur)[-8:]
prev = 0
print("level-2 detail (us): loads landed at", round(((s[:,13]-s[:,0])/100).mean(),2), " scan done +", round(((s[:,14]-s[:,13])/100).mean(),2), " LDS/ballots +", round(((s[:,15]-s[:,14])/100).mean(),2), " finalize+rest +", round(((s[:,2]-s[:,15])/100).mean(),2))
print("slowest 8 blocks, per-phase us:")
for k in order[1:]:
    d = (s[idx, k] - s[idx, prev]) / 100.0
    print(f"  {names[k]:28s}", np.round(d, 2))
    prev = k
print("  block ids", idx)
# first-dispatched (id < 256: one per CU) vs second-dispatched workgroups of each CU
half = s.shape[0] // 2
for nm, sl in (("first half ", slice(0, half)), ("second half", slice(half, None))):
    d = (s[sl, 11] - s[sl, 0]) / 100.0
    st = (s[sl, 0] - t0) / 100.0
    en = (s[sl, 11] - t0) / 100.0
    print(f"{nm}: start {st.mean():.2f}  duration {d.mean():.2f} (min {d.min():.2f} max {d.max():.2f})  end {en.mean():.2f}")
    prev = 0
    row = []
    for k in order[1:]:
        row.append(round(float(((s[sl, k] - s[sl, prev]) / 100.0).mean()), 2)); prev = k
    print("   phases:", row)
