# BASELINE.json configs[2] end to end: 10 000 ada-PMMH iterations at N = 2^16 through examples/estimate_univ_svol_gpu (progress every ~1000 iterations)
cd "${GRAFT_REPO_ROOT:-.}"
g++ -O2 -std=c++17 -Iinclude examples/estimate_univ_svol_gpu.cpp ssme_amd/libssme_pf.so -Wl,-rpath,$PWD/ssme_amd -o examples/estimate_univ_svol_gpu
mkdir -p gpurun_out/pmmh
( while sleep 45; do echo "  ... $(wc -l < gpurun_out/pmmh/M 2>/dev/null) iterations"; done ) &
HB=$!
timeout -k 10 900 ./examples/estimate_univ_svol_gpu tests/golden/spy_returns.csv gpurun_out/pmmh/S gpurun_out/pmmh/M 10000 1 65536 20261004
kill $HB
echo "# messages file, last 3 lines:"; tail -3 gpurun_out/pmmh/M
python3 - <<PY
import numpy as np
s = np.loadtxt("gpurun_out/pmmh/S", delimiter=",")
s = s[2000:]
print("posterior mean (beta, phi, ss) after 2000 burn-in:", np.round(s.mean(0), 5).tolist(), " sd:", np.round(s.std(0), 5).tolist())
PY
for cfg in "200 100 500" "200 1 500" "200 1 100" "20 8 65536"; do set -- $cfg; echo "iters=$1 filters=$2 N=$3"; ./examples/estimate_univ_svol_gpu tests/golden/spy_returns.csv gpurun_out/pmmh/S2 gpurun_out/pmmh/M2 $1 $2 $3 20261004 | tail -1; done
