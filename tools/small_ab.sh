# one-tile whole-series kernels: us per step at the shipped example's shapes
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do for cfg in "100 1" "500 1" "500 100" "64 1" "256 1" "1000 1" "2048 1"; do set -- $cfg
  echo "N=$1 R=$2: $(python3 tools/prof_run.py --T 3084 --passes 3 --n $1 --filters $2 2>&1 | grep -o 'us/step [0-9.]*')"
done; done
