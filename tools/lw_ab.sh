# Liu-West timing at two sizes (best of 3 passes, three repetitions) after the LW parity tests
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "liu or lw" 2>&1 | tail -3
for rep in 1 2 3; do
  echo "LW N=2^20 $(python3 tools/prof_run.py --lw --T 64 --passes 3 2>&1 | grep -o 'us/step [0-9.]*') | N=2^22 $(python3 tools/prof_run.py --lw --T 32 --passes 3 --n 4194304 2>&1 | grep -o 'us/step [0-9.]*')"
done
