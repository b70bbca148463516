# A/B of the split level-2 of unsharded bootstrap filters: --split 1 = one launch + ranges in the step kernel, --split 2 = table kernels
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "level2 or 2048_tiles or source_ranges" 2>&1 | tail -3
for rep in 1 2; do for n in 4194304 8388608 16777216 33554432; do for sp in 1 2; do
  echo "N=$n split=$sp: $(python3 tools/prof_run.py --T 64 --passes 3 --n $n --split $sp 2>&1 | grep -o 'us/step [0-9.]*' | head -1)"
done; done; done
echo "headline: $(python3 tools/prof_run.py --T 512 --passes 3 2>&1 | grep -o 'us/step [0-9.]*' | head -1)"
