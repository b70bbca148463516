# threads per 2048-particle tile (256 / 512 / 1024) and tile sizes at three shapes
cd "${GRAFT_REPO_ROOT:-.}"
for n in 1048576 8388608; do
  echo "N=$n tile 2048, threads sweep:"; python3 tools/prof_run.py --T 256 --passes 3 --n $n --tile 2048 --sweep 2>&1 | grep -o "nt.*us/step [0-9.]*\|threads.*us/step [0-9.]*\|us/step [0-9.]*"
done
for t in 512 1024 2048; do
  echo "N=2^20 tile $t: $(python3 tools/prof_run.py --T 256 --passes 3 --tile $t 2>&1 | grep -o 'us/step [0-9.]*')"
  echo "N=2^16 tile $t: $(python3 tools/prof_run.py --T 512 --passes 3 --n 65536 --tile $t 2>&1 | grep -o 'us/step [0-9.]*')"
  echo "N=2^18 tile $t: $(python3 tools/prof_run.py --T 512 --passes 3 --n 262144 --tile $t 2>&1 | grep -o 'us/step [0-9.]*')"
done
