# in-kernel level-2 (--split 0) against the one-launch split level-2 (--split 1) around the crossover
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for tiles in 512 640 768 896 1024; do for sp in 0 1; do
  n=$((tiles * 2048))
  echo "tiles=$tiles split=$sp: $(python3 tools/prof_run.py --T 384 --passes 3 --n $n --split $sp 2>&1 | grep -o 'us/step [0-9.]*' | head -1)"
done; done; done
