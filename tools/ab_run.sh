#!/bin/bash
# Times the variant libraries of build/ab/ on ONE box: for each lib_NAME.so the headline shape (N = 2^20), N = 2^23,
# the 512 x 2^14 leverage bank and N = 2^16, us per step (best of 3 passes).   bash tools/ab_run.sh [NAME ...]
cd "${GRAFT_REPO_ROOT:-.}"
names="$@"; [ -z "$names" ] && names=$(ls build/ab/lib_*.so | sed 's/.*lib_\(.*\)\.so/\1/')
for rep in 1 2; do
for n in $names; do
  export SSME_PF_LIB=$PWD/build/ab/lib_$n.so
  a=$(python3 tools/prof_run.py --T 512 --passes 3 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  b=$(python3 tools/prof_run.py --T 96 --passes 3 --n 8388608 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  c=$(python3 tools/prof_run.py --T 256 --passes 3 --n 16384 --filters 512 --model 1 --tile 2048 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  d=$(python3 tools/prof_run.py --T 512 --passes 3 --n 65536 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  echo "$n: N=2^20 $a | N=2^23 $b | 512x2^14 lev $c | N=2^16 $d"
done
done
