cd $GRAFT_REPO_ROOT
for p in 0 1 2 3 4 5; do
  a=$(SSME_PRIO_MODE=$p python3 tools/prof_run.py --T 512 --passes 3 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  b=$(SSME_PRIO_MODE=$p python3 tools/prof_run.py --T 96 --passes 3 --n 8388608 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  c=$(SSME_PRIO_MODE=$p python3 tools/prof_run.py --T 256 --passes 3 --n 16384 --filters 512 --model 1 --tile 2048 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  echo "prio $p: N=2^20 $a | N=2^23 $b | 512x2^14 $c"
done
for s in 0 1; do
  a=$(SSME_STREAM_STORES=$s python3 tools/prof_run.py --T 512 --passes 3 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  b=$(SSME_STREAM_STORES=$s python3 tools/prof_run.py --T 96 --passes 3 --n 8388608 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  c=$(SSME_STREAM_STORES=$s python3 tools/prof_run.py --T 256 --passes 3 --n 16384 --filters 512 --model 1 --tile 2048 2>&1 | grep -o "us/step [0-9.]*" | head -1)
  echo "stream_stores $s: N=2^20 $a | N=2^23 $b | 512x2^14 $c"
done
