set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_t11.log 2>&1 || true
tail -8 gpurun_out/r2_t11.log
for n in 262144 1048576 2097152 16777216; do
  python tools/prof_run.py --lw --T 64 --passes 3 --n $n >> gpurun_out/r2_lw11.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r2_lw11.log | cut -c1-100
