set -e
mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_cpp_adaptor.py tests/test_stat_anchor_gpu.py -x -q -m gpu > gpurun_out/r2_t21.log 2>&1 || true
tail -6 gpurun_out/r2_t21.log
for cfg in "--n 1000" "--n 2000" "--n 2000 --filters 64"; do
  echo "$cfg" >> gpurun_out/r2_small21.log
  python tools/prof_run.py --T 3084 --passes 3 $cfg >> gpurun_out/r2_small21.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r2_small21.log | cut -c1-120
