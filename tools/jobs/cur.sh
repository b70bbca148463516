set -e
mkdir -p gpurun_out
for cfg in "--n 50" "--n 100" "--n 250" "--n 500" "--n 1000" "--n 500 --filters 100" "--n 100 --filters 100" "--n 500 --filters 100 --model 1" "--n 500 --resampler 1"; do
  for sp in 1 0; do
  echo "$cfg pair=$sp" >> gpurun_out/r2_small20.log
  SSME_SMALL_PAIR=$sp python tools/prof_run.py --T 3084 --passes 3 $cfg >> gpurun_out/r2_small20.log 2>&1
  done
done
grep -v amdgpu.ids gpurun_out/r2_small20.log | cut -c1-120
