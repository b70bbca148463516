set -e
mkdir -p gpurun_out
for n in 1048576 2097152; do
  for ss in 0 1 0 1; do
    echo "LW n=$n stream=$ss" >> gpurun_out/r2_ss_lw2.log
    SSME_LW_STREAM_STORES=$ss python tools/prof_run.py --lw --T 64 --passes 3 --n $n >> gpurun_out/r2_ss_lw2.log 2>&1
  done
done
grep -v amdgpu.ids gpurun_out/r2_ss_lw2.log | cut -c1-90
