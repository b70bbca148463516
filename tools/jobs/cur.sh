set -e
mkdir -p gpurun_out
for sd in 1 2 3 4; do
python tests/soak_parity.py 3000 800 $sd > gpurun_out/r2_soak_$sd.log 2>&1
tail -2 gpurun_out/r2_soak_$sd.log
done
