set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_t7.log 2>&1 || true
tail -4 gpurun_out/r2_t7.log
for n in 4096 16384 65536 131072 262144 524288 1048576; do
  python tools/prof_run.py --T 512 --passes 3 --n $n >> gpurun_out/r2_p7.log 2>&1
done
python tools/prof_run.py --T 512 --passes 3 --n 65536 --filters 8 >> gpurun_out/r2_p7.log 2>&1
python tools/prof_run.py --T 512 --passes 3 --n 65536 --filters 8 --tile 512 >> gpurun_out/r2_p7.log 2>&1
python tools/prof_run.py --T 512 --passes 3 --n 65536 --filters 8 --tile 2048 >> gpurun_out/r2_p7.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_p7.log
g++ -O2 -std=c++17 -Iinclude examples/estimate_univ_svol_gpu.cpp -Lssme_amd -l:libssme_pf.so -Wl,-rpath,$PWD/ssme_amd -o examples/estimate_univ_svol_gpu
mkdir -p gpurun_out/pm
E=examples/estimate_univ_svol_gpu; D=tests/golden/spy_returns.csv; O=gpurun_out/pm
TIMEFORMAT='wall %R s'
{
for cfg in "200 100 500" "200 1 500" "200 1 100" "20 8 65536"; do set -- $cfg; echo "iters=$1 filters=$2 N=$3"; time $E $D $O/S_$1_$2_$3 $O/M_$1_$2_$3 $1 $2 $3 20261004 2>&1 | tail -3; done
echo "iters=10000 filters=1 N=65536"; time $E $D $O/S10k $O/M10k 10000 1 65536 20261004 2>&1 | tail -4
} > gpurun_out/r02_pmmh.txt 2>&1
ls $O >> gpurun_out/r02_pmmh.txt
cat gpurun_out/r02_pmmh.txt
