#!/bin/bash
# Round-end measurement on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r02
# 1. bench.py (default flags, in-run PMC traffic)     -> profiles/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same cmd -> profiles/<tag>_bench_kernel_stats.csv
# 3. SQ instruction counters of the step kernel       -> profiles/<tag>_valu_counters.txt
# 4. Liu-West: kernel stats + FETCH/WRITE PMC         -> profiles/<tag>_liu_west_*.{csv,txt}
# 5. throughput by N, config 4 slice, sharded world 1 -> profiles/<tag>_*.txt
set -eo pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cat $OUT/${TAG}_bench.json | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof.err
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
head -6 $OUT/${TAG}_bench_kernel_stats.csv
# SQ counters (own passes, no tracing)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -- python3 tools/prof_run.py --T 24 --passes 1 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 tools/prof_run.py --T 24 --passes 1 > $OUT/sq2.log 2>&1
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 > $OUT/${TAG}_valu_counters.txt
cat $OUT/${TAG}_valu_counters.txt | head -40
# Liu-West
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lwprof -- python3 tools/prof_run.py --lw --T 64 --passes 2 > $OUT/lw.log 2>&1
cp $(find $OUT/lwprof -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_liu_west_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/lwf -- python3 tools/prof_run.py --lw --T 16 --passes 1 > $OUT/lwf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/lww -- python3 tools/prof_run.py --lw --T 16 --passes 1 > $OUT/lww.log 2>&1
python3 tools/pmc_summary.py $OUT/lwf $OUT/lww > $OUT/${TAG}_liu_west_traffic.txt
cat $OUT/${TAG}_liu_west_traffic.txt
# throughput by N, other configurations
{
for n in 4096 65536 262144 1048576 2097152 4194304 8388608 16777216 33554432; do python3 tools/prof_run.py --T 128 --passes 3 --n $n 2>&1 | grep "nt=" | sed "s/^/N=$n /"; done
} > $OUT/${TAG}_throughput_vs_N.txt
cat $OUT/${TAG}_throughput_vs_N.txt
python3 tools/prof_run.py --T 3084 --passes 2 --n 16384 --filters 512 --model 1 2>&1 | grep "nt=" > $OUT/${TAG}_config4_slice.txt
python3 tools/prof_run.py --T 3084 --passes 2 --n 65536 --filters 8 2>&1 | grep "nt=" > $OUT/${TAG}_config3_R8.txt
python3 tools/prof_run.py --T 3084 --passes 2 --n 65536 --filters 1 2>&1 | grep "nt=" >> $OUT/${TAG}_config3_R8.txt
cat $OUT/${TAG}_config4_slice.txt $OUT/${TAG}_config3_R8.txt
{
for n in 1048576 2097152 16777216; do python3 tools/prof_run.py --lw --T 32 --passes 2 --n $n 2>&1 | grep "liu-west"; done
} > $OUT/${TAG}_liu_west_vs_N.txt
cat $OUT/${TAG}_liu_west_vs_N.txt
# kernel statistics of the other shapes: N = 2^23 (one level-2 launch per step), config 4's per-GPU share, Liu-West at config 5's N
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p23 -- python3 tools/prof_run.py --T 96 --passes 2 --n 8388608 > $OUT/p23.log 2>&1
cp $(find $OUT/p23 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_n2p23_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pc4 -- python3 tools/prof_run.py --T 256 --passes 2 --n 16384 --filters 512 --model 1 > $OUT/pc4.log 2>&1
cp $(find $OUT/pc4 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_config4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plw -- python3 tools/prof_run.py --lw --T 24 --passes 2 --n 16777216 > $OUT/plw.log 2>&1
cp $(find $OUT/plw -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_liu_west_2p24_kernel_stats.csv
head -4 $OUT/${TAG}_n2p23_kernel_stats.csv | cut -c1-150; head -3 $OUT/${TAG}_config4_kernel_stats.csv | cut -c1-150; head -8 $OUT/${TAG}_liu_west_2p24_kernel_stats.csv | cut -c1-150
python3 bench.py --mode sharded --steps 2 --no-cpu-baseline > $OUT/${TAG}_bench_sharded_world1_native.json 2> $OUT/sh1.err
python3 bench.py --mode sharded --driver python --steps 2 --no-cpu-baseline > $OUT/${TAG}_bench_sharded_world1_python.json 2> $OUT/sh2.err
python3 bench.py --mode sharded --lw --particles 2097152 --steps 2 --no-cpu-baseline > $OUT/${TAG}_bench_sharded_world1_lw_native.json 2> $OUT/sh3.err
cut -c1-300 $OUT/${TAG}_bench_sharded_world1_native.json; cut -c1-300 $OUT/${TAG}_bench_sharded_world1_python.json
g++ -std=c++17 -O2 -Iinclude tools/step_latency.cpp -o tools/step_latency -Lssme_amd -l:libssme_pf.so -Wl,-rpath,$PWD/ssme_amd
./tools/step_latency tests/golden/spy_returns.csv > $OUT/${TAG}_step_api_latency.txt
cat $OUT/${TAG}_step_api_latency.txt
