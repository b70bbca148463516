#!/bin/bash
# Round-end measurement on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r01
# 1. bench.py (default flags)                         -> profiles/<tag>_bench.json            (copied from gpurun_out)
# 2. rocprofv3 --kernel-trace --stats of the same cmd -> profiles/<tag>_bench_kernel_stats.csv
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (own runs)    -> profiles/traffic_latest.json
set -eo pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/traffic.py collect $OUT/traffic
python3 tools/traffic.py summarize $OUT/traffic $OUT/traffic_latest.json
cp $OUT/traffic_latest.json profiles/traffic_latest.json
python3 bench.py > $OUT/${TAG}_bench.json
cat $OUT/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cat $OUT/${TAG}_bench_kernel_stats.csv
