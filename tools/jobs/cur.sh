set -e
mkdir -p gpurun_out
python -m pytest tests/test_stat_anchor_gpu.py -x -q -m gpu > gpurun_out/r2_t16.log 2>&1 || true
tail -12 gpurun_out/r2_t16.log
