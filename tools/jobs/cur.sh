set -e
mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_cpp_adaptor.py -x -q -m gpu -k "functionals or adaptor" > gpurun_out/r2_t14.log 2>&1 || true
tail -25 gpurun_out/r2_t14.log
