set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_t18.log 2>&1 || true
tail -5 gpurun_out/r2_t18.log
g++ -O2 -std=c++17 -Iinclude tools/step_latency.cpp -Lssme_amd -l:libssme_pf.so -Wl,-rpath,$PWD/ssme_amd -o tools/step_latency
./tools/step_latency tests/golden/spy_returns.csv > gpurun_out/r2_lat18.txt 2>&1
cat gpurun_out/r2_lat18.txt
