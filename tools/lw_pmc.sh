# VALU / SALU / LDS instructions per wave of the Liu-West stage kernels (N = 2^20)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/lwpmc -- python3 tools/prof_run.py --lw --T 16 --passes 1 > gpurun_out/lwpmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/lwpmc
