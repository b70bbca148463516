#!/bin/bash
# Dynamic VALU instruction count per kernel section: ablation build under rocprofv3 --pmc.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SSME_PF_LIB=$PWD/ssme_amd/libssme_pf_ablate.so
for m in 0 1 2 4 8 16 32 63; do
  export SSME_ABLATE_MASK=$m
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d gpurun_out/ablpmc/m$m -- python3 tools/prof_run.py --T 12 --passes 1 --nt ${NT:-512} --resampler ${RS:-0} > gpurun_out/ablpmc_m$m.log 2>&1
  echo "mask $m rc=$?"
done
