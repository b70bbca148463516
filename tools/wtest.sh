#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SSME_PF_LIB=$PWD/ssme_amd/libssme_pf_ablate.so
i=0
for cfg in "0 0" "63 0" "0 1" "4 1"; do
  set -- $cfg; export SSME_ABLATE_MASK=$1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/wtest/c$i -- python3 tools/prof_run.py --T 6 --passes 1 --resampler $2 > gpurun_out/wtest_$i.log 2>&1
  echo "cfg mask=$1 rs=$2 rc=$?"; i=$((i+1))
done
