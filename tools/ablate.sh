#!/bin/bash
# Timing-only ablation of kernel sections (measurement build -DSSME_ABLATE; results are invalid by design).
# bits: 0 normals, 1 E-gen, 2 search, 3 logG, 4 weight exp, 5 weight scan
export SSME_PF_LIB=$PWD/ssme_amd/libssme_pf_ablate.so
for m in 0 1 2 4 8 15 16 32 48 63; do
  echo -n "mask=$m  "; SSME_ABLATE_MASK=$m python3 tools/prof_run.py --T 256 --passes 3 --nt ${NT:-512} --resampler ${RS:-0} 2>&1 | tail -1
done
