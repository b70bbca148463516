#!/usr/bin/env python3
"""Timing A/B builds of the library: python3 tools/ab_build.py NAME -DMACRO ...  ->  build/ab/lib_NAME.so
(build/ is git-ignored but travels to the GPU box with the snapshot; gpurun_out/ does not).
Run a variant with SSME_PF_LIB=build/ab/lib_NAME.so python3 tools/prof_run.py ...  (results of a variant may differ
from the specification; these builds exist to compare kernel timings on ONE box in ONE gpurun call)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssme_amd import build  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "build", "ab"), exist_ok=True)
print(build.build(force=True, extra=tuple(extra), out=os.path.join(ROOT, "build", "ab", f"lib_{name}.so")))
