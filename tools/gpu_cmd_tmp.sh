#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in t3w_wide; do echo "== $v"; LENV_TIMING_POP=8 LENV_TIMING_LIB=gpurun_out_lib_$v.so timeout 300 python tools/phase_timing_t3w.py 2>&1 | grep "wgrad\|generation wall" ; done
