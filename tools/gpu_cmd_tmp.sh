#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in wct_new; do echo "== $v"; LENV_TIMING_LIB=gpurun_out_lib_$v.so timeout 300 python tools/phase_timing_wc.py 2>&1 | tail -n +14; done
