#!/bin/bash
# Diagnostic: cycles per GEMM product, current lenv_gemm.cuh vs the committed one (tools/ubench/_old, scratch copy).
# usage: run_gemm_ubench.sh [extra -D flags for additional builds of the current header, one build per argument]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F tools/ubench/gemm_ubench.hip -o /tmp/gemm_ub_new 2>/dev/null && echo "== new" && /tmp/gemm_ub_new
for D in "$@"; do
  /opt/rocm/bin/hipcc $F $D tools/ubench/gemm_ubench.hip -o /tmp/gemm_ub_d 2>/dev/null && echo "== new $D" && /tmp/gemm_ub_d | head -3
done
if [ -d tools/ubench/_old ]; then (cd tools/ubench/_old && /opt/rocm/bin/hipcc $F tools/ubench/gemm_ubench.hip -o /tmp/gemm_ub_old 2>/dev/null) && echo "== old" && /tmp/gemm_ub_old; fi
