#!/bin/bash
# Build a -DLENV_PHASE_TIMING variant of the DDQN kernel here (cross-compile) and link it with the other kernels' objects
# from the normal build: tools/build_timing_variant.sh NAME [extra hipcc flags]  ->  gpurun_out_lib_NAME.so (git-ignored;
# travels to the GPU box; load with LENV_TIMING_LIB=gpurun_out_lib_NAME.so python tools/phase_timing.py).  LENV_VARIANT_TIMING= (empty)
# builds without the stamps: a product-like library for tools/ab_bench.sh.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/learning_environments_amd/csrc
n=$1; shift
mkdir -p /tmp/lenv_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fno-slp-vectorize ${LENV_VARIANT_TIMING--DLENV_PHASE_TIMING} "$@" \
    -c $C/ddqn_se_inner_loop.hip -o /tmp/lenv_variants/ddqn_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out_lib_$n.so /tmp/lenv_variants/ddqn_$n.o $(ls $C/_build/*.o | grep -v ddqn_se_inner_loop)
