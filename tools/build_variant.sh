#!/bin/bash
# Cross-compile ONE kernel source with extra flags here and link it with the other kernels' objects of the normal build:
#   tools/build_variant.sh NAME SRC.hip [extra hipcc flags]  ->  gpurun_out_lib_NAME.so  (git-ignored; travels to the GPU box;
#   the diagnostics load it through LENV_TIMING_LIB=gpurun_out_lib_NAME.so).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/learning_environments_amd/csrc
n=$1; src=$2; shift; shift
base=$(basename $src .hip)
mkdir -p /tmp/lenv_variants
extra=""
if [ "$base" = "ddqn_se_inner_loop" ]; then extra="-fno-slp-vectorize"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC $extra "$@" -c $C/$base.hip -o /tmp/lenv_variants/${base}_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out_lib_$n.so /tmp/lenv_variants/${base}_$n.o $(ls $C/_build/*.o | grep -v "/$base.o")
echo built $R/gpurun_out_lib_$n.so
