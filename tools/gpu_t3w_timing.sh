#!/bin/bash
# GPU-box diagnostic: phase timing of the TD3 wave-chain kernel at the cfg5 shard (24 chains, teams) with prebuilt variant libraries
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for v in t3w_timing t3w_timing_sub; do
  if [ -f gpurun_out_lib_$v.so ]; then
    LENV_TIMING_POP=${LENV_TIMING_POP:-8} LENV_TIMING_LIB=gpurun_out_lib_$v.so timeout 300 python tools/phase_timing_t3w.py 2>&1 | grep -v "warning" | tail -40 | tee gpurun_out/$v.log
  fi
done
timeout 300 python bench.py --only-config 4 2>gpurun_out/bench_c4.err | tail -1 | tee gpurun_out/bench_c4.json
