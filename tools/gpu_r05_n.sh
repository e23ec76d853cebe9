#!/bin/bash
# round 5, GPU batch N: the shipped fixed-shape configurations that still run on the GEMM-queue kernels -- what a learn step costs there
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1200 python tools/bench_configs.py mountaincar_ddqn cartpole_rn_ddqn cmc_opt_td3 2>gpurun_out/r05n.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(round(d['s_per_generation'] * 1e3, 1), 'ms', round(d['us_per_learn_step_per_chain'], 1), 'us per learn step;', d['chains'], 'chains;', d['config'])" 2>&1 | tee gpurun_out/r05n_generic_shapes.log
tail -3 gpurun_out/r05n.err
