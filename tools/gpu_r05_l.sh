#!/bin/bash
# round 5, GPU batch L: default_config_cmc.yaml on the wave-chain kernel against the GEMM-queue kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1200 python tools/bench_configs.py ${1:-cmc_venv_td3} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])" 2>&1 | tee gpurun_out/r05l_${1:-cmc_venv_td3}.log
