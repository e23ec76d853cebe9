#!/bin/bash
# round 5, GPU batch J: A/B of two libraries on the GEMM-queue TD3 shapes (tools/bench_configs.py pendulum_td3 cmc_td3), alternating on ONE box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ORIG=/tmp/liblenv_hip_orig.so
cp learning_environments_amd/liblenv_hip.so $ORIG
trap 'cp $ORIG learning_environments_amd/liblenv_hip.so' EXIT
for round in 1 2; do
  for v in "$@"; do
    cp $v learning_environments_amd/liblenv_hip.so
    echo "== $v (round $round)"
    timeout 600 python tools/bench_configs.py pendulum_td3 cmc_td3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'GEMM' in d['config']: print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])"
  done
done 2>&1 | tee gpurun_out/r05j_ab.log
