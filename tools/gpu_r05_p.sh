#!/bin/bash
# round 5, GPU batch P: the GEMM-queue TD3 kernel's one-row products (batched loads, the three SE nets side by side): A/B on the shipped
# small-net / generic VirtualEnv configurations, then every TD3 test
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
    timeout 600 python tools/bench_configs.py cmc_opt_td3 venv_td3 cmc_venv_td3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'GEMM' in d['config']: print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])"
  done
done 2>&1 | tee gpurun_out/r05p_ab.log
cp $ORIG learning_environments_amd/liblenv_hip.so
timeout 1500 python -m pytest tests -x -q -m gpu -k "td3 and not td3d and not discrete" 2>&1 | tail -5 | tee gpurun_out/r05p_td3.log
