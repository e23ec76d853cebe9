#!/bin/bash
# round 5, GPU batch Q: the wave-chain TD3 kernel's VirtualEnv step on K-major matrices: A/B on the three VirtualEnv shapes, then their tests
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
    timeout 600 python tools/bench_configs.py venv_td3 cmc_venv_td3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'wave-chain' in d['config']: print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])"
  done
done 2>&1 | tee gpurun_out/r05q_ab.log
cp $ORIG learning_environments_amd/liblenv_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_env or other_published_shapes" 2>&1 | tail -5 | tee gpurun_out/r05q_tests.log
