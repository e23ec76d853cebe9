#!/bin/bash
# round 5, GPU batch D: -fno-optimize-sibling-calls (callees stop saving callee-saved VGPRs through scratch, see DESIGN) against the
# shipped build: configs[2] / configs[4] shards, the GEMM-queue shapes of tools/bench_configs.py, alternating on ONE box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ORIG=/tmp/liblenv_hip_orig.so
cp learning_environments_amd/liblenv_hip.so $ORIG
trap 'cp $ORIG learning_environments_amd/liblenv_hip.so' EXIT
A=${1:-$ORIG}; B=${2:-gpurun_out_lib_nosiball.so}
for round in 1 2; do
  for v in $A $B; do
    cp $v learning_environments_amd/liblenv_hip.so
    for n in 2 4; do
      timeout 300 python bench.py --only-config $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('$v cfg[$n]', round(d['ms_per_step'],1), 'ms', round(d['us_per_learn_step_per_chain'],2), 'us/learn')"
    done
  done
done 2>&1 | tee gpurun_out/r05d_ab.log
for v in $A $B; do
  cp $v learning_environments_amd/liblenv_hip.so
  echo "== $v"
  timeout 900 python tools/bench_configs.py 5 3 td3d acrobot_ddqn pendulum_td3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])"
done 2>&1 | tee gpurun_out/r05d_configs.log
cp gpurun_out_lib_nosiball.so learning_environments_amd/liblenv_hip.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dueling or td3 or wavechain" 2>&1 | tail -4 | tee gpurun_out/r05d_pytest.log
