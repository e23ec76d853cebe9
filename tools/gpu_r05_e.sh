#!/bin/bash
# round 5, GPU batch E: TD3 team learn step as one out-of-line routine (gpurun_out_lib_learnstep.so) against the build before it
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ORIG=/tmp/liblenv_hip_orig.so
cp learning_environments_amd/liblenv_hip.so $ORIG
trap 'cp $ORIG learning_environments_amd/liblenv_hip.so' EXIT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "td3" 2>&1 | tail -4 | tee gpurun_out/r05e_pytest.log
for round in 1 2 3; do
  for v in gpurun_out_lib_nosiball.so gpurun_out_lib_learnstep.so; do
    cp $v learning_environments_amd/liblenv_hip.so
    timeout 300 python bench.py --only-config 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('$v cfg[4]', round(d['ms_per_step'],1), 'ms', round(d['us_per_learn_step_per_chain'],2), 'us/learn')"
  done
done 2>&1 | tee gpurun_out/r05e_ab.log
cp $ORIG learning_environments_amd/liblenv_hip.so
timeout 600 python tools/bench_configs.py pendulum_td3 cmc_td3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['s_per_generation'] * 1e3, 1), 'ms', d['config'])" | tee gpurun_out/r05e_configs.log
