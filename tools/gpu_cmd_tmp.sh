#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
cp gpurun_out_lib_new.so learning_environments_amd/liblenv_hip.so
timeout 1200 python -m pytest tests -m gpu -q -x -k "wavechain or (dueling and bench_launch)" 2>&1 | tail -5
bash tools/ab_config.sh 2 gpurun_out_lib_base.so gpurun_out_lib_new.so 3
timeout 600 python tools/bench_configs.py pendulum_td3 acrobot_ddqn 2>&1 | grep "^{" | grep "wave-chain" | cut -c1-200
