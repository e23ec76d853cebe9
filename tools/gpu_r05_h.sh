#!/bin/bash
# round 5, GPU batch H: headline A/B of two libraries on ONE box (tools/ab_bench.sh) + the DDQN parity tests on the second
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
bash tools/ab_bench.sh $1 $2 3 2>&1 | tee gpurun_out/r05h_ab.log
cp learning_environments_amd/liblenv_hip.so /tmp/liblenv_hip_orig.so
trap 'cp /tmp/liblenv_hip_orig.so learning_environments_amd/liblenv_hip.so' EXIT
cp $2 learning_environments_amd/liblenv_hip.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_long_horizon.py -m gpu -x -q -k "inner_loop or ddqn or team or cartpole" 2>&1 | tail -3 | tee gpurun_out/r05h_pytest.log
