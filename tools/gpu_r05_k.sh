#!/bin/bash
# round 5, GPU batch K: the TD3 wave-chain kernel's fourth shape (default_config_cmc.yaml) -- its parity tests first, then every TD3 test
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_env_shapes or cmc_virtual_env or other_published_shapes" 2>&1 | tail -25 | tee gpurun_out/r05k_new.log
if [ "$1" != "quick" ]; then timeout 1500 python -m pytest tests -x -q -m gpu -k "td3 and not td3d and not discrete" 2>&1 | tail -8 | tee gpurun_out/r05k_td3.log; fi
