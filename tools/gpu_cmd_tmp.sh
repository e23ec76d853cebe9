#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x -k "ddqn_layer_norm" 2>&1 | tail -15
