#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x -k "layer_norm or td3_cheetah" 2>&1 | tail -15
