#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_parity.py -m gpu -x -q -k "host_mirrors_with_layer_norm or rn_shape_rows or reward_env or virtual_env or se_step" 2>&1 | tail -25
