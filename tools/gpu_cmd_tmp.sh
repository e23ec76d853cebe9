#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_td3_discrete.py tests/test_abi_and_host.py -m gpu -x -q -k "layer_norm or layernorm or ql_rn or td3_pendulum or td3_virtual_env or td3_discrete or tape_mode or abi" 2>&1 | tail -15
