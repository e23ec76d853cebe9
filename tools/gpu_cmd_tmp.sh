#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_host_api.py -m gpu -x -q -k "continuous_action_virtual_env or host_mirrors or envwrapper_step" 2>&1 | tail -30
