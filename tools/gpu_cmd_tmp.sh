#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "dueling or wavechain or acrobot" 2>&1 | tail -4
