#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reward_env" 2>&1 | tail -15
