#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic_env_of_the_ddqn or test_g8_calc_score or layer_norm" 2>&1 | tail -15
