#!/bin/bash
timeout 900 python3 tools/bench_configs.py acrobot_ddqn pendulum_td3 cmc_td3 2>/dev/null | tail -12
