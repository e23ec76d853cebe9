#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests -m gpu -q -x -k "plain_dqn or (dueling and wavechain) or acrobot" 2>&1 | tail -15
timeout 600 python tools/bench_configs.py acrobot_ddqn 2>&1 | grep "^{" | cut -c1-230
