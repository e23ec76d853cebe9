#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python tools/bench_configs.py acrobot_ddqn 3 2>&1 | grep "^{"
