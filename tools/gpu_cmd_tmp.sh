#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python tools/bench_configs.py cmc_td3 2>&1 | grep "^{\|Error\|error" | cut -c1-260
