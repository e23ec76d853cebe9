#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests -m gpu -q -x -k "wavechain_td3" 2>&1 | tail -4
