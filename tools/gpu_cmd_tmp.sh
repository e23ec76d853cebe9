#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests -m gpu -q -x -k "se_step" 2>&1 | tail -15
