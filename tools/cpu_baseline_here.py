#!/usr/bin/env python3
"""Run bench.py's CPU-baseline legs (oracle threads leg + file-IO worker-mode leg) without a GPU, e.g. in the build
container, to record the (i) reference <-> (ii) port pair of SURVEY.md §8(d) in BASELINE.md."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from learning_environments_amd.config import ddqn_cfg_from_config  # noqa: E402

cfgd = bench.bench_config(bench.POP)
_, theta = bench.host_theta(cfgd)
print(json.dumps(bench.cpu_baseline(cfgd, theta, grad_chunk=ddqn_cfg_from_config(cfgd).grad_chunk, file_io=True), indent=1))
