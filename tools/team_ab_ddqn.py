#!/usr/bin/env python3
"""Measure (one GPU) what a DDQN chain on a team of workgroups buys at the chain counts the strong-scaling shards of BASELINE
configs[1] put on a GPU: pop 64 over 2 / 4 / 8 GPUs = 32 / 16 / 8 workers = 96 / 48 / 24 chains.  For each, the generation time
with one workgroup per chain (gtn.team_size 1) and with the automatic team size.  usage: tools/team_ab_ddqn.py [steps]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from learning_environments_amd import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for pop in (64, 32, 16, 8):
    row = {"workers_on_this_gpu": pop, "chains": 3 * pop}
    for mode in ("1", "auto"):
        master, cfgd = bench.build_master(pop, team_size=1 if mode == "1" else 0)
        G = _lib.lib().lenv_ddqn_se_team_size(C.byref(master.cfg), 3 * pop)
        master.step(0)
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(steps):
            master.step(1 + it)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        assert master.inner.status.cpu().abs().max().item() == 0
        row["G%s" % ("1" if mode == "1" else "auto")] = {"workgroups_per_chain": G, "ms_per_generation": round(ms, 3), "evals_per_s": round(pop / ms * 1e3, 1)}
        del master
    row["speedup"] = round(row["G1"]["ms_per_generation"] / row["Gauto"]["ms_per_generation"], 3)
    print(json.dumps(row), flush=True)
