#!/usr/bin/env python3
"""Experiment (round 6, docs/notebook_r06.md section 1): what would a primary workgroup gain if a HELPER workgroup took part of its minibatch?
Upper bound by emulation: 96 chains on teams of two = 192 workgroups, dealt evenly (6 + 6 micro-chunks, what ships) or unevenly (9 + 3, 10 + 2:
builds with -DLENV_DDQN_UNEVEN_MC=9 / 10) -- the generation ends when member 0 ends, i.e. this is the learn step of a primary whose helper
serves ONE chain and has nothing else to do; a helper serving three chains can only be slower.  Against: 96 and 192 chains with one workgroup
per chain.  usage: LENV_TIMING_LIB=<variant.so> tools/uneven_team_ab.py <label> [steps]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_environments_amd import _lib
if os.environ.get("LENV_TIMING_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LENV_TIMING_LIB"])
import torch
import bench

label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for pop, team in ((32, 2), (32, 1), (64, 1)):
    master, cfgd = bench.build_master(pop, team_size=team)
    G = _lib.lib().lenv_ddqn_se_team_size(C.byref(master.cfg), 3 * pop)
    for it in range(2):
        master.step(it)
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(steps):
        master.step(2 + it)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    assert master.inner.status.cpu().abs().max().item() == 0
    print(json.dumps({"lib": label, "chains": 3 * pop, "workgroups_per_chain": G, "ms_per_generation": round(ms, 3),
                      "score_checksum": float(master.inner.score.sum().item())}), flush=True)
    del master
    torch.cuda.empty_cache()
