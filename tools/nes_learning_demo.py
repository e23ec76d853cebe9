#!/usr/bin/env python3
"""Demonstration (run on a GPU box): NES meta-training of a CartPole synthetic environment at BASELINE configs[1]'s FULL size
(pop 64, DDQN agents with up to 1000 training episodes x 200 steps and early-out, default_config_cartpole_syn_env.yaml) --
GTN_Master.step() per generation, one JSON line each: the mean real-env return of agents trained on the unperturbed SE
(`mean_score_orig`), the best worker, the env steps taken and the wall time.  usage: nes_learning_demo.py [generations]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.makedirs("/tmp/lenv_demo", exist_ok=True); os.chdir("/tmp/lenv_demo")
import torch
from learning_environments_amd.agents.GTN import GTN_Master
from learning_environments_amd import configs
cfg = configs.cartpole_syn_env_ddqn(num_workers=64, max_iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 25)
torch.manual_seed(0)
m = GTN_Master(cfg, bohb_id=0, seed=1)
t0 = time.time()
hist = []
for it in range(m.max_iterations):
    mean_score, solved = m.step(it)
    st = m.inner.stats.cpu().numpy()
    hist.append(dict(gen=it, mean_score_orig=float(mean_score), best=float(max(m.score_list)), train_steps=int(st[:, 1].sum()), wall_s=round(time.time() - t0, 2)))
    print(json.dumps(hist[-1]), flush=True)
    if solved:
        print("solved at generation", it); break
