#!/usr/bin/env python3
"""Diagnostic: time per learn step of the other published DDQN shapes (default_config_cartpole.yaml, default_config_acrobot_syn_env.yaml),
specialised instantiation by default, generic with LENV_NO_FIXED_SHAPE=1.  usage: ab_shapes23.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from learning_environments_amd import configs, engine
from learning_environments_amd.config import ddqn_cfg_from_config
for which in ("cartpole", "acrobot_syn_env"):
    cfgd = configs.fixed_work(configs.cartpole_syn_env_ddqn(64), 20)
    if which == "cartpole":
        cfgd["envs"]["CartPole-v0"].update(hidden_size=128)
        cfgd["agents"]["ddqn"].update(hidden_size=64, batch_size=32, activation_fn="relu", test_episodes=1)
    else:
        cfgd["env_name"] = "Acrobot-v1"
        cfgd["envs"] = {"Acrobot-v1": {"solved_reward": -100.0, "max_steps": 500, "activation_fn": "prelu", "hidden_size": 167, "hidden_layer": 1,
                                       "info_dim": 0, "reward_env_type": 0}}
        cfgd["agents"]["ddqn"].update(hidden_size=112, batch_size=149, activation_fn="leakyrelu", test_episodes=10, train_episodes=4)
    cfg = ddqn_cfg_from_config(cfgd)
    chains = 192
    rng = np.random.RandomState(3)
    il = engine.InnerLoop(cfg, chains)
    S, A, H = cfg.state_dim, cfg.num_actions, cfg.se_hidden
    P_se = ((S + A) * H + H + H * S + S) + 2 * ((S + A) * H + H + H + 1)
    d = torch.device("cuda")
    theta = torch.from_numpy((rng.randn(P_se) * 0.15).astype(np.float32)).to(d)
    eps = torch.from_numpy((rng.randn(64, P_se) * 0.0124).astype(np.float32)).to(d)
    init = torch.from_numpy(rng.uniform(-0.3, 0.3, (chains, il.p_agent)).astype(np.float32)).to(d)
    worker = torch.from_numpy(np.repeat(np.arange(64), 3).astype(np.int32)).to(d)
    sign = torch.from_numpy(np.tile(np.array([0.0, 1.0, -1.0], np.float32), 64)).to(d)
    keys = torch.arange(chains, dtype=torch.int64, device=d) * 7919 + 13
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        il.run(theta, eps, worker, sign, init, rng_keys=keys)
        torch.cuda.synchronize(); dt = time.time() - t0
    st = il.stats.cpu().numpy()
    print("%-16s us per learn step per chain: %.3f  (%.2f ms per launch)" % (which, dt * 1e6 / st[0, 2], dt * 1e3))
