"""Trained-SE checkpoint round trip + "train agents on the SE, test on the real env" evaluation.

Mirrors the two functions every downstream experiment of the reference starts from
(experiments/syn_env_evaluate_cartpole_vary_hp_2.py:12-48, same names / arguments / return shapes):

    load_envs_and_config(file_name, model_dir, device) -> (virtual_env, real_env, config)
        reads a reference-format checkpoint {'model': state_dict, 'config': dict} (agents/GTN_master.py:133-139)
    train_test_agents(train_env, test_env, config, agents_num) -> (reward_list, train_steps_needed, episodes_needed)
        trains `agents_num` fresh agents on `train_env` (BaseAgent.train) and tests each on the real env (BaseAgent.test)

Here the `agents_num` agents are independent chains of ONE fused-kernel launch (sign = 0: the unperturbed checkpoint
weights).  The reference's experiment trains `DDQN_vary` (hyper-parameter resampling, agents/DDQN_vary.py:26-59); that
variant is not built (§8(f).2) -- this harness trains the configured inner agent (`agent_name`, default the config's GTN
agent) and raises NotImplementedError for `*_vary`."""
import os

import numpy as np
import torch

from ..agents import tasks
from ..agents.nes_common import chain_keys, fresh_agent_init
from ..engine import HipNesEngine
from ..envs.env_factory import EnvFactory


def load_envs_and_config(file_name, model_dir, device):
    file_path = os.path.join(model_dir, file_name)
    save_dict = torch.load(file_path, map_location="cpu")
    config = save_dict['config']
    config['device'] = device
    env_factory = EnvFactory(config=config)
    virtual_env = env_factory.generate_virtual_env()
    virtual_env.load_state_dict(save_dict['model'])
    real_env = env_factory.generate_real_env()
    return virtual_env, real_env, config


def train_test_agents(train_env, test_env, config, agents_num, agent_name=None, seed=0):
    """Returns (reward_list, train_steps_needed, episodes_needed) like the reference: reward_list[i] = the i-th agent's
    list of real-env test returns (BaseAgent.test), train_steps_needed[i] = [sum(episode_length)],
    episodes_needed[i] = [number of training episodes run]."""
    name = (agent_name or config["agents"]["gtn"]["agent_name"]).lower()
    if name.endswith("_vary"):
        raise NotImplementedError("agent '%s': hyper-parameter-resampling agents are not built (SURVEY.md §8(f).2)" % name)
    if test_env.is_virtual_env():
        raise ValueError("test_env must be the real environment")
    cfg = dict(config)
    cfg["agents"] = dict(config["agents"])
    cfg["agents"]["gtn"] = dict(config["agents"]["gtn"], agent_name=name)
    engine = HipNesEngine()
    task = tasks.select_task(cfg, engine, train_env)
    dev = engine.device
    theta = train_env.env.flat_params()
    chains = int(agents_num)
    inner = task.make_inner(chains, want_episode_stats=True)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    agent_init = fresh_agent_init(task.agent_bounds, chains, g, dev) if task.needs_agent_init() else None
    keys = chain_keys(int(seed), 0, np.arange(chains), np.zeros(chains, np.int64))
    keys_t = torch.from_numpy(keys.view(np.int64)).to(dev)
    worker = torch.zeros(chains, dtype=torch.int32, device=dev)
    sign = torch.zeros(chains, dtype=torch.float32, device=dev)
    eps = torch.zeros((1, theta.numel()), dtype=torch.float32, device=dev)
    task.scores(inner, theta, eps, worker, sign, keys_t, agent_init)
    engine.check_status(inner)
    stats = inner.stats.cpu().numpy()
    finals = inner.final_returns.cpu().numpy()
    reward_list = [finals[i].tolist() for i in range(chains)]
    train_steps_needed = [[int(stats[i, 1])] for i in range(chains)]
    episodes_needed = [[int(stats[i, 0])] for i in range(chains)]
    return reward_list, train_steps_needed, episodes_needed
