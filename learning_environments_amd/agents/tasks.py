"""Inner-loop tasks: which fused kernel evaluates a population member for a given (inner agent, synthetic env type).

The reference dispatches through select_agent (agents/agent_utils.py:15-66) + EnvFactory; here each supported
combination owns one fused kernel:
    DDQN on a VirtualEnv (synthetic_env_type 0)   -> lenv_ddqn_se_inner_loop   (BASELINE configs 1-2, Acrobot-DDQN)
    DuelingDDQN on a VirtualEnv                   -> lenv_dueling_se_inner_loop (BASELINE config 3)
    QL / QL_cb / SARSA / SARSA_cb on a RewardEnv over a gridworld (type 1) -> lenv_ql_rn_inner_loop (BASELINE config 4)
    TD3  on a RewardEnv over the HalfCheetah stand-in -> lenv_td3_rn_inner_loop (BASELINE config 5)
Anything else raises NotImplementedError, like the reference does for unknown agents."""
import numpy as np
import torch

from ..config import TABULAR_AGENTS, agent_layer_dims, ddqn_cfg_from_config, ql_cfg_from_config, td3_cfg_from_config, td3_layer_dims
from .nes_common import chain_keys, fresh_agent_init, linear_init_bounds


class DdqnSeTask(object):
    name = "ddqn_se"

    def __init__(self, config, engine):
        self.engine = engine
        self.cfg = ddqn_cfg_from_config(config) if engine.name == "hip" else engine.cfg_from_config(config)
        dims = agent_layer_dims(self.cfg) if engine.name == "hip" else [(self.cfg.state_dim, self.cfg.q_hidden),
                                                                         (self.cfg.q_hidden, self.cfg.num_actions)]
        self.agent_bounds = torch.from_numpy(linear_init_bounds(dims)).to(engine.device)

    def make_inner(self, chains, want_episode_stats=True):
        return self.engine.make_inner(self.cfg, chains, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        return self.engine.inner_scores(inner, theta, eps, chain_worker, chain_sign, agent_init, keys_t)

    def needs_agent_init(self):
        return True


class QlRnTask(object):
    name = "ql_rn"

    def __init__(self, config, engine, tables):
        self.engine = engine
        self.tables = tables
        self.cfg = ql_cfg_from_config(config, tables)
        self.agent_bounds = None

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_ql(self.cfg, chains, self.tables, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        return self.engine.inner_scores_ql(inner, theta, eps, chain_worker, chain_sign, keys_t)

    def needs_agent_init(self):
        return False          # a fresh QL agent is an all-zero table (QL.py:25)


class Td3RnTask(object):
    name = "td3_rn"

    def __init__(self, config, engine):
        self.engine = engine
        self.cfg = td3_cfg_from_config(config)
        self.agent_bounds = torch.from_numpy(linear_init_bounds(td3_layer_dims(self.cfg))).to(engine.device)

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_td3(self.cfg, chains, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        return self.engine.inner_scores_td3(inner, theta, eps, chain_worker, chain_sign, agent_init, keys_t)

    def needs_agent_init(self):
        return True


def select_task(config, engine, synthetic_env):
    agent_name = config["agents"]["gtn"]["agent_name"].lower()
    env_type = config["agents"]["gtn"]["synthetic_env_type"]
    if agent_name in ("ddqn", "duelingddqn") and env_type == 0:
        return DdqnSeTask(config, engine)
    if agent_name in TABULAR_AGENTS and env_type == 1:
        real = synthetic_env.env.real_env
        if not hasattr(real, "tables"):
            raise NotImplementedError("QL needs a discrete (gridworld) real env")
        return QlRnTask(config, engine, real.tables)
    if agent_name == "td3" and env_type == 1:
        return Td3RnTask(config, engine)
    raise NotImplementedError("inner agent '%s' on synthetic_env_type %s has no fused kernel yet" % (agent_name, env_type))
