"""Inner-loop tasks: which fused kernel evaluates a population member for a given (inner agent, synthetic env type).

The reference dispatches through select_agent (agents/agent_utils.py:15-66) + EnvFactory; here each supported
combination owns one fused kernel:
    DDQN on a VirtualEnv (synthetic_env_type 0)   -> lenv_ddqn_se_inner_loop   (BASELINE configs 1-2, Acrobot-DDQN)
    DuelingDDQN on a VirtualEnv                   -> lenv_dueling_se_inner_loop (BASELINE config 3)
    DDQN_vary / DuelingDDQN_vary on a VirtualEnv  -> lenv_dueling_se_inner_loop_hp (per-chain lr / batch / width / depth)
    TD3_vary on a RewardEnv over the stand-in     -> lenv_td3_rn_inner_loop_hp
    QL / QL_cb / SARSA / SARSA_cb on a RewardEnv over a gridworld (type 1) -> lenv_ql_rn_inner_loop (BASELINE config 4)
    TD3  on a RewardEnv over the HalfCheetah stand-in -> lenv_td3_rn_inner_loop (BASELINE config 5)
    TD3_discrete_vary on a VirtualEnv (CartPole / Acrobot / MountainCar) -> lenv_td3d_inner_loop
Anything else raises NotImplementedError, like the reference does for unknown agents."""
import numpy as np
import torch

from ..config import (TABULAR_AGENTS, TD3_DISCRETE_ENVS, agent_layer_dims, agent_layer_norm_slice, ddqn_cfg_from_config, icm_layer_dims, ql_cfg_from_config,
                      td3_cfg_from_config, td3_layer_dims, td3_layer_norm_slices, td3d_cfg_from_config)
from .nes_common import chain_keys, fresh_agent_init, linear_init_bounds, set_layer_norm_init, with_layer_norm_block


class DdqnSeTask(object):
    name = "ddqn_se"

    def __init__(self, config, engine, test_mode=0):
        self.engine = engine
        self.cfg = ddqn_cfg_from_config(config, test_mode=test_mode) if engine.name == "hip" else engine.cfg_from_config(config)
        dims = agent_layer_dims(self.cfg)
        self.ln_slice = agent_layer_norm_slice(self.cfg)      # use_layer_norm: the shared LayerNorm's block in the flat parameter vector
        self.agent_bounds = torch.from_numpy(with_layer_norm_block(linear_init_bounds(dims), self.ln_slice)).to(engine.device)
        # "ddqn_icm" / "duelingddqn_icm": the agent carries an Intrinsic Curiosity Module (agents/DDQN.py:40-58); every chain
        # gets a fresh one (nn.Linear default init), drawn from its own counter-RNG stream
        self.icm_bounds = None
        if engine.name == "hip" and self.cfg.icm_enabled:
            self.icm_bounds = torch.from_numpy(linear_init_bounds(icm_layer_dims(self.cfg))).to(engine.device)

    def make_inner(self, chains, want_episode_stats=True):
        return self.engine.make_inner(self.cfg, chains, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        if self.icm_bounds is not None:
            inner.draw_icm_init(keys_t, self.icm_bounds)
        set_layer_norm_init(agent_init, self.ln_slice)
        return self.engine.inner_scores(inner, theta, eps, chain_worker, chain_sign, agent_init, keys_t)

    def needs_agent_init(self):
        return True


class DdqnVaryTask(object):
    """DDQN_vary / DuelingDDQN_vary on a VirtualEnv (agents/DDQN_vary.py, agents/DuelingDDQN_vary.py): every chain draws its
    own lr / batch_size / hidden_size / hidden_layer (agents/vary.py) and the whole heterogeneous population still runs as ONE
    launch of the GEMM-tiled kernel (per-chain hyper-parameter arrays, workspace sized for the largest draw)."""
    name = "ddqn_vary_se"

    def __init__(self, config, engine, test_mode=0):
        import copy
        from . import vary
        self.engine = engine              # HipNesEngine, or the test suite's oracle-backed stand-in (CPU tensors)
        self.agent_key = config["agents"]["gtn"]["agent_name"].lower()[:-5]
        if self.agent_key.endswith("_icm"):               # "ddqn_icm_vary": DDQN_vary(icm=True), agents/agent_utils.py:43-44
            self.agent_key = self.agent_key[:-4]
        self.base = config["agents"][self.agent_key]
        bd = vary.hp_bounds(self.base)
        big = copy.deepcopy(config)
        big["agents"][self.agent_key].update(batch_size=bd["batch_size"][1], hidden_size=bd["hidden_size"][1],
                                             hidden_layer=bd["hidden_layer"][1])
        self.cfg = ddqn_cfg_from_config(big, test_mode=test_mode)      # the maxima: workspace / LDS / row strides
        self.cfg.grad_chunk = 0                           # one sequential batch gradient (GEMM-tiled kernel)
        self.agent_bounds = None
        self.last_hp = None
        self.fixed_hp = None
        self.icm_bounds = None
        if engine.name == "hip" and self.cfg.icm_enabled:
            self.icm_bounds = torch.from_numpy(linear_init_bounds(icm_layer_dims(self.cfg))).to(engine.device)

    def make_inner(self, chains, want_episode_stats=True):
        return self.engine.make_inner(self.cfg, chains, want_episode_stats=want_episode_stats, vary=True)

    def draw_hp(self, keys):
        from . import vary
        if self.fixed_hp is not None:     # a recorded draw replayed (parity tests against the reference's runs)
            return list(self.fixed_hp)
        return [vary.vary_hyperparameters(self.base, vary.chain_units(k)) for k in keys]

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        # the draws are a host function of the chain keys (ConfigSpace's role in the reference); reading the keys back
        # waits only for the generation's draw kernel
        keys = keys_t.cpu().numpy().view(np.uint64)
        hp = self.last_hp = self.draw_hp(keys)
        inner.set_hp([h["lr"] for h in hp], [h["batch_size"] for h in hp], [h["hidden_size"] for h in hp],
                     [h["hidden_layer"] for h in hp])
        inner.draw_agent_init(keys_t)
        if self.icm_bounds is not None:
            inner.draw_icm_init(keys_t, self.icm_bounds)
        return self.engine.inner_scores(inner, theta, eps, chain_worker, chain_sign, None, keys_t)

    def needs_agent_init(self):
        return False          # drawn inside scores(), once the chains' shapes are known


class QlRnTask(object):
    name = "ql_rn"

    def __init__(self, config, engine, tables, test_mode=0):
        self.engine = engine
        self.tables = tables
        self.cfg = ql_cfg_from_config(config, tables, test_mode=test_mode)
        self.agent_bounds = None

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_ql(self.cfg, chains, self.tables, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        return self.engine.inner_scores_ql(inner, theta, eps, chain_worker, chain_sign, keys_t)

    def needs_agent_init(self):
        return False          # a fresh QL agent is an all-zero table (QL.py:25)


class Td3RnTask(object):
    name = "td3_rn"

    def __init__(self, config, engine, test_mode=0):
        self.engine = engine
        self.cfg = td3_cfg_from_config(config, test_mode=test_mode)
        self.ln_slice = td3_layer_norm_slices(self.cfg)       # use_layer_norm: the three nets' LayerNorm blocks in the flat parameter vector
        self.agent_bounds = torch.from_numpy(with_layer_norm_block(linear_init_bounds(td3_layer_dims(self.cfg)), self.ln_slice)).to(engine.device)
        self.icm_bounds = None                    # "td3_icm": TD3(icm=True), a fresh ICM per chain
        if self.cfg.icm_enabled:
            self.icm_bounds = torch.from_numpy(linear_init_bounds(icm_layer_dims(self.cfg))).to(engine.device)

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_td3(self.cfg, chains, want_episode_stats=want_episode_stats)

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        if self.icm_bounds is not None:
            inner.draw_icm_init(keys_t, self.icm_bounds)
        set_layer_norm_init(agent_init, self.ln_slice)
        return self.engine.inner_scores_td3(inner, theta, eps, chain_worker, chain_sign, agent_init, keys_t)

    def needs_agent_init(self):
        return True


class Td3VaryTask(object):
    """TD3_vary on the stand-in RewardEnv (agents/TD3_vary.py:24-58): per-chain lr / batch_size / hidden_size / hidden_layer in
    one launch of the TD3 kernel (lenv_td3_rn_inner_loop_hp), like DdqnVaryTask."""
    name = "td3_vary_rn"

    def __init__(self, config, engine, test_mode=0):
        import copy
        from . import vary
        if engine.name != "hip":
            raise NotImplementedError("the *_vary agents need the HIP engine")
        self.engine = engine
        self.base = config["agents"]["td3"]
        bd = vary.hp_bounds(self.base)
        big = copy.deepcopy(config)
        big["agents"]["td3"].update(batch_size=bd["batch_size"][1], hidden_size=bd["hidden_size"][1], hidden_layer=bd["hidden_layer"][1])
        self.cfg = td3_cfg_from_config(big, test_mode=test_mode)
        self.agent_bounds = None
        self.last_hp = None
        self.fixed_hp = None
        self.icm_bounds = None
        if self.cfg.icm_enabled:
            self.icm_bounds = torch.from_numpy(linear_init_bounds(icm_layer_dims(self.cfg))).to(engine.device)

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_td3(self.cfg, chains, want_episode_stats=want_episode_stats, vary=True)

    def draw_hp(self, keys):
        from . import vary
        if self.fixed_hp is not None:     # a recorded draw replayed (parity tests against the reference's runs)
            return list(self.fixed_hp)
        return [vary.vary_hyperparameters(self.base, vary.chain_units(k)) for k in keys]

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        keys = keys_t.cpu().numpy().view(np.uint64)
        hp = self.last_hp = self.draw_hp(keys)
        inner.set_hp([h["lr"] for h in hp], [h["batch_size"] for h in hp], [h["hidden_size"] for h in hp],
                     [h["hidden_layer"] for h in hp])
        inner.draw_agent_init(keys_t)
        if self.icm_bounds is not None:
            inner.draw_icm_init(keys_t, self.icm_bounds)
        return self.engine.inner_scores_td3(inner, theta, eps, chain_worker, chain_sign, None, keys_t)

    def needs_agent_init(self):
        return False


class Td3DiscreteTask(object):
    """TD3_discrete_vary on a VirtualEnv (agents/TD3_discrete_vary.py): one launch of lenv_td3d_inner_loop per generation.  With
    vary_hp (:21-26,119-157) every chain draws its own lr / batch_size / hidden_size / hidden_layer (agents/vary.py) and the launch is
    sized for the largest possible draw, like Td3VaryTask.  The fresh agents (nn.Linear default init, LayerNorm 1 / 0) are drawn on
    the device from the chain keys."""
    name = "td3_discrete_se"

    def __init__(self, config, engine, test_mode=0):
        import copy
        from . import vary
        if engine.name != "hip":
            raise NotImplementedError("TD3_discrete_vary needs the HIP engine")
        self.engine = engine
        self.base = config["agents"]["td3_discrete_vary"]
        self.vary = bool(self.base["vary_hp"])
        big = config
        if self.vary:
            bd = vary.hp_bounds(self.base)
            big = copy.deepcopy(config)
            big["agents"]["td3_discrete_vary"].update(batch_size=bd["batch_size"][1], hidden_size=bd["hidden_size"][1],
                                                      hidden_layer=bd["hidden_layer"][1])
        self.cfg = td3d_cfg_from_config(big, test_mode=test_mode)
        self.agent_bounds = None
        self.last_hp = None
        self.fixed_hp = None

    def make_inner(self, chains, want_episode_stats=False):
        return self.engine.make_inner_td3d(self.cfg, chains, want_episode_stats=want_episode_stats, vary=self.vary)

    def draw_hp(self, keys):
        from . import vary
        if self.fixed_hp is not None:     # a recorded draw replayed (parity tests against the reference's runs)
            return list(self.fixed_hp)
        return [vary.vary_hyperparameters(self.base, vary.chain_units(k)) for k in keys]

    def scores(self, inner, theta, eps, chain_worker, chain_sign, keys_t, agent_init):
        if self.vary:
            keys = keys_t.cpu().numpy().view(np.uint64)
            hp = self.last_hp = self.draw_hp(keys)
            inner.set_hp([h["lr"] for h in hp], [h["batch_size"] for h in hp], [h["hidden_size"] for h in hp],
                         [h["hidden_layer"] for h in hp])
        inner.draw_agent_init(keys_t)
        return self.engine.inner_scores_td3(inner, theta, eps, chain_worker, chain_sign, None, keys_t)

    def needs_agent_init(self):
        return False          # drawn inside scores() (the LayerNorm parameters are not uniform draws)


TD3_ENVS = ("HalfCheetah-v3", "Pendulum-v0", "MountainCarContinuous-v0")       # continuous real envs of the TD3 kernel


def select_task(config, engine, synthetic_env, test_mode=0):
    """test_mode: lenv_ddqn_cfg::test_mode -- 0 = GTN_Worker.calc_score's train(env, test_env=real_env); 1 = train(env) without a test env
    (the evaluation harness, experiments/syn_env_evaluate.py)."""
    agent_name = config["agents"]["gtn"]["agent_name"].lower()
    env_type = config["agents"]["gtn"]["synthetic_env_type"]
    tm = dict(test_mode=int(test_mode))
    if agent_name in ("ddqn", "duelingddqn", "ddqn_icm", "duelingddqn_icm") and env_type == 0:
        return DdqnSeTask(config, engine, **tm)
    if agent_name in ("ddqn_vary", "duelingddqn_vary", "ddqn_icm_vary", "duelingddqn_icm_vary") and env_type == 0:
        # vary_hp False: the agent IS its base agent (DDQN_vary.py:16-21); the *_icm_vary names read the same `<agent>_vary`
        # section (DDQN_vary.py:16) and switch the ICM on
        section = agent_name.replace("_icm", "")
        return DdqnVaryTask(config, engine, **tm) if config["agents"][section]["vary_hp"] else DdqnSeTask(config, engine, **tm)
    if agent_name in ("ddqn", "duelingddqn", "ddqn_icm", "duelingddqn_icm") and env_type == 1 and config["env_name"] in ("CartPole-v0", "Acrobot-v1", "MountainCar-v0"):
        return DdqnSeTask(config, engine, **tm)         # RewardEnv over the real env (default_config_cartpole_reward_env.yaml): same kernel
    if agent_name in ("ddqn_vary", "duelingddqn_vary", "ddqn_icm_vary", "duelingddqn_icm_vary") and env_type == 1 and config["env_name"] in ("CartPole-v0", "Acrobot-v1", "MountainCar-v0"):
        # the *_vary agents on a RewardEnv / on the real env itself (experiments/syn_env_run_vary_hp.py:47-54, mode 0): same kernel, per-chain shapes
        section = agent_name.replace("_icm", "")
        return DdqnVaryTask(config, engine, **tm) if config["agents"][section]["vary_hp"] else DdqnSeTask(config, engine, **tm)
    if agent_name in TABULAR_AGENTS and env_type == 1:
        real = synthetic_env.env.real_env
        if not hasattr(real, "tables"):
            raise NotImplementedError("QL needs a discrete (gridworld) real env")
        return QlRnTask(config, engine, real.tables, **tm)
    # TD3 on the HalfCheetah stand-in / Pendulum-v0 / MountainCarContinuous-v0: RewardEnv (type 1, BASELINE config 5) or VirtualEnv (type 0, default_config_halfcheetah.yaml)
    if agent_name in ("td3", "td3_icm") and env_type in (0, 1) and config["env_name"] in TD3_ENVS:
        return Td3RnTask(config, engine, **tm)
    if agent_name in ("td3_vary", "td3_icm_vary") and env_type in (0, 1) and config["env_name"] in TD3_ENVS:
        return Td3VaryTask(config, engine, **tm) if config["agents"]["td3_vary"]["vary_hp"] else Td3RnTask(config, engine, **tm)
    if agent_name == "td3_discrete_vary" and env_type == 0 and config["env_name"] in TD3_DISCRETE_ENVS:
        return Td3DiscreteTask(config, engine, **tm)
    raise NotImplementedError("inner agent '%s' on synthetic_env_type %s has no fused kernel yet" % (agent_name, env_type))
