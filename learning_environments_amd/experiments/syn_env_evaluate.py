"""Trained-SE checkpoint round trip + the reference's evaluation harness.

Mirrors the two functions every downstream experiment of the reference starts from
(experiments/syn_env_evaluate_cartpole_vary_hp_2.py:12-48 and its siblings *_DuelingDDQN.py, *_TD3_discrete.py, syn_env_evaluate_acrobot_*;
same names / arguments / return shapes), called the way experiments/syn_env_run_vary_hp.py:32-117 calls them:

    load_envs_and_config(file_name, model_dir, device) -> (virtual_env, real_env, config)
        reads a reference-format checkpoint {'model': state_dict, 'config': dict} (agents/GTN_master.py:133-139)
    train_test_agents(train_env, test_env, config, agents_num) -> (reward_list, train_steps_needed, episodes_needed)
        applies the harness's "settings for comparability" (:29-36) to `config` IN PLACE like the reference, then for each of
        `agents_num` fresh `DDQN_vary` agents (hyper-parameters resampled per agent, agents/DDQN_vary.py:26-59):
            reward_train, episode_length, _ = agent.train(env=train_env)      # NO test env: base_agent.py:134-148 with test_env=None
            reward, _, _ = agent.test(env=test_env)
        `train_env` is the loaded VirtualEnv (modes 1 / 2 of run_vary_hp), a RewardEnv, or the REAL env (mode 0, the paper's baseline).

Here the `agents_num` agents are the chains of ONE fused-kernel launch with lenv_ddqn_cfg::test_mode = 1 (include/lenv_hip.h): the meter is
fed by the training env's own episode reward, training ends on the virtual rule (early_out_virtual_diff) or, on a real / reward env, on the
real rule; sign = 0 (the unperturbed checkpoint weights).  Pinned by fixtures G12 = the reference's own function run on a reference-written
checkpoint (tests/test_train_test_agents.py)."""
import copy
import os

import numpy as np
import torch

from ..agents import tasks
from ..agents.nes_common import chain_keys, fresh_agent_init
from ..engine import HipNesEngine, mlp_desc, mlp_num_params
from ..envs.env_factory import EnvFactory
from ..envs.reward_env import RewardEnv

# agent_name -> (config section the harness's settings go to, section that carries vary_hp); the sibling scripts differ in these only
LPT_MIN_CHAINS = 256      # launches above this many chains (= the compute units of an MI355X) run their expensive draws first (see _launch)
HARNESS_AGENTS = {"ddqn_vary": ("ddqn", "ddqn_vary"), "duelingddqn_vary": ("duelingddqn", "duelingddqn_vary"),
                  "td3_discrete_vary": ("td3_discrete_vary", "td3_discrete_vary")}


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


def apply_comparability_settings(config, agent_name="DDQN_vary", train_episodes=1000, test_episodes=10, early_out_num=10):
    """The block every harness script opens with (syn_env_evaluate_cartpole_vary_hp_2.py:29-36; 500 train episodes in
    syn_env_evaluate_acrobot_vary_hp_2_TD3_discrete.py; 100 test episodes and early_out_num 1000 in the *_correlation scripts, :41-43
    there).  Mutates `config` like the reference."""
    section, vary_section = HARNESS_AGENTS[agent_name.lower()]
    config['agents'][vary_section]['vary_hp'] = True
    config['agents'][section]['print_rate'] = 10
    config['agents'][section]['early_out_num'] = early_out_num
    config['agents'][section]['train_episodes'] = train_episodes
    config['agents'][section]['init_episodes'] = 10
    config['agents'][section]['test_episodes'] = test_episodes
    config['agents'][section]['early_out_virtual_diff'] = 0.01
    return config


def _task_config(train_env, config, agent_name):
    """The configuration the fused launch is built from: the caller's config with the harness's agent as the inner agent and the kind
    of `train_env` as the synthetic env type -- a VirtualEnv (0), a RewardEnv (1), or the real env itself = a RewardEnv of type 0
    (envs/reward_env.py:80-81: the real reward passes through; its dummy network is never evaluated)."""
    cfg = copy.deepcopy(config)
    cfg["agents"]["gtn"] = dict(cfg["agents"].get("gtn", {}), agent_name=agent_name)
    if train_env.is_virtual_env():
        cfg["agents"]["gtn"]["synthetic_env_type"] = 0
        return cfg, train_env.env.flat_params()
    cfg["agents"]["gtn"]["synthetic_env_type"] = 1
    if isinstance(train_env.env, RewardEnv):
        return cfg, train_env.env.flat_params()
    e = cfg["envs"][cfg["env_name"]]
    e["reward_env_type"] = 0
    # RewardEnv.build_reward_net for type 0 (envs/reward_env.py:44-53): a 1-input dummy MLP -- the launch stages it and never evaluates it
    n = mlp_num_params(mlp_desc(1, int(e["hidden_size"]), int(e["hidden_layer"]), 1, e["activation_fn"]))
    return cfg, torch.zeros(n, dtype=torch.float32, device=HipNesEngine().device)


def train_test_agents(train_env, test_env, config, agents_num, agent_name="DDQN_vary", train_episodes=1000, vary_hp=True, seed=0, replay=None,
                      model_index=0):
    """Returns (reward_list, train_steps_needed, episodes_needed) like the reference: reward_list[i] = the i-th agent's list of
    real-env test returns (BaseAgent.test), train_steps_needed[i] = [sum(episode_length)], episodes_needed[i] = [len(reward_train)].

    agent_name / train_episodes select the sibling script (HARNESS_AGENTS); vary_hp=False keeps the base hyper-parameters (the
    `_vary` agent with vary_hp off IS its base agent, DDQN_vary.py:16-21).  `seed` / `model_index` key the agents' counter RNG streams
    (the reference draws from the process-global generators).  replay = dict(hp=[...], agent_init=[...], tapes={name: [per-agent array]}):
    the recorded draws of a reference run, replayed in tape mode (parity tests); the returned dict `train_test_agents.last` holds the
    per-agent training lists (reward_train, episode_length) of the last call."""
    return _launch([train_env], test_env, config, agents_num, agent_name, train_episodes, vary_hp, seed, replay, [model_index])[0]


def train_test_agents_models(train_envs, test_env, config, agents_num, agent_name="DDQN_vary", train_episodes=1000, vary_hp=True, seed=0,
                             model_indices=None):
    """The model loop of experiments/syn_env_run_vary_hp.py:47-117 as ONE fused launch: `agents_num` agents on every env of `train_envs`
    (loaded SE checkpoints of one experiment -- same shapes --, or the real env repeated: mode 0) = len(train_envs) * agents_num chains, chain
    (m, i) reading model m's weights.  Returns [train_test_agents(train_envs[m], ..., model_index=model_indices[m]) for m], bit for bit
    (tests/test_run_vary_hp.py) -- where the reference spreads the models over a multiprocessing pool (:66-72,102-110), here they fill the GPU."""
    if model_indices is None:
        model_indices = list(range(len(train_envs)))
    return _launch(list(train_envs), test_env, config, agents_num, agent_name, train_episodes, vary_hp, seed, None, list(model_indices))


def _launch(train_envs, test_env, config, agents_num, agent_name, train_episodes, vary_hp, seed, replay, model_indices, settings=None):
    key = agent_name.lower()
    if key not in HARNESS_AGENTS:
        raise NotImplementedError("train_test_agents: agent '%s' (the harness scripts train %s)" % (agent_name, sorted(HARNESS_AGENTS)))
    if test_env.is_virtual_env():
        raise ValueError("test_env must be the real environment")
    apply_comparability_settings(config, agent_name, train_episodes, **(settings or {}))
    section, vary_section = HARNESS_AGENTS[key]
    if not vary_hp:
        config['agents'][vary_section]['vary_hp'] = False
    M, n_ag = len(train_envs), int(agents_num)
    cfg, theta = _task_config(train_envs[0], config, agent_name)
    engine = HipNesEngine()
    task = tasks.select_task(cfg, engine, train_envs[0], test_mode=1)
    dev = engine.device
    chains = M * n_ag
    if replay is not None:
        assert M == 1
        task.cfg.rng_mode = 1
        if hasattr(task, "fixed_hp"):
            task.fixed_hp = list(replay["hp"])
    inner = task.make_inner(chains, want_episode_stats=True)
    keys = np.concatenate([chain_keys(int(seed), int(mi), np.arange(n_ag), np.zeros(n_ag, np.int64)) for mi in model_indices])
    keys_t = torch.from_numpy(keys.view(np.int64)).to(dev)
    if M == 1:
        # one model: its weights are theta itself, sign 0 (the unperturbed checkpoint)
        worker = torch.zeros(chains, dtype=torch.int32, device=dev)
        sign = torch.zeros(chains, dtype=torch.float32, device=dev)
        eps = torch.zeros((1, theta.numel()), dtype=torch.float32, device=dev)
    else:
        # several models: chain (m, i) reads 0 + 1 * weights[m] (exact; a stored -0.0 becomes +0.0, which no sum downstream can tell apart)
        thetas = [theta] + [_task_config(e, config, agent_name)[1] for e in train_envs[1:]]
        if any(t.numel() != theta.numel() for t in thetas) or any(e.is_virtual_env() != train_envs[0].is_virtual_env() for e in train_envs):
            raise ValueError("train_test_agents_models: the models of one launch must have the same shapes")
        eps = torch.stack([t.to(device=dev, dtype=torch.float32) for t in thetas])
        theta = torch.zeros_like(eps[0])
        worker = torch.arange(chains, dtype=torch.int32, device=dev) // n_ag
        sign = torch.ones(chains, dtype=torch.float32, device=dev)
    # More chains than compute units (one workgroup per CU): the hardware hands the waiting workgroups out in launch order, so the expensive
    # draws go first (longest-processing-time-first; the chains are independent and keyed, the order changes nothing but the makespan).
    # Results are returned in (model, agent) order; `last["order"][k]` = the (model, agent) index that ran as chain k of `last["inner"]`.
    order = None
    if replay is None and chains > LPT_MIN_CHAINS and getattr(task, "fixed_hp", 0) is None:
        hp_pre = task.draw_hp(keys)
        cost = np.array([(h["hidden_layer"] + 1) * (20000.0 + h["batch_size"] * h["hidden_size"] * (h["hidden_size"] if h["hidden_layer"] > 1 else 8) / 8.0)
                         for h in hp_pre])
        order = np.argsort(-cost, kind="stable")
        idx = torch.from_numpy(order.copy()).to(dev)
        keys_t, worker, sign = keys_t[idx], worker[idx], sign[idx]
    agent_init = None
    if task.needs_agent_init():
        rows = []
        for mi in model_indices:
            g = torch.Generator(device=dev)
            g.manual_seed(int(seed) + 1000003 * int(mi))
            rows.append(fresh_agent_init(task.agent_bounds, n_ag, g, dev))
        agent_init = torch.cat(rows)
    if replay is None:
        task.scores(inner, theta, eps, worker, sign, keys_t, agent_init)
    else:
        _replay_launch(task, inner, theta, eps, worker, sign, keys_t, agent_init, replay)
    engine.check_status(inner)
    stats = inner.stats.cpu().numpy()
    finals = inner.final_returns.cpu().numpy()
    ep_mean, ep_len = inner.episode_test_mean.cpu().numpy(), inner.episode_len.cpu().numpy()
    hp_last = getattr(task, "last_hp", None)
    if order is not None:
        inv = np.argsort(order)
        stats, finals, ep_mean, ep_len = stats[inv], finals[inv], ep_mean[inv], ep_len[inv]
        hp_last = [hp_last[k] for k in inv] if hp_last is not None else None
    reward_list = [finals[i].tolist() for i in range(chains)]
    train_steps_needed = [[int(ep_len[i, :int(stats[i, 0])].sum())] for i in range(chains)]
    episodes_needed = [[int(stats[i, 0])] for i in range(chains)]
    train_test_agents.last = dict(reward_train=[ep_mean[i, :int(stats[i, 0])].tolist() for i in range(chains)],
                                  episode_length=[ep_len[i, :int(stats[i, 0])].tolist() for i in range(chains)],
                                  hp=hp_last, inner=inner, task=task, keys=keys, order=order)
    return [(reward_list[m * n_ag:(m + 1) * n_ag], train_steps_needed[m * n_ag:(m + 1) * n_ag], episodes_needed[m * n_ag:(m + 1) * n_ag])
            for m in range(M)]


def train_test_agents_correlation(train_env, test_env, config, agents_num, repeats=100, train_episodes=1000, seed=0, model_index=0):
    """The `train_test_agents` of the *_correlation scripts (experiments/syn_env_evaluate_cartpole_vary_hp_2_correlation.py:25-87): for each of
    `agents_num` drawn DDQN configurations, `repeats` (100) fresh DDQN agents with THAT configuration are trained on `train_env` and as many on
    the real env (`test_env`), each tested 100 episodes on the real env; early_out_num 1000 (:43).  Returns the three dicts of the reference:
    {"config", "synthetic": [...], "real": [...]} for the test returns, [sum(episode_length)] and [len(reward_train)] -- entries in the
    reference's order (configuration by configuration, its `repeats` agents).  Here a configuration's `repeats` agents share their shapes, so they
    are the chains of ONE launch of the fixed-shape kernels (the register-resident one where the drawn net fits it): two launches per
    configuration."""
    from ..agents import vary
    settings = dict(test_episodes=100, early_out_num=1000)
    apply_comparability_settings(config, "DDQN_vary", train_episodes, **settings)
    out = {"synthetic": ([], [], []), "real": ([], [], [])}
    drawn = []
    for i in range(int(agents_num)):
        key = chain_keys(int(seed), int(model_index), np.array([i]), np.array([1], np.int64))[0]           # kind 1: the configuration draws
        hp = vary.vary_hyperparameters(config["agents"]["ddqn"], vary.chain_units(key))
        drawn.append(hp)
        cfg_i = copy.deepcopy(config)
        cfg_i["agents"]["ddqn"].update(hp)
        for name, env in (("synthetic", train_env), ("real", test_env)):
            # agent j of configuration i: DDQN(config_varied) (:49,60) = the `_vary` agent with vary_hp off on the varied section
            r, t, e = _launch([env], test_env, copy.deepcopy(cfg_i), int(repeats), "DDQN_vary", train_episodes, False, int(seed) + 7919 * (i + 1),
                              None, [model_index], settings=settings)[0]
            out[name][0].extend(r)
            out[name][1].extend(t)
            out[name][2].extend(e)
    train_test_agents_correlation.last = dict(hp=drawn)
    return tuple({"config": config, "synthetic": out["synthetic"][k], "real": out["real"][k]} for k in range(3))


# experiments/syn_env_evaluate_cartpole_vary_hp_2_eval_generalization_gap.py:31,39-50: vary_hp off and these DDQN hyper-parameters (the optimised
# CartPole agent of default_config_cartpole_syn_env.yaml = BASELINE configs[1]'s 4-57-2 tanh net, batch 199)
GENERALIZATION_GAP_HP = dict(batch_size=199, gamma=0.988, lr=0.000304, tau=0.00848, eps_init=0.809, eps_min=0.0371, eps_decay=0.961,
                             same_action_num=1, activation_fn="tanh", hidden_size=57, hidden_layer=1)


def train_test_agents_generalization_gap(train_env, test_env, config, agents_num, seed=0, model_index=0):
    """The `train_test_agents` of the *_eval_generalization_gap script: the same function with vary_hp OFF and the fixed hyper-parameters above --
    every agent has the headline kernel's shape, so the launch runs on the register-resident kernel (test_mode 1)."""
    config['agents']['ddqn'].update(GENERALIZATION_GAP_HP)
    return train_test_agents(train_env, test_env, config, agents_num, vary_hp=False, seed=seed, model_index=model_index)


def _generalization_gap_models(train_envs, test_env, config, agents_num, model_indices=None, seed=0):
    config['agents']['ddqn'].update(GENERALIZATION_GAP_HP)
    return train_test_agents_models(train_envs, test_env, config, agents_num, vary_hp=False, seed=seed, model_indices=model_indices)


train_test_agents_generalization_gap.fused = _generalization_gap_models
train_test_agents.last = None
train_test_agents.fused = train_test_agents_models       # run_vary_hp (syn_env_run_vary_hp.py) takes the one-launch path when it finds this


def _pad_rows(rows, dtype, width=None):
    """[chains, max_len(, width)] device tensor of per-agent tapes of different lengths (a chain never reads past its own run)."""
    rows = [np.asarray(r, dtype).reshape(-1) if width is None else np.asarray(r, dtype).reshape(-1, width) for r in rows]
    n = max(1, max(r.shape[0] for r in rows))
    out = np.zeros((len(rows), n) + (() if width is None else (width,)), dtype)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return torch.from_numpy(out).cuda()


def _replay_launch(task, inner, theta, eps, worker, sign, keys_t, agent_init, replay):
    """One tape-mode launch with the recorded hyper-parameters, fresh agents and RNG draws of a reference run."""
    t = replay["tapes"]
    if "gumbel_act" in t:
        # the TD3-discrete loop's tapes (agents/TD3_discrete_vary.py: Gaussian action / policy noise and the Gumbel draws, one row per draw)
        A = int(np.asarray(t["act_noise"][0]).reshape(len(t["act_noise"][0]), -1).shape[1]) if len(t["act_noise"][0]) else 2
        tapes = dict(rand_action=_pad_rows(t["rand_action"], np.int32), replay_idx=_pad_rows(t["replay_idx"], np.int32),
                     train_reset=_pad_rows(t["train_reset"], np.float64, 4), test_reset=_pad_rows(t["test_reset"], np.float64, 4))
        for k in ("act_noise", "test_noise", "policy_noise", "gumbel_act", "gumbel_test", "gumbel_target", "gumbel_actor"):
            tapes[k] = _pad_rows(t[k], np.float32, A)
    else:
        tapes = dict(eps_uniform=_pad_rows(t["eps_uniform"], np.float64), rand_action=_pad_rows(t["rand_action"], np.int32),
                     replay_idx=_pad_rows(t["replay_idx"], np.int32), train_reset=_pad_rows(t["train_reset"], np.float64, 4),
                     test_reset=_pad_rows(t["test_reset"], np.float64, 4))
    if getattr(inner, "vary", False):
        hp = task.last_hp = task.draw_hp(None)
        inner.set_hp([h["lr"] for h in hp], [h["batch_size"] for h in hp], [h["hidden_size"] for h in hp], [h["hidden_layer"] for h in hp])
        init = torch.zeros_like(inner.agent_init)
    else:
        init = torch.zeros_like(agent_init)
    for i, a in enumerate(replay["agent_init"]):
        a = torch.from_numpy(np.ascontiguousarray(a, np.float32))
        init[i, :a.numel()] = a.to(init.device)
    inner.run(theta, eps, worker, sign, init, tapes=tapes)
