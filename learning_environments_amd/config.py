"""Reference YAML config dict -> kernel configuration structs."""
import ctypes as C

from . import _lib

ENV_DIMS = {"CartPole-v0": (4, 2), "Acrobot-v1": (6, 3), "HalfCheetah-v3": (17, 6), "MountainCar-v0": (2, 3), "Pendulum-v0": (3, 1), "MountainCarContinuous-v0": (2, 1)}
# continuous real envs of the TD3 path: EnvWrapper.get_max_action (envs/env_wrapper.py:106-110)
TD3_MAX_ACTION = {"HalfCheetah-v3": 1.0, "Pendulum-v0": 2.0, "MountainCarContinuous-v0": 1.0}


def _launch_knobs(cfg, config):
    """Launch knobs of this implementation (no counterpart in the reference; absent keys = automatic): `team_size` (workgroups per
    chain: 0 automatic, 1 never a team, G at most G) and `kernel_variant` (_lib.VARIANT_* bits) in the `gtn` section."""
    gtn = config["agents"].get("gtn", {})
    cfg.team_size = int(gtn.get("team_size", 0))
    cfg.kernel_variant = int(gtn.get("kernel_variant", 0))


def ddqn_cfg_from_config(config, rng_mode=_lib.RNG_COUNTER, grad_chunk=0, **overrides):
    """Fields read at reference agents/DDQN.py:15-38, agents/base_agent.py:9-26, envs/env_factory.py:45-59.
    grad_chunk=0 picks the smallest micro-chunk (>= ceil(batch/16)) whose LDS footprint fits one CU; it stays 0 (one
    sequential batch gradient) for DuelingDDQN and for DDQN nets that only the GEMM-tiled kernel takes."""
    env_name = config["env_name"]
    if env_name not in ENV_DIMS:
        raise NotImplementedError("real env '%s' has no device implementation yet" % env_name)
    e = config["envs"][env_name]
    agent_key = config["agents"]["gtn"]["agent_name"].lower() if "gtn" in config["agents"] else "ddqn"
    if agent_key.endswith("_vary"):                  # DDQN_vary / DuelingDDQN_vary read the base agent's section (DDQN_vary.py:14)
        agent_key = agent_key[:-5]
    icm = agent_key.endswith("_icm")                 # select_agent: "ddqn_icm" / "duelingddqn_icm" = the agent with icm=True
    if icm:
        agent_key = agent_key[:-4]
    if agent_key not in ("ddqn", "duelingddqn"):
        raise NotImplementedError("ddqn_cfg_from_config: agent '%s'" % agent_key)
    a = config["agents"][agent_key]
    dueling = agent_key == "duelingddqn"
    S, A = ENV_DIMS[env_name]

    def val(v):  # env_factory.py:54-58: list-valued entries -> float(value[1])
        return float(v[1]) if isinstance(v, list) else v

    cfg = _lib.DdqnCfg(env_id=_lib.ENV[env_name], state_dim=S, num_actions=A, max_steps=int(val(e["max_steps"])),
                       se_hidden=int(val(e["hidden_size"])), se_layers=int(val(e["hidden_layer"])),
                       se_act=_lib.ACT[e["activation_fn"]], se_prelu=0.25,
                       # build_nn_from_config (models/model_utils.py:33-37) adds `hidden_layer - 1` extra blocks: 0 builds the same
                       # network as 1 (the *_vary agents sample hidden_layer in {L-1, L, L+1})
                       q_hidden=int(a["hidden_size"]), q_layers=max(1, int(a["hidden_layer"])), q_act=_lib.ACT[a["activation_fn"]],
                       q_prelu=0.25, batch_size=int(a["batch_size"]), rb_size=int(a["rb_size"]),
                       train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]),
                       init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]),
                       grad_chunk=int(grad_chunk), rng_mode=int(rng_mode), agent_kind=1 if dueling else 0,
                       feature_dim=int(a.get("feature_dim", 0)) if dueling else 0,
                       solved_reward=float(val(e["solved_reward"])),
                       gamma=float(a["gamma"]), lr=float(a["lr"]), tau=float(a["tau"]), eps_init=float(a["eps_init"]),
                       eps_min=float(a["eps_min"]), eps_decay=float(a["eps_decay"]),
                       adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                       step_budget=int(a.get("step_budget", 0)))      # env-step stand-in for time_remaining (base_agent.py:30-47)
    cfg.same_action_num = int(a["same_action_num"])    # env steps per chosen action (base_agent.py:104,194); > 1: GEMM-tiled kernel
    # base_agent.py:24: read by env_solved when BaseAgent.train runs WITHOUT a test env (cfg.test_mode 1, the evaluation harness)
    cfg.early_out_virtual_diff = float(a.get("early_out_virtual_diff", 0.0))
    # use_layer_norm: ONE shared nn.LayerNorm behind hidden Linear 2..L of the Q-net / the feature stream (model_utils.py:22-37); a net with
    # one hidden layer has no position for it (and no parameters: the module is created but never registered)
    cfg.q_layer_norm = 1 if a.get("use_layer_norm", False) else 0
    # the ENV section's use_layer_norm: the synthetic env's three nets normalise behind their hidden Linear 2..L too (virtual_env.py:16-33 builds
    # them with build_nn_from_config).  NES perturbs and updates nn.Linear modules only (GTN_worker.py:156-175, GTN_master.py:281-296), so the
    # module keeps its constructor weight 1 / bias 0 on every worker and theta stays the Linear parameters: the kernel normalises without
    # parameters
    cfg.se_layer_norm = 1 if e.get("use_layer_norm", False) else 0      # RewardEnv mode: the reward net's
    _launch_knobs(cfg, config)
    if icm:                                          # config section `icm` (agents/DDQN.py:43-49)
        ic = config["agents"]["icm"]
        cfg.icm_enabled, cfg.icm_feature_dim, cfg.icm_hidden = 1, int(ic["feature_dim"]), int(ic["hidden_size"])
        cfg.icm_lr, cfg.icm_beta, cfg.icm_eta = float(ic["lr"]), float(ic["beta"]), float(ic["eta"])
    if "gtn" in config["agents"] and int(config["agents"]["gtn"].get("synthetic_env_type", 0)) == 1:
        # RewardEnv over the real env (default_config_cartpole_reward_env.yaml): the `envs` section describes the reward network
        cfg.synthetic_env_type, cfg.reward_env_type = 1, int(val(e["reward_env_type"]))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    if cfg.grad_chunk == 0 and cfg.agent_kind == 0 and not cfg.icm_enabled:
        cfg.grad_chunk = pick_grad_chunk(cfg)          # DuelingDDQN / ICM agents: one sequential chunk (grad_chunk stays 0)
    return cfg


def icm_layer_dims(cfg):
    """[(fan_in, fan_out), ...] of ICMModel's nn.Linear layers in state-dict order (models/icm_baseline.py:42-78)."""
    S, F, H = cfg.state_dim, cfg.icm_feature_dim, cfg.icm_hidden
    discrete = hasattr(cfg, "num_actions")           # a TD3 cfg has action_dim: continuous actions, the vector itself is the input
    A = cfg.num_actions if discrete else cfg.action_dim
    Ai = 1 if (discrete and A == 2) else A
    C = F + Ai
    return ([(S, H), (H, H), (H, F)] + [(2 * F, H), (H, H), (H, Ai)] + [(C, H), (H, H), (H, F)] + [(C, F), (C, F)] * 4
            + [(F, H), (H, F)])


def agent_layer_dims(cfg):
    """[(fan_in, fan_out), ...] of the agent's nn.Linear layers in flat state-dict order (for fresh-agent initialisation)."""
    S, A, H, L = cfg.state_dim, cfg.num_actions, cfg.q_hidden, cfg.q_layers
    if cfg.agent_kind == 1:
        F = cfg.feature_dim
        feat = [(S, H)] + [(H, H)] * (L - 1) + [(H, F)]
        return feat + [(F, F), (F, 1)] + [(F, F), (F, A)]
    return [(S, H)] + [(H, H)] * (L - 1) + [(H, A)]


def agent_layer_norm_slice(cfg):
    """(offset, H) of the shared LayerNorm's weight | bias block in the agent's flat parameter vector (behind the second Linear:
    Module.parameters() order), or None."""
    if not getattr(cfg, "q_layer_norm", 0) or cfg.q_layers < 2:
        return None
    S, H = cfg.state_dim, cfg.q_hidden
    return (S * H + H) + (H * H + H), H


def pick_grad_chunk(cfg):
    L = _lib.lib()
    B = cfg.batch_size
    chunk = (B + 15) // 16
    probe = _lib.DdqnCfg.from_buffer_copy(cfg)
    while chunk <= B:
        probe.grad_chunk = chunk
        if L.lenv_ddqn_se_lds_bytes(C.byref(probe)) > 0:
            return chunk
        chunk += 1
    # Critic_DQN shapes the register-resident kernel does not take (hidden_layer >= 2, wide layers): the GEMM-tiled kernel
    # trains them with one sequential batch gradient (grad_chunk 0) -- engine.InnerLoop routes on the same probe
    probe.grad_chunk = 0
    if L.lenv_dueling_num_params(C.byref(probe)) > 0:
        return 0
    raise NotImplementedError("DDQN/SE shapes fit neither fused kernel")


TABULAR_AGENTS = ("ql", "ql_cb", "sarsa", "sarsa_cb")     # agents/agent_utils.py:57-64


def ql_cfg_from_config(config, tables, rng_mode=_lib.RNG_COUNTER, **overrides):
    """QL agent + RewardEnv on a gridworld.  Fields read at reference agents/QL.py:13-27, agents/base_agent.py:9-26,
    envs/reward_env.py:8-27, envs/env_factory.py:45-59."""
    env_name = config["env_name"]
    e = config["envs"][env_name]
    name = config["agents"]["gtn"]["agent_name"].lower() if "gtn" in config["agents"] else "ql"
    if name not in TABULAR_AGENTS:
        raise NotImplementedError("ql_cfg_from_config: agent '%s'" % name)
    a = config["agents"]["sarsa" if name.startswith("sarsa") else "ql"]      # SARSA reads its own section (SARSA.py:14-18)
    if int(a["rb_size"]) != 1:
        raise NotImplementedError("tabular agents with rb_size != 1 (the reference configs keep the single latest transition)")

    def val(v):
        return float(v[1]) if isinstance(v, list) else v

    cfg = _lib.QlCfg(agent_kind=1 if name.startswith("sarsa") else 0, count_based=1 if name.endswith("_cb") else 0,
                     beta=float(a.get("beta", 0.0)), n_states=tables["n_states"], n_actions=tables["n_actions"], start_state=tables["start_state"],
                     max_steps=int(val(e["max_steps"])), rn_hidden=int(val(e["hidden_size"])), rn_layers=int(val(e["hidden_layer"])),
                     rn_act=_lib.ACT[e["activation_fn"]], rn_prelu=0.25, reward_env_type=int(val(e["reward_env_type"])),
                     train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]),
                     init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]),
                     batch_size=int(a["batch_size"]), rng_mode=int(rng_mode), solved_reward=float(val(e["solved_reward"])),
                     alpha=float(a["alpha"]), gamma=float(a["gamma"]), eps_init=float(a["eps_init"]),
                     eps_min=float(a["eps_min"]), eps_decay=float(a["eps_decay"]), step_budget=int(a.get("step_budget", 0)),
                       early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    cfg.same_action_num = int(a["same_action_num"])
    cfg.rn_layer_norm = 1 if e.get("use_layer_norm", False) else 0     # the reward net's own LayerNorm (never perturbed: see ddqn_cfg_from_config)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def td3_cfg_from_config(config, rng_mode=_lib.RNG_COUNTER, **overrides):
    """TD3 agent + RewardEnv / VirtualEnv on the HalfCheetah stand-in or Pendulum-v0.  Fields read at reference agents/TD3.py:13-29,
    agents/base_agent.py:9-26, envs/reward_env.py:8-27, envs/env_wrapper.py:106-110 (max_action)."""
    env_name = config["env_name"]
    if env_name not in TD3_MAX_ACTION:
        raise NotImplementedError("TD3 fused kernel: real env '%s'" % env_name)
    S, A = ENV_DIMS[env_name]
    e = config["envs"][env_name]
    a = config["agents"]["td3"]

    def val(v):
        return float(v[1]) if isinstance(v, list) else v

    cfg = _lib.Td3Cfg(env_id=_lib.ENV[env_name], state_dim=S, action_dim=A, max_steps=int(val(e["max_steps"])),
                      rn_hidden=int(val(e["hidden_size"])), rn_layers=int(val(e["hidden_layer"])), rn_act=_lib.ACT[e["activation_fn"]],
                      rn_prelu=0.25, reward_env_type=int(val(e["reward_env_type"])), info_dim=int(val(e.get("info_dim", 0))),
                      hidden=int(a["hidden_size"]),
                      layers=max(1, int(a["hidden_layer"])), act=_lib.ACT[a["activation_fn"]], prelu=0.25, batch_size=int(a["batch_size"]),
                      rb_size=int(a["rb_size"]), train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]),
                      init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]),
                      policy_delay=int(a["policy_delay"]), rng_mode=int(rng_mode), solved_reward=float(val(e["solved_reward"])),
                      gamma=float(a["gamma"]), lr=float(a["lr"]), tau=float(a["tau"]), action_std=float(a["action_std"]),
                      policy_std=float(a["policy_std"]), policy_std_clip=float(a["policy_std_clip"]), max_action=TD3_MAX_ACTION[env_name],
                      adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, step_budget=int(a.get("step_budget", 0)),
                       early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    if "gtn" in config["agents"] and int(config["agents"]["gtn"].get("synthetic_env_type", 1)) == 0:
        cfg.virtual_env = 1                           # VirtualEnv (default_config_halfcheetah.yaml): `envs` describes the three SE nets
    cfg.same_action_num = int(a["same_action_num"])   # env steps per chosen action (the MountainCarContinuous configs ship 2)
    # use_layer_norm: one shared nn.LayerNorm per net (actor, critic_1, critic_2) behind its hidden Linear 2..L (model_utils.py:22-37)
    cfg.use_layer_norm = 1 if a.get("use_layer_norm", False) else 0
    # the ENV section's use_layer_norm: the reward net / the three SE nets normalise too; theta stays Linear-only (see ddqn_cfg_from_config)
    cfg.rn_layer_norm = 1 if e.get("use_layer_norm", False) else 0
    _launch_knobs(cfg, config)
    name = config["agents"]["gtn"]["agent_name"].lower() if "gtn" in config["agents"] else "td3"
    if name.replace("_vary", "").endswith("_icm"):   # select_agent "td3_icm" / "td3_icm_vary": TD3(icm=True), agents/TD3.py:44-60
        ic = config["agents"]["icm"]
        cfg.icm_enabled, cfg.icm_feature_dim, cfg.icm_hidden = 1, int(ic["feature_dim"]), int(ic["hidden_size"])
        cfg.icm_lr, cfg.icm_beta, cfg.icm_eta = float(ic["lr"]), float(ic["beta"]), float(ic["eta"])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


TD3_DISCRETE_ENVS = ("CartPole-v0", "Acrobot-v1", "MountainCar-v0")      # discrete-action real envs of the TD3_discrete_vary kernel


def td3d_cfg_from_config(config, rng_mode=_lib.RNG_COUNTER, **overrides):
    """TD3_discrete_vary + VirtualEnv on CartPole-v0 / Acrobot-v1 / MountainCar-v0.  Fields read at reference
    agents/TD3_discrete_vary.py:30-42, models/actor_critic.py:27-31 (gumbel_softmax_temp / _hard), models/model_utils.py:5-29
    (use_layer_norm), agents/base_agent.py:9-26, envs/env_wrapper.py:106-110 (max_action 1 for these envs)."""
    env_name = config["env_name"]
    if env_name not in TD3_DISCRETE_ENVS:
        raise NotImplementedError("TD3_discrete_vary fused kernel: real env '%s'" % env_name)
    if int(config["agents"]["gtn"].get("synthetic_env_type", 0)) != 0:
        raise NotImplementedError("TD3_discrete_vary trains on a VirtualEnv here (synthetic_env_type 0)")
    S, A = ENV_DIMS[env_name]
    e = config["envs"][env_name]
    a = config["agents"]["td3_discrete_vary"]
    if a["same_action_num"] != 1:
        raise NotImplementedError("same_action_num != 1")

    def val(v):
        return float(v[1]) if isinstance(v, list) else v

    cfg = _lib.Td3dCfg(env_id=_lib.ENV[env_name], state_dim=S, action_dim=A, max_steps=int(val(e["max_steps"])),
                       se_hidden=int(val(e["hidden_size"])), se_layers=int(val(e["hidden_layer"])), se_act=_lib.ACT[e["activation_fn"]],
                       se_prelu=0.25, hidden=int(a["hidden_size"]), layers=max(1, int(a["hidden_layer"])), act=_lib.ACT[a["activation_fn"]],
                       prelu=0.25, use_layer_norm=1 if a.get("use_layer_norm", False) else 0,
                       se_layer_norm=1 if e.get("use_layer_norm", False) else 0,      # the SE nets' own LayerNorm: theta stays Linear-only
                       gumbel_hard=1 if a["gumbel_softmax_hard"] else 0, batch_size=int(a["batch_size"]), rb_size=int(a["rb_size"]),
                       train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]), init_episodes=int(a["init_episodes"]),
                       early_out_num=int(a["early_out_num"]), policy_delay=int(a["policy_delay"]), rng_mode=int(rng_mode),
                       solved_reward=float(val(e["solved_reward"])), gamma=float(a["gamma"]), lr=float(a["lr"]), tau=float(a["tau"]),
                       action_std=float(a["action_std"]), policy_std=float(a["policy_std"]), policy_std_clip=float(a["policy_std_clip"]),
                       max_action=1.0, gumbel_temp=float(a["gumbel_softmax_temp"]), adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                       step_budget=int(a.get("step_budget", 0)),
                       early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def td3_layer_norm_slices(cfg):
    """[(offset, H)] of the three LayerNorm blocks (actor, critic_1, critic_2) in TD3's flat parameter vector, or []."""
    if not getattr(cfg, "use_layer_norm", 0) or cfg.layers < 2:
        return []
    H, L, S, A = cfg.hidden, cfg.layers, cfg.state_dim, cfg.action_dim

    def net(n_in, n_out):
        return (n_in * H + H) + (L - 1) * (H * H + H) + 2 * H + (n_out * H + n_out)
    pa, pc = net(S, A), net(S + A, 1)
    return [((S * H + H) + (H * H + H), H), (pa + ((S + A) * H + H) + (H * H + H), H), (pa + pc + ((S + A) * H + H) + (H * H + H), H)]


def td3_layer_dims(cfg):
    H, L = cfg.hidden, cfg.layers
    actor = [(cfg.state_dim, H)] + [(H, H)] * (L - 1) + [(H, cfg.action_dim)]
    critic = [(cfg.state_dim + cfg.action_dim, H)] + [(H, H)] * (L - 1) + [(H, 1)]
    return actor + critic + critic
