"""Hyper-parameter sets of the BASELINE configurations as plain dicts (same keys as the reference's YAML files, so
`yaml.safe_load(open('default_config_cartpole_syn_env.yaml'))` from a reference checkout is interchangeable)."""
import copy


def cartpole_syn_env_ddqn(num_workers=64, max_iterations=200):
    """BASELINE configs 1/2: CartPole-v0 SE + DDQN (values = the published hyper-parameters of
    default_config_cartpole_syn_env.yaml: gtn section :5-26, ddqn :28-46, env :124-132)."""
    return copy.deepcopy({
        "env_name": "CartPole-v0", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.0124, "step_size": 0.148, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 600, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 3, "quit_when_solved": False, "synthetic_env_type": 0,
                    "unsolved_weight": 10000, "agent_name": "DDQN"},
            "ddqn": {"train_episodes": 1000, "test_episodes": 10, "init_episodes": 1, "batch_size": 199, "gamma": 0.988,
                     "lr": 0.000304, "tau": 0.00848, "eps_init": 0.809, "eps_min": 0.0371, "eps_decay": 0.961,
                     "rb_size": 100000, "same_action_num": 1, "activation_fn": "tanh", "hidden_size": 57,
                     "hidden_layer": 1, "print_rate": 10, "early_out_num": 10, "early_out_virtual_diff": 0.01},
        },
        "envs": {"CartPole-v0": {"solved_reward": 195.0, "max_steps": 200, "activation_fn": "leakyrelu", "hidden_size": 83,
                                 "hidden_layer": 1, "info_dim": 0, "reward_env_type": 0}},
    })


def acrobot_syn_env_duelingddqn(num_workers=256, max_iterations=50):
    """BASELINE config 3: Acrobot-v1 SE + DuelingDDQN (values = the published hyper-parameters of default_config_acrobot.yaml:
    gtn :5-26 with agent_name DuelingDDQN, duelingddqn :61-80, env :131-139)."""
    return copy.deepcopy({
        "env_name": "Acrobot-v1", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.05, "step_size": 1.0, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 300, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 7, "quit_when_solved": True, "synthetic_env_type": 0,
                    "unsolved_weight": 10000, "agent_name": "DuelingDDQN"},
            "duelingddqn": {"train_episodes": 1000, "test_episodes": 10, "init_episodes": 10, "batch_size": 128, "gamma": 0.99,
                            "lr": 1e-3, "tau": 0.01, "eps_init": 1.0, "eps_min": 0.01, "eps_decay": 0.9, "rb_size": 100000,
                            "same_action_num": 1, "activation_fn": "relu", "hidden_size": 128, "hidden_layer": 2,
                            "feature_dim": 128, "print_rate": 1, "early_out_num": 10, "early_out_virtual_diff": 0.01},
        },
        "envs": {"Acrobot-v1": {"solved_reward": -100.0, "max_steps": 500, "activation_fn": "leakyrelu", "hidden_size": 128,
                                "hidden_layer": 1, "info_dim": 0, "reward_env_type": 0}},
    })


def acrobot_syn_env_ddqn(num_workers=256, max_iterations=50):
    """Acrobot-v1 SE + the DDQN section of default_config_acrobot.yaml (:30-48: Critic_DQN 6-128-128-3, relu, batch 128,
    init_episodes 1).  Its two-hidden-layer Q-net runs in the GEMM-tiled kernel's plain-DQN mode."""
    cfg = acrobot_syn_env_duelingddqn(num_workers, max_iterations)
    cfg["agents"]["gtn"]["agent_name"] = "DDQN"
    d = cfg["agents"].pop("duelingddqn")
    d.pop("feature_dim")
    d.update(init_episodes=1, print_rate=10)
    cfg["agents"]["ddqn"] = d
    return cfg


def mountaincar_syn_env_ddqn(num_workers=16, max_iterations=50):
    """MountainCar-v0 SE (2+1 -> 128 -> 2/1/1, leakyrelu) + DDQN 2-256-256-3: the published values of
    default_config_mountaincar.yaml (gtn :5-26, ddqn :30-48, env :52-59).  100 random init episodes fill the replay buffer; the
    two-hidden-layer Q-net runs in the GEMM-tiled kernel's plain-DQN mode."""
    cfg = acrobot_syn_env_ddqn(num_workers, max_iterations)
    cfg["env_name"] = "MountainCar-v0"
    cfg["agents"]["gtn"].update(noise_std=0.05, step_size=1.0, time_max=300, score_transform_type=7, unsolved_weight=10000,
                                agent_name="DDQN")
    cfg["agents"]["ddqn"].update(train_episodes=1000, test_episodes=10, init_episodes=100, batch_size=128, gamma=0.99, lr=1e-3,
                                 tau=0.01, eps_init=1.0, eps_min=0.01, eps_decay=0.99, rb_size=100000, same_action_num=1,
                                 activation_fn="relu", hidden_size=256, hidden_layer=2, print_rate=10, early_out_num=10,
                                 early_out_virtual_diff=0.01)
    cfg["envs"] = {"MountainCar-v0": {"solved_reward": -110.0, "max_steps": 200, "activation_fn": "leakyrelu", "hidden_size": 128,
                                      "hidden_layer": 1, "info_dim": 0, "reward_env_type": 0}}
    return cfg


def cartpole_reward_env_ddqn(num_workers=16, max_iterations=50):
    """CartPole-v0 RewardEnv (potential-shaped, type 2, PReLU reward net 4-64-1) + DDQN 4-64-2: the published values of
    default_config_cartpole_reward_env.yaml (gtn :5-26, ddqn :28-46, env :49-56).  synthetic_env_type 1: the agents train on the
    REAL CartPole with the learned reward."""
    return copy.deepcopy({
        "env_name": "CartPole-v0", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.1, "step_size": 0.5, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 3600, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 3, "quit_when_solved": True, "synthetic_env_type": 1,
                    "unsolved_weight": 100, "agent_name": "DDQN"},
            "ddqn": {"train_episodes": 100, "test_episodes": 1, "init_episodes": 1, "batch_size": 192, "gamma": 0.99,
                     "lr": 0.003, "tau": 0.01, "eps_init": 0.8, "eps_min": 0.03, "eps_decay": 0.95, "rb_size": 1000000,
                     "same_action_num": 1, "activation_fn": "leakyrelu", "hidden_size": 64, "hidden_layer": 1,
                     "print_rate": 10, "early_out_num": 10, "early_out_virtual_diff": 0.02},
        },
        "envs": {"CartPole-v0": {"solved_reward": 195.0, "max_steps": 200, "activation_fn": "prelu", "hidden_size": 64,
                                 "hidden_layer": 1, "info_dim": 0, "reward_env_type": 2}},
    })


def cliff_reward_env_ql(num_workers=128, max_iterations=50):
    """BASELINE config 4: Cliff gridworld RewardEnv (potential shaped, type 2) + tabular QL (values = the published
    hyper-parameters of default_config_gridworld_reward_env.yaml: gtn :5-26, ql :28-43, Cliff :126-133)."""
    return copy.deepcopy({
        "env_name": "Cliff", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.1, "step_size": 0.5, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 300, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 3, "quit_when_solved": True, "synthetic_env_type": 1, "unsolved_weight": 100,
                    "agent_name": "QL"},
            "ql": {"train_episodes": 100, "test_episodes": 1, "init_episodes": 0, "batch_size": 1, "alpha": 1.0, "gamma": 0.8,
                   "eps_init": 0.01, "eps_min": 0.01, "eps_decay": 0.0, "rb_size": 1, "same_action_num": 1, "beta": 0.005,
                   "print_rate": 100, "early_out_num": 10, "early_out_virtual_diff": 0.02},
        },
        "envs": {"Cliff": {"solved_reward": -20.0, "max_steps": 50, "activation_fn": "prelu", "hidden_size": 32,
                           "hidden_layer": 1, "info_dim": 0, "reward_env_type": 2}},
    })


def halfcheetah_reward_env_td3(num_workers=64, max_iterations=50):
    """BASELINE config 5: HalfCheetah-v3 RewardEnv (potential shaped, type 2) + TD3 (values = the published hyper-parameters of
    default_config_halfcheetah_reward_env.yaml: td3 :31-50, env :53-60; gtn as in the gridworld RN config).  The real env is
    the documented stand-in (tools/gen_cheetah_standin.py) on BOTH sides: MuJoCo cannot be installed."""
    return copy.deepcopy({
        "env_name": "HalfCheetah-v3", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.1, "step_size": 0.5, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 3600, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 3, "quit_when_solved": True, "synthetic_env_type": 1, "unsolved_weight": 100,
                    "agent_name": "TD3"},
            "td3": {"train_episodes": 100, "test_episodes": 1, "init_episodes": 20, "batch_size": 192, "gamma": 0.98, "lr": 0.003,
                    "tau": 0.01, "policy_delay": 1, "rb_size": 1000000, "same_action_num": 1, "activation_fn": "relu",
                    "hidden_size": 128, "hidden_layer": 2, "action_std": 0.05, "policy_std": 0.2, "policy_std_clip": 0.5,
                    "print_rate": 5, "early_out_num": 5, "early_out_virtual_diff": 0.02},
        },
        "envs": {"HalfCheetah-v3": {"solved_reward": 3000.0, "max_steps": 1000, "activation_fn": "prelu", "hidden_size": 128,
                                    "hidden_layer": 1, "info_dim": 4, "reward_env_type": 2}},
    })


def halfcheetah_syn_env_td3(num_workers=128, max_iterations=100):
    """HalfCheetah-v3 (stand-in) VirtualEnv + TD3: the published values of default_config_halfcheetah.yaml (gtn :6-27 with
    synthetic_env_type 0, td3 :29-48, env :79-86: three SE nets 23-128-128-128-{17,1,1}, relu).  The yaml's agent_name is
    td3_vary: wrap with with_vary()."""
    return copy.deepcopy({
        "env_name": "HalfCheetah-v3", "device": "cuda", "render_env": False,
        "agents": {
            "gtn": {"mode": "multi", "max_iterations": max_iterations, "num_threads_per_worker": 1,
                    "num_workers": num_workers, "noise_std": 0.1, "step_size": 1.0, "nes_step_size": False,
                    "mirrored_sampling": True, "num_grad_evals": 1, "grad_eval_type": "mean", "weight_decay": 0.0,
                    "time_mult": 3, "time_max": 360000, "time_sleep_master": 0.2, "time_sleep_worker": 2,
                    "score_transform_type": 7, "quit_when_solved": True, "synthetic_env_type": 0,
                    "unsolved_weight": 1, "agent_name": "td3"},
            "td3": {"train_episodes": 1000, "test_episodes": 10, "init_episodes": 20, "batch_size": 256, "gamma": 0.99,
                    "lr": 3e-4, "tau": 0.005, "policy_delay": 2, "rb_size": 1000000, "same_action_num": 1,
                    "activation_fn": "relu", "hidden_size": 128, "hidden_layer": 2, "action_std": 0.1, "policy_std": 0.2,
                    "policy_std_clip": 0.5, "print_rate": 10, "early_out_num": 50, "early_out_virtual_diff": 0.02},
        },
        "envs": {"HalfCheetah-v3": {"solved_reward": 3000.0, "max_steps": 1000, "activation_fn": "relu", "hidden_size": 128,
                                    "hidden_layer": 3, "info_dim": 4, "reward_env_type": 0}},
    })


def pendulum_syn_env_td3(num_workers=16, max_iterations=50):
    """Pendulum-v0 VirtualEnv (three SE nets 4-32-32-{3,1,1}, leakyrelu) + TD3 (max_action 2): the published values of
    default_config_pendulum.yaml (gtn :5-26, td3 :29-48, env :78-85).  The yaml's agent_name is td3_vary: wrap with with_vary()."""
    cfg = halfcheetah_syn_env_td3(num_workers, max_iterations)
    cfg["env_name"] = "Pendulum-v0"
    cfg["agents"]["gtn"].update(time_max=600, quit_when_solved=False)
    cfg["agents"]["td3"].update(train_episodes=70, lr=1e-3, tau=0.02, print_rate=1, early_out_num=5, early_out_virtual_diff=0.1)
    cfg["envs"] = {"Pendulum-v0": {"solved_reward": -300.0, "max_steps": 200, "activation_fn": "leakyrelu", "hidden_size": 32,
                                   "hidden_layer": 2, "info_dim": 0, "reward_env_type": 2}}
    return cfg


def pendulum_reward_env_td3(num_workers=16, max_iterations=20):
    """Pendulum-v0 RewardEnv (potential-shaped, type 2, PReLU reward net 3-128-128-1) + TD3: the published values of
    default_config_pendulum_reward_env.yaml (gtn :5-26, td3 :28-47, env :50-57)."""
    cfg = pendulum_syn_env_td3(num_workers, max_iterations)
    cfg["agents"]["gtn"].update(synthetic_env_type=1, noise_std=0.01, step_size=0.5, score_transform_type=3)
    cfg["agents"]["td3"].update(batch_size=192, gamma=0.98, lr=0.003, tau=0.03, policy_delay=1, activation_fn="leakyrelu",
                                action_std=0.05, policy_std=0.1, policy_std_clip=0.25, print_rate=5, early_out_virtual_diff=0.04)
    cfg["envs"]["Pendulum-v0"].update(activation_fn="prelu", hidden_size=128, hidden_layer=2, reward_env_type=2)
    return cfg


def cmc_syn_env_td3(num_workers=128, max_iterations=20):
    """MountainCarContinuous-v0 VirtualEnv (three SE nets 3-96-96-{2,1,1}, leakyrelu) + TD3 with same_action_num 2: the published
    values of default_config_cmc.yaml (gtn :5-26, td3 :28-47, env :78-85)."""
    cfg = halfcheetah_syn_env_td3(num_workers, max_iterations)
    cfg["env_name"] = "MountainCarContinuous-v0"
    cfg["agents"]["gtn"].update(num_threads_per_worker=2, time_max=1500, quit_when_solved=False)
    cfg["agents"]["td3"].update(train_episodes=500, test_episodes=1, init_episodes=50, same_action_num=2, early_out_num=5,
                                early_out_virtual_diff=0.1)
    cfg["envs"] = {"MountainCarContinuous-v0": {"solved_reward": 90.0, "max_steps": 999, "activation_fn": "leakyrelu", "hidden_size": 96,
                                                "hidden_layer": 2, "info_dim": 0, "reward_env_type": 0}}
    return cfg


def cmc_reward_env_td3(num_workers=16, max_iterations=50):
    """MountainCarContinuous-v0 RewardEnv (potential-shaped, type 2, tanh reward net 2-128-1) + TD3 with same_action_num 2: the
    published values of default_config_cmc_reward_env.yaml (gtn :5-26, td3 :28-47, env :50-57)."""
    cfg = cmc_syn_env_td3(num_workers, max_iterations)
    cfg["agents"]["gtn"].update(num_threads_per_worker=1, step_size=0.5, time_max=3600, score_transform_type=3, synthetic_env_type=1,
                                unsolved_weight=100)
    cfg["agents"]["td3"].update(train_episodes=100, batch_size=192, lr=0.003, tau=0.01, policy_delay=1, activation_fn="leakyrelu",
                                action_std=0.05, print_rate=5, early_out_virtual_diff=0.02)
    cfg["envs"]["MountainCarContinuous-v0"].update(activation_fn="tanh", hidden_size=128, hidden_layer=1, reward_env_type=2)
    return cfg


def _td3_discrete_section(**over):
    """The `td3_discrete_vary` section both syn-env YAMLs ship (default_config_cartpole_syn_env.yaml:79-102 =
    default_config_acrobot_syn_env.yaml:58-81): actor S-510-510-A / critics (S+A)-510-510-1, tanh, hard Gumbel softmax."""
    d = {"train_episodes": 1000, "test_episodes": 10, "init_episodes": 1, "batch_size": 122, "gamma": 0.9989, "lr": 0.0017496,
         "tau": 0.0724303, "policy_delay": 1, "rb_size": 1000000, "same_action_num": 1, "activation_fn": "tanh", "hidden_size": 510,
         "hidden_layer": 2, "action_std": 0.037275, "policy_std": 0.2225286, "policy_std_clip": 0.5, "print_rate": 1, "early_out_num": 1,
         "early_out_virtual_diff": 0.01, "gumbel_softmax_temp": 2.3076235, "gumbel_softmax_hard": True, "vary_hp": False}
    d.update(over)
    return d


def cartpole_syn_env_td3_discrete(num_workers=16, max_iterations=50, **agent_over):
    """CartPole-v0 SE + TD3_discrete_vary (select_agent "td3_discrete_vary"): the GTN / env sections of
    default_config_cartpole_syn_env.yaml with its td3_discrete_vary agent section; agent_over e.g. use_layer_norm=True
    (default_config_cartpole.yaml:106-130 `td3_discrete_vary_layer_norm`) or vary_hp=True."""
    cfg = cartpole_syn_env_ddqn(num_workers, max_iterations)
    cfg["agents"]["gtn"]["agent_name"] = "TD3_discrete_vary"
    cfg["agents"].pop("ddqn")
    cfg["agents"]["td3_discrete_vary"] = _td3_discrete_section(**agent_over)
    return cfg


def acrobot_syn_env_td3_discrete(num_workers=16, max_iterations=50, **agent_over):
    """Acrobot-v1 SE + TD3_discrete_vary (default_config_acrobot_syn_env.yaml:58-81)."""
    cfg = acrobot_syn_env_duelingddqn(num_workers, max_iterations)
    cfg["agents"]["gtn"]["agent_name"] = "TD3_discrete_vary"
    cfg["agents"].pop("duelingddqn")
    cfg["agents"]["td3_discrete_vary"] = _td3_discrete_section(**agent_over)
    return cfg


def with_vary(config, vary_hp=True):
    """The same experiment with the *_vary agent of the family (default_config_acrobot.yaml:26 ships `agent_name: DDQN_vary`;
    the `<agent>_vary: {vary_hp: ...}` section is :27-28 there)."""
    cfg = copy.deepcopy(config)
    base = cfg["agents"]["gtn"]["agent_name"]
    cfg["agents"]["gtn"]["agent_name"] = base + "_vary"
    cfg["agents"][base.lower().replace("_icm", "") + "_vary"] = {"vary_hp": bool(vary_hp)}     # DDQN_vary(icm=True) reads `ddqn_vary` too
    return cfg


def with_icm(config, lr=1e-4, beta=0.2, eta=0.5, feature_dim=32, hidden_size=128):
    """The same experiment with the agent's Intrinsic Curiosity Module switched on (select_agent "<agent>_icm"; the `icm`
    section's defaults are default_config_cartpole_syn_env.yaml:48-53)."""
    cfg = copy.deepcopy(config)
    cfg["agents"]["gtn"]["agent_name"] += "_icm"
    cfg["agents"]["icm"] = {"lr": lr, "beta": beta, "eta": eta, "feature_dim": feature_dim, "hidden_size": hidden_size}
    return cfg


def fixed_work(config, train_episodes):
    """BASELINE.md §3 fixed-work variant: early-out disabled (solved_reward=+1e9) and a fixed number of train episodes,
    so both the GPU path and the CPU baseline do identical, data-independent amounts of work."""
    cfg = copy.deepcopy(config)
    key = cfg["agents"]["gtn"]["agent_name"].lower()
    for suffix in ("_vary", "_icm"):
        if key.endswith(suffix) and key != "td3_discrete_vary":       # TD3_discrete_vary reads its own section
            key = key[:-len(suffix)]
    cfg["agents"][key]["train_episodes"] = train_episodes
    cfg["envs"][cfg["env_name"]]["solved_reward"] = 1e9
    return cfg
