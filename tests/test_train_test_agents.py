"""SURVEY.md §8(f).1 -- the evaluation harness every downstream experiment of the reference starts from:
experiments/syn_env_evaluate_cartpole_vary_hp_2.py:25-48 `train_test_agents`, called as experiments/syn_env_run_vary_hp.py:32-117 calls it.

The G12 fixtures are runs of the REFERENCE'S OWN FUNCTION (oracle/gen_golden.py g12) on a reference-written checkpoint:
    mode 2  DDQN_vary agents (hyper-parameters drawn per agent) trained on the loaded VirtualEnv -- agent.train(env) with NO test env:
            the meter is fed by the SE's own episode reward and training ends on the virtual rule
            |avg - avg_last| / (|avg_last| + 1e-9) < early_out_virtual_diff (base_agent.py:49-56,134-148; utils.py:94-105)
    mode 1  the same with the base hyper-parameters (vary_hp off: DDQN_vary IS DDQN, DDQN_vary.py:16-21)
    mode 0  the agents trained on the REAL env (syn_env_run_vary_hp.py:47-54): the paper's baseline; real early-out rule on the
            training rewards
The CPU half pins the oracle's `test_mode 1` to them; the GPU half runs the product's train_test_agents (one fused launch per call)
in tape mode against the oracle bit for bit and against the reference's lists."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
G12 = ["g12_train_test_agents_cartpole_mode2_vary", "g12_train_test_agents_cartpole_mode1_plain", "g12_train_test_agents_cartpole_mode0_real_env"]
# the DuelingDDQN sibling script (experiments/syn_env_evaluate_cartpole_vary_hp_2_DuelingDDQN.py: DuelingDDQN_vary, settings in `duelingddqn`)
G12D = "g12d_train_test_agents_cartpole_mode2_dueling_vary"


# the TD3_discrete sibling script (experiments/syn_env_evaluate_cartpole_vary_hp_2_TD3_discrete.py: td3_discrete_vary from the LayerNorm section of
# default_config_cartpole.yaml; drawn shapes 42 x 3 / batch 105 and 63 x 1 / batch 58; 32 and 36 episodes = 880 and 1 040 learn steps)
G12T = "g12t_train_test_agents_cartpole_mode2_td3_discrete_vary"


# the Acrobot script (experiments/syn_env_evaluate_acrobot_vary_hp_2.py: the same function on an Acrobot-v1 SE -- 6-dim states, 3 actions --,
# DDQN_vary over default_config_acrobot.yaml's 128 x 2 DDQN; drawn shapes 42 x 3 / 63 x 1; stops at episodes 28 / 21)
G12A = "g12a_train_test_agents_acrobot_mode2_vary"


def g12_agent(name):
    """(agent_name of the harness, config section, base agent name) of a G12 fixture."""
    return ("DuelingDDQN_vary", "duelingddqn", "DuelingDDQN") if "dueling" in name else ("DDQN_vary", "ddqn", "DDQN")


def g12_oracle_cfg(g, i, grad_chunk=0, dueling=False, **over):
    """The oracle configuration of agent i of a G12 fixture: the config the reference function left behind (its "settings for
    comparability" :29-36 are in config_json), agent i's recorded draw, test_mode 1; mode 0 = the real env as the training env =
    a RewardEnv of type 0 (reward_env.py:80-81: the reward passes through).  grad_chunk: the micro-chunk of the batch gradient's canonical
    order (0 = one piece: the GEMM-tiled kernel; the register-resident kernel's launches carry config.pick_grad_chunk's value)."""
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["a%d_hp_json" % i]))
    cfgd["agents"]["gtn"]["agent_name"] = "DuelingDDQN" if dueling else "DDQN"
    a = cfgd["agents"]["duelingddqn" if dueling else "ddqn"]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (1000, 10, 10, 10, 0.01)
    extra = dict(synthetic_env_type=1, reward_env_type=0) if int(g["mode"]) == 0 else {}
    extra.update(over)
    return orc.ddqn_cfg_from_config(cfgd, grad_chunk=grad_chunk, rng_mode=1, test_mode=1, **orc.hp_overrides(hp), **extra), cfgd, hp


def g12_tapes(g, i):
    pre = "a%d_" % i
    return [g[pre + "tape_" + k] for k in ("eps_uniform", "rand_action", "replay_idx", "train_reset", "test_reset")]


@pytest.mark.parametrize("name", G12 + [G12D, G12A])
def test_g12_oracle_reproduces_the_reference_train_test_agents(golden, name):
    g = golden(name)
    n_agents = int(g["agents_num"])
    assert n_agents == 2 and g["reward_list"].shape == (2, 10)
    for i in range(n_agents):
        pre = "a%d_" % i
        cfg, cfgd, hp = g12_oracle_cfg(g, i, dueling="dueling" in name)
        assert cfg.test_mode == 1 and cfg.early_out_virtual_diff == 0.01 and cfg.agent_kind == int("dueling" in name)
        n = g[pre + "tr_action"].size
        out = orc.ddqn_se_chain(cfg, g["theta"], g[pre + "agent_init"], tapes=orc.make_tapes(*g12_tapes(g, i)), trace_cap=n + 10)
        assert out["rc"] == 0
        tr = out["trace"]
        assert tr["action"].size == n
        assert np.array_equal(tr["action"], g[pre + "tr_action"]) and np.array_equal(tr["explored"], g[pre + "tr_explored"])
        np.testing.assert_allclose(tr["next_state"], g[pre + "tr_next_state"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tr["reward"], g[pre + "tr_reward"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tr["done"], g[pre + "tr_done"], rtol=1e-5, atol=1e-5)      # (a VirtualEnv's done is the raw net output)
        assert np.array_equal(tr["done"] > 0.5, g[pre + "tr_done"] > 0.5)
        losses = tr["loss"][~np.isnan(tr["loss"])]
        np.testing.assert_allclose(losses, g[pre + "losses"], rtol=2e-3, atol=5e-5)      # (the Acrobot run's late losses are ~3e-3: 2e-5 absolute there)
        # what the function returns: reward_list (the final test's returns), [sum(episode_length)], [len(reward_train)] -- the two counters EXACTLY
        e = int(g["episodes_needed"][i, 0])
        assert out["episodes_run"] == e and out["train_steps"] == int(g["train_steps_needed"][i, 0])
        assert np.array_equal(out["episode_len"][:e], g[pre + "episode_length"]) and not out["episode_len"][e:].any()
        np.testing.assert_allclose(out["episode_test_mean"][:e], g[pre + "reward_train"], rtol=0, atol=1e-4)
        assert np.isnan(out["episode_test_mean"][e:]).all()
        np.testing.assert_allclose(out["final_test_returns"], g["reward_list"][i], rtol=0, atol=1e-4)
        # no per-episode tests ran: the only real-env test steps are the final test's
        assert out["test_steps"] == abs(int(np.sum(g["reward_list"][i])))     # CartPole: return == episode length (Acrobot: its negative)


def g12t_oracle_cfg(g, i):
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["a%d_hp_json" % i]))
    a = cfgd["agents"]["td3_discrete_vary"]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (1000, 10, 10, 10, 0.01)
    assert a["use_layer_norm"] is True and a["vary_hp"] is True
    return orc.td3d_cfg_from_config(cfgd, rng_mode=1, hp=hp, test_mode=1), cfgd, hp


def g12t_tapes(g, i):
    return {k: g["a%d_tape_%s" % (i, k)] for k in orc.TD3D_TAPE_KEYS}


def test_g12t_oracle_reproduces_the_td3_discrete_sibling_script(golden):
    """The reference's TD3_discrete harness run replayed by the oracle chain with test_mode 1: the replay buffer's action vectors (Gumbel
    softmax + Gaussian noise) within 1e-5 -- LayerNorm nets, 880 / 1 040 learn steps --, the env's view of them (argmax) exact, the per-episode
    training rewards and the final test within 1e-4, the two counters the function returns EXACTLY."""
    g = golden(G12T)
    assert int(g["agents_num"]) == 2 and g["reward_list"].shape == (2, 10)
    assert g["episodes_needed"].ravel().tolist() == [32, 36]            # the virtual rule said "no" 12 / 16 times before "yes"
    for i in range(2):
        pre = "a%d_" % i
        cfg, cfgd, hp = g12t_oracle_cfg(g, i)
        assert cfg.test_mode == 1 and cfg.early_out_virtual_diff == 0.01 and cfg.use_layer_norm == 1
        assert (cfg.batch_size, cfg.hidden, cfg.layers) == (hp["batch_size"], hp["hidden_size"], hp["hidden_layer"])
        assert orc.td3d_num_params(cfg)[0] == g[pre + "agent_init"].size
        n = g[pre + "tr_reward"].size
        out = orc.td3d_chain(cfg, g["theta"], g[pre + "agent_init"], tapes=orc.make_td3d_tapes(cfg.action_dim, **g12t_tapes(g, i)), trace_cap=n + 4)
        assert out["rc"] == 0
        e = int(g["episodes_needed"][i, 0])
        assert out["episodes_run"] == e and out["train_steps"] == int(g["train_steps_needed"][i, 0]) == n
        assert np.array_equal(out["episode_len"][:e], g[pre + "episode_length"])
        assert out["learn_steps"] == g[pre + "tape_gumbel_target"].shape[0] // cfg.batch_size
        tr = out["trace"]
        np.testing.assert_allclose(tr["action"], g[pre + "rb_action"][:n], rtol=0, atol=1e-5)
        assert np.array_equal(tr["action"].argmax(1), g[pre + "tr_action"].reshape(-1).astype(np.int64))
        np.testing.assert_allclose(tr["next_state"], g[pre + "tr_next_state"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(tr["reward"], g[pre + "tr_reward"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out["episode_test_mean"][:e], g[pre + "reward_train"], rtol=0, atol=1e-4)
        assert np.isnan(out["episode_test_mean"][e:]).all()
        np.testing.assert_allclose(out["final_test_returns"], g["reward_list"][i], rtol=0, atol=1e-4)
        assert out["test_steps"] == int(np.sum(g["reward_list"][i]))          # no per-episode tests; CartPole: return == episode length
        np.testing.assert_allclose(out["final_params"], g[pre + "final_params"], rtol=0, atol=2e-5)      # measured 3.6e-7


def test_g12_fixtures_exercise_the_early_out_rules(golden):
    """The virtual rule said "no" several times before "yes" (first possible stop = episode 20), the real rule fired at its first
    evaluation for one agent and after 200+ episodes of learning for the other; the drawn shapes cover 1 and 2 hidden layers."""
    g2, g1, g0 = (golden(n) for n in G12)
    assert g2["episodes_needed"].ravel().tolist() == [24, 24] and g1["episodes_needed"].ravel().tolist() == [31, 26]
    assert g0["episodes_needed"].ravel().tolist() == [11, 214]
    hp = [json.loads(str(g2["a%d_hp_json" % i])) for i in range(2)]
    assert [h["hidden_layer"] for h in hp] == [1, 2] and [h["batch_size"] for h in hp] == [80, 12]
    assert json.loads(str(g1["a0_hp_json"])) == {"lr": 0.00025, "batch_size": 32, "hidden_size": 64, "hidden_layer": 1}


def test_meter_rules_against_a_python_restatement():
    """AverageMeter.get_mean / get_mean_last + BaseAgent.env_solved (utils.py:94-105, base_agent.py:49-62) restated in python on the
    oracle's per-episode list: the episode at which a test_mode 1 chain stops is the first one the restated rule accepts."""
    g = np.load(os.path.join(HERE, "golden", G12[1] + ".npz"))
    cfg, _, _ = g12_oracle_cfg(g, 0)
    out = orc.ddqn_se_chain(cfg, g["theta"], g["a0_agent_init"], tapes=orc.make_tapes(*g12_tapes(g, 0)))
    vals = out["episode_test_mean"][:out["episodes_run"]].tolist()

    def mean(v, num, ignore_last):
        sl = v[max(len(v) - num - ignore_last, 0): max(len(v) - ignore_last, 0)]
        return sum(sl) / (len(sl) + 1e-9)
    stops = []
    for ep in range(10, len(vals)):
        v = vals[:ep + 1]
        avg, last = mean(v, 10, 0), mean(v, 10, 10)
        if abs(avg - last) / (abs(last) + 1e-9) < 0.01 and ep >= 20:
            stops.append(ep)
    assert stops == [len(vals) - 1]


# ------------------------------------------------------------------------------------------------------------------------------
# GPU half: the product's train_test_agents (learning_environments_amd/experiments/syn_env_evaluate.py), one fused launch per call
# ------------------------------------------------------------------------------------------------------------------------------
def _load_ckpt_b(tmp_path, ckpt="ckpt_cartpole_se_reference_b.pt"):
    import shutil
    from learning_environments_amd.experiments.syn_env_evaluate import load_envs_and_config
    shutil.copy(os.path.join(HERE, "golden", ckpt), tmp_path / "model.pt")
    return load_envs_and_config("model.pt", str(tmp_path), "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("name", G12 + [G12D, G12A])
def test_g12_product_train_test_agents_replays_the_reference_run(golden, tmp_path, name):
    """The reference's recorded draws (hyper-parameters, fresh agents, RNG tapes) replayed through the product function: the three returned
    lists equal the reference's (`reward_list` within 1e-4, `train_steps_needed` / `episodes_needed` EXACTLY) and the oracle's bit for bit,
    as do the per-episode training rewards and lengths."""
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    g = golden(name)
    mode, n_agents = int(g["mode"]), int(g["agents_num"])
    venv, real_env, config = _load_ckpt_b(tmp_path, "ckpt_acrobot_se_reference_c.pt" if "acrobot" in name else "ckpt_cartpole_se_reference_b.pt")
    assert np.array_equal(venv.env.flat_params().cpu().numpy(), g["theta"])
    hps = [json.loads(str(g["a%d_hp_json" % i])) for i in range(n_agents)]
    replay = dict(hp=hps, agent_init=[g["a%d_agent_init" % i] for i in range(n_agents)],
                  tapes={k: [g["a%d_tape_%s" % (i, k)] for i in range(n_agents)] for k in ("eps_uniform", "rand_action", "replay_idx", "train_reset", "test_reset")})
    train_env = real_env if mode == 0 else venv
    agent_name, section, _ = g12_agent(name)
    rewards, steps, episodes = train_test_agents(train_env, real_env, config, agents_num=n_agents, agent_name=agent_name, vary_hp=(mode != 1), replay=replay)
    # the settings for comparability were applied to the caller's config in place, like the reference does
    a = config["agents"][section]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (1000, 10, 10, 10, 0.01)
    last = train_test_agents.last
    assert last["inner"].cfg.test_mode == 1 and last["inner"].cfg.synthetic_env_type == (1 if mode == 0 else 0)
    # the base hyper-parameters (one hidden layer of 64) run in the register-resident kernel, the drawn ones (per-chain shapes) in the GEMM-tiled one
    assert last["inner"].dueling == (mode != 1) and (last["inner"].cfg.grad_chunk > 0) == (mode == 1)
    assert steps == g["train_steps_needed"].tolist() and episodes == g["episodes_needed"].tolist()
    np.testing.assert_allclose(np.array(rewards), g["reward_list"], rtol=0, atol=1e-4)
    for i in range(n_agents):
        pre = "a%d_" % i
        np.testing.assert_allclose(last["reward_train"][i], g[pre + "reward_train"], rtol=0, atol=1e-4)
        assert last["episode_length"][i] == g[pre + "episode_length"].tolist()
        ocfg, _, _ = g12_oracle_cfg(g, i, grad_chunk=last["inner"].cfg.grad_chunk, dueling="dueling" in name)
        o = orc.ddqn_se_chain(ocfg, g["theta"], g[pre + "agent_init"], tapes=orc.make_tapes(*g12_tapes(g, i)))
        assert rewards[i] == o["final_test_returns"].tolist()
        assert last["reward_train"][i] == o["episode_test_mean"][:o["episodes_run"]].tolist()
        assert last["inner"].stats[i].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.gpu
def test_g12t_product_train_test_agents_replays_the_td3_discrete_sibling_run(golden, tmp_path):
    """The TD3_discrete sibling script's reference run replayed through the product function (agent_name "td3_discrete_vary": the fused
    TD3-discrete loop in tape mode, the two drawn shapes in one launch): equal to the oracle bit for bit, to the reference within the bars of
    the CPU half."""
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    g = golden(G12T)
    venv, real_env, config = _load_ckpt_b(tmp_path)
    assert np.array_equal(venv.env.flat_params().cpu().numpy(), g["theta"])
    # the sibling script takes its agent section from default_config_cartpole.yaml's `td3_discrete_vary_layer_norm_2` (:35-40); the fixture's
    # config_json holds what load_envs_and_config produced there
    config["agents"]["td3_discrete_vary"] = dict(json.loads(str(g["config_json"]))["agents"]["td3_discrete_vary"], train_episodes=5, vary_hp=False)
    hps = [json.loads(str(g["a%d_hp_json" % i])) for i in range(2)]
    replay = dict(hp=hps, agent_init=[g["a%d_agent_init" % i] for i in range(2)],
                  tapes={k: [g["a%d_tape_%s" % (i, k)] for i in range(2)] for k in orc.TD3D_TAPE_KEYS})
    rewards, steps, episodes = train_test_agents(venv, real_env, config, agents_num=2, agent_name="td3_discrete_vary", replay=replay)
    a = config["agents"]["td3_discrete_vary"]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"], a["vary_hp"]) == (1000, 10, 10, 10, 0.01, True)
    last = train_test_agents.last
    assert last["inner"].cfg.test_mode == 1
    assert steps == g["train_steps_needed"].tolist() and episodes == g["episodes_needed"].tolist()
    np.testing.assert_allclose(np.array(rewards), g["reward_list"], rtol=0, atol=1e-4)
    for i in range(2):
        pre = "a%d_" % i
        np.testing.assert_allclose(last["reward_train"][i], g[pre + "reward_train"], rtol=0, atol=1e-4)
        assert last["episode_length"][i] == g[pre + "episode_length"].tolist()
        ocfg, _, _ = g12t_oracle_cfg(g, i)
        o = orc.td3d_chain(ocfg, g["theta"], g[pre + "agent_init"], tapes=orc.make_td3d_tapes(ocfg.action_dim, **g12t_tapes(g, i)))
        assert rewards[i] == o["final_test_returns"].tolist()
        assert last["reward_train"][i] == o["episode_test_mean"][:o["episodes_run"]].tolist()
        assert last["inner"].stats[i].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,vary", [(2, True), (1, False), (0, True), (0, False)])
def test_product_train_test_agents_counter_mode_vs_oracle(tmp_path, mode, vary):
    """The function as a user calls it (own counter-RNG draws): 5 agents in ONE launch -- drawn hyper-parameters (per-chain shapes: the
    GEMM-tiled kernel) or the base ones (the register-resident kernel: test_mode 1 on the VirtualEnv, and its RENV instantiation for the
    real env as the training env) -- each bit-identical to the oracle chain run with that agent's key, draw and fresh parameters."""
    import torch
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    venv, real_env, config = _load_ckpt_b(tmp_path)
    config["agents"]["ddqn"]["early_out_virtual_diff_unused"] = 0      # (unknown keys are ignored like in the reference)
    train_env = real_env if mode == 0 else venv
    rewards, steps, episodes = train_test_agents(train_env, real_env, config, agents_num=5, vary_hp=vary, seed=11)
    last = train_test_agents.last
    inner = last["inner"]
    assert inner.dueling == vary and inner.cfg.test_mode == 1 and inner.cfg.synthetic_env_type == (1 if mode == 0 else 0)
    keys = chain_keys(11, 0, np.arange(5), np.zeros(5, np.int64))
    theta = venv.env.flat_params().cpu().numpy()
    cfgd = json.loads(json.dumps(config))
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    if not vary:
        from learning_environments_amd.agents.nes_common import fresh_agent_init
        gen = torch.Generator(device="cuda")
        gen.manual_seed(11)
        inits = fresh_agent_init(last["task"].agent_bounds, 5, gen, torch.device("cuda")).cpu().numpy()
        hps = [None] * 5
    else:
        inits, hps = inner.agent_init.cpu().numpy(), last["hp"]
        assert len({(h["batch_size"], h["hidden_size"], h["hidden_layer"]) for h in hps}) >= 4       # a heterogeneous population
    lens = []
    for c in range(5):
        over = dict(synthetic_env_type=1, reward_env_type=0) if mode == 0 else {}
        if hps[c] is not None:
            over.update(orc.hp_overrides(hps[c]))
        ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=inner.cfg.grad_chunk, rng_mode=0, test_mode=1, **over)
        p_c = orc.mlp_num_params(orc.mlp_desc(4, ocfg.q_hidden, ocfg.q_layers, 2, ocfg.q_act))
        o = orc.ddqn_se_chain(ocfg, theta if mode != 0 else np.zeros(1, np.float32), inits[c][:p_c], rng_key=int(keys[c]))
        assert o["rc"] == 0
        assert rewards[c] == o["final_test_returns"].tolist()
        assert steps[c] == [o["train_steps"]] and episodes[c] == [o["episodes_run"]]
        assert last["reward_train"][c] == o["episode_test_mean"][:o["episodes_run"]].tolist()
        assert last["episode_length"][c] == o["episode_len"][:o["episodes_run"]].tolist()
        lens.append(o["episodes_run"])
    assert min(lens) >= 11 and len(set(lens)) > 1          # the early-out fired at different episodes


@pytest.mark.gpu
def test_product_train_test_agents_acrobot_script_vs_oracle():
    """experiments/syn_env_evaluate_acrobot_vary_hp_2.py: the same function on an Acrobot-v1 SE (DDQN_vary over default_config_acrobot.yaml's
    128-wide two-layer DDQN; here shrunk so that the CPU oracle finishes in seconds).  Three agents with drawn shapes in one launch, each
    bit-identical to the oracle chain with test_mode 1; training ends on the virtual rule, the final test runs on the real Acrobot."""
    import torch
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.envs.env_factory import EnvFactory
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    config = configs.with_vary(configs.acrobot_syn_env_ddqn(num_workers=1))
    config["device"] = "cuda"
    config["agents"]["ddqn"].update(hidden_size=24, batch_size=16)
    config["envs"]["Acrobot-v1"].update(max_steps=30, hidden_size=32)
    torch.manual_seed(3)
    fac = EnvFactory(config)
    venv, real_env = fac.generate_virtual_env(), fac.generate_real_env()
    rewards, steps, episodes = train_test_agents(venv, real_env, config, agents_num=3, train_episodes=40, seed=21)
    last = train_test_agents.last
    inner, hps = last["inner"], last["hp"]
    assert inner.cfg.test_mode == 1 and inner.cfg.env_id == 1 and len(hps) == 3
    keys = chain_keys(21, 0, np.arange(3), np.zeros(3, np.int64))
    theta = venv.env.flat_params().cpu().numpy()
    inits = inner.agent_init.cpu().numpy()
    cfgd = json.loads(json.dumps(config))
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    for c in range(3):
        ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=inner.cfg.grad_chunk, rng_mode=0, test_mode=1, **orc.hp_overrides(hps[c]))
        p_c = orc.mlp_num_params(orc.mlp_desc(6, ocfg.q_hidden, ocfg.q_layers, 3, ocfg.q_act))
        o = orc.ddqn_se_chain(ocfg, theta, inits[c][:p_c], rng_key=int(keys[c]))
        assert o["rc"] == 0 and o["episodes_run"] >= 21
        assert rewards[c] == o["final_test_returns"].tolist() and len(rewards[c]) == 10
        assert steps[c] == [o["train_steps"]] and episodes[c] == [o["episodes_run"]]
        assert last["reward_train"][c] == o["episode_test_mean"][:o["episodes_run"]].tolist()
        assert inner.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.gpu
def test_product_train_test_agents_correlation_variant(tmp_path):
    """experiments/syn_env_evaluate_cartpole_vary_hp_2_correlation.py:25-87: per drawn configuration, `repeats` DDQN agents with THAT configuration on
    the SE and as many on the real env, 100 test episodes each, early_out_num 1000; three dicts {"config", "synthetic", "real"}.  Shapes and
    settings as the reference's; the last launch (last configuration, real env) against the oracle chain of one of its agents."""
    import torch
    from learning_environments_amd.agents.nes_common import chain_keys, fresh_agent_init
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents, train_test_agents_correlation
    venv, real_env, config = _load_ckpt_b(tmp_path)
    rewards, steps, episodes = train_test_agents_correlation(venv, real_env, config, agents_num=2, repeats=3, train_episodes=14, seed=4)
    a = config["agents"]["ddqn"]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (14, 10, 1000, 100, 0.01)
    for d in (rewards, steps, episodes):
        assert set(d) == {"config", "synthetic", "real"} and d["config"] is config and len(d["synthetic"]) == len(d["real"]) == 6
    assert all(len(r) == 100 for r in rewards["synthetic"] + rewards["real"])
    assert all(e == [14] for e in episodes["synthetic"])              # early_out_num 1000: the virtual rule can never fire, every agent trains all its episodes
    assert all(11 <= e[0] <= 14 for e in episodes["real"])            # (on the real env the REAL rule applies: mean reward >= solved_reward, 16 in this checkpoint)
    hps = train_test_agents_correlation.last["hp"]
    assert len(hps) == 2 and hps[0] != hps[1]
    # the last launch: configuration 1 on the real env, three agents of ONE shape (a fixed-shape launch, not the per-chain-shape kernel)
    last = train_test_agents.last
    inner = last["inner"]
    assert inner.chains == 3 and inner.cfg.test_mode == 1 and inner.cfg.synthetic_env_type == 1
    assert (inner.cfg.batch_size, inner.cfg.q_hidden, inner.cfg.q_layers) == (hps[1]["batch_size"], hps[1]["hidden_size"], max(1, hps[1]["hidden_layer"]))
    seed_l = 4 + 7919 * 2
    keys = chain_keys(seed_l, 0, np.arange(3), np.zeros(3, np.int64))
    cfgd = json.loads(json.dumps(config))
    cfgd["agents"]["ddqn"].update(hps[1])
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=inner.cfg.grad_chunk, rng_mode=0, test_mode=1, synthetic_env_type=1, reward_env_type=0)
    if last["task"].needs_agent_init():
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed_l)
        inits = fresh_agent_init(last["task"].agent_bounds, 3, gen, torch.device("cuda")).cpu().numpy()
    else:
        inits = inner.agent_init.cpu().numpy()
    c = 2
    p_c = orc.mlp_num_params(orc.mlp_desc(4, ocfg.q_hidden, ocfg.q_layers, 2, ocfg.q_act))
    o = orc.ddqn_se_chain(ocfg, np.zeros(1, np.float32), inits[c][:p_c], rng_key=int(keys[c]))
    assert o["rc"] == 0 and episodes["real"][3 + c] == [o["episodes_run"]]
    assert rewards["real"][3 + c] == o["final_test_returns"].tolist() and steps["real"][3 + c] == [o["train_steps"]]


@pytest.mark.gpu
def test_product_train_test_agents_refuses_what_the_harness_does_not_train(tmp_path):
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    venv, real_env, config = _load_ckpt_b(tmp_path)
    with pytest.raises(NotImplementedError):
        train_test_agents(venv, real_env, config, agents_num=1, agent_name="PPO")
    with pytest.raises(ValueError):
        train_test_agents(venv, venv, config, agents_num=1)


@pytest.mark.gpu
@pytest.mark.parametrize("agent_name", ["DuelingDDQN_vary", "td3_discrete_vary"])
def test_product_train_test_agents_sibling_scripts_vs_oracle(tmp_path, agent_name):
    """experiments/syn_env_evaluate_cartpole_vary_hp_2_DuelingDDQN.py / _TD3_discrete.py: the same function around another `_vary` agent
    (their settings go to that agent's section).  Four agents with drawn hyper-parameters in one launch, each bit-identical to the oracle
    chain with test_mode 1.  (The sections are shrunk first -- the config is the caller's input -- so the CPU oracle finishes in seconds.)"""
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.experiments.syn_env_evaluate import HARNESS_AGENTS, train_test_agents
    venv, real_env, config = _load_ckpt_b(tmp_path)
    section = HARNESS_AGENTS[agent_name.lower()][0]
    config["agents"][section].update(hidden_size=20, batch_size=12)
    if section == "duelingddqn":
        config["agents"][section]["feature_dim"] = 16
    rewards, steps, episodes = train_test_agents(venv, real_env, config, agents_num=4, agent_name=agent_name, train_episodes=60, seed=5)
    a = config["agents"][section]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (60, 10, 10, 10, 0.01)
    last = train_test_agents.last
    inner, hps = last["inner"], last["hp"]
    assert inner.cfg.test_mode == 1 and len(hps) == 4
    keys = chain_keys(5, 0, np.arange(4), np.zeros(4, np.int64))
    theta = venv.env.flat_params().cpu().numpy()
    inits = inner.agent_init.cpu().numpy()
    cfgd = json.loads(json.dumps(config))
    for c in range(4):
        if section == "duelingddqn":
            cfgd["agents"]["gtn"]["agent_name"] = "DuelingDDQN"
            ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=0, test_mode=1, **orc.hp_overrides(hps[c]))
            o = orc.ddqn_se_chain(ocfg, theta, inits[c][:orc.dueling_num_params(ocfg)], rng_key=int(keys[c]))
        else:
            ocfg = orc.td3d_cfg_from_config(cfgd, rng_mode=0, hp=hps[c], test_mode=1)
            o = orc.td3d_chain(ocfg, theta, inits[c][:orc.td3d_num_params(ocfg)[0]], rng_key=int(keys[c]))
        assert o["rc"] == 0
        assert rewards[c] == o["final_test_returns"].tolist()
        assert steps[c] == [o["train_steps"]] and episodes[c] == [o["episodes_run"]]
        assert last["reward_train"][c] == o["episode_test_mean"][:o["episodes_run"]].tolist()
        assert inner.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert o["test_steps"] == int(sum(rewards[c]))          # no per-episode tests: only the final test touched the real env


def _mirror(ocfg, cls):
    c = cls()
    for f, _ in cls._fields_:
        setattr(c, f, getattr(ocfg, f, 0))      # (team_size / kernel_variant exist only in the HIP cfg)
    return c


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("virtual", [True, False])
def test_td3_kernel_test_mode_1_vs_oracle(golden, virtual):
    """BaseAgent.train without a test env in the TD3 loop (lenv_td3_cfg::test_mode 1): on a VirtualEnv the virtual rule ends training, on a
    RewardEnv the real rule on the shaped training rewards; no per-episode tests either way.  Bit-exact against the oracle; the chains
    stop at different episodes."""
    import torch
    from learning_environments_amd import _lib, engine
    engine.require_device()
    g = golden("g8ts_calc_score_cheetah_td3_virtual_env" if virtual else "g8t_calc_score_cheetah_td3")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["td3"].update(train_episodes=40, init_episodes=2, early_out_num=3, test_episodes=2, batch_size=16, hidden_size=24,
                                 early_out_virtual_diff=0.25)
    cfgd["envs"]["HalfCheetah-v3"].update(max_steps=6, solved_reward=1.0)
    ocfg = orc.td3_cfg_from_config(cfgd, rng_mode=0, test_mode=1)
    assert ocfg.virtual_env == int(virtual) and ocfg.early_out_virtual_diff == 0.25
    cfg = _mirror(ocfg, _lib.Td3Cfg)
    chains = 6
    il = engine.Td3InnerLoop(cfg, chains, want_episode_stats=True)
    rng = np.random.RandomState(5)
    theta = g["theta"]
    eps = (rng.randn(chains, theta.size) * 0.05).astype(np.float32)
    worker, sign = np.arange(chains).astype(np.int32), np.ones(chains, np.float32)
    agent_init = rng.uniform(-0.2, 0.2, (chains, il.p_agent)).astype(np.float32)
    keys = np.array([orc.chain_key(77, 0, c, 1) for c in range(chains)], np.uint64)
    il.run(_dev(theta), _dev(eps), _dev(worker), _dev(sign), _dev(agent_init), rng_keys=_dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    stops = []
    for c in range(chains):
        o = orc.td3_rn_chain(ocfg, (eps[c] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c]))
        assert o["rc"] == 0
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]), c
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]) and float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        stops.append(o["episodes_run"])
    assert len(set(stops)) > 1 and min(stops) < 40, stops


@pytest.mark.gpu
def test_ql_kernel_test_mode_1_vs_oracle(golden):
    """QL on the Cliff RewardEnv trained without a test env (lenv_ql_cfg::test_mode 1): the shaped training rewards feed the meter, the real
    rule ends training, only the final test walks the real grid."""
    import torch
    from learning_environments_amd import _lib, engine
    from learning_environments_amd.envs.gridworld import transition_tables
    engine.require_device()
    g = golden("g9_calc_score_cliff_a")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["ql"].update(eps_init=0.3, eps_min=0.05, eps_decay=0.9, alpha=0.7, train_episodes=60, early_out_num=3)
    cfgd["envs"]["Cliff"]["solved_reward"] = -40.0
    tables = transition_tables("Cliff")
    ocfg = orc.ql_cfg_from_config(cfgd, tables, rng_mode=0, test_mode=1)
    cfg = _mirror(ocfg, _lib.QlCfg)
    chains = 8
    rng = np.random.RandomState(3)
    theta = g["theta"]
    eps = (rng.randn(chains, theta.size) * 0.2).astype(np.float32)
    worker, sign = np.arange(chains).astype(np.int32), np.ones(chains, np.float32)
    keys = np.array([orc.chain_key(9, 2, c, 1) for c in range(chains)], np.uint64)
    il = engine.QlInnerLoop(cfg, chains, tables)
    il.run(_dev(theta), _dev(eps), _dev(worker), _dev(sign), rng_keys=_dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    stops = []
    for c in range(chains):
        o = orc.ql_rn_chain(ocfg, (eps[c] + theta).astype(np.float32), tables, rng_key=int(keys[c]))
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.q_table[c].cpu().numpy().reshape(o["q_table"].shape), o["q_table"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        stops.append(o["episodes_run"])
    assert len(set(stops)) > 1, stops
