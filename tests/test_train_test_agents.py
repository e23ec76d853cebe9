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


def g12_oracle_cfg(g, i, **over):
    """The oracle configuration of agent i of a G12 fixture: the config the reference function left behind (its "settings for
    comparability" :29-36 are in config_json), agent i's recorded draw, test_mode 1; mode 0 = the real env as the training env =
    a RewardEnv of type 0 (reward_env.py:80-81: the reward passes through)."""
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["a%d_hp_json" % i]))
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    a = cfgd["agents"]["ddqn"]
    assert (a["train_episodes"], a["init_episodes"], a["early_out_num"], a["test_episodes"], a["early_out_virtual_diff"]) == (1000, 10, 10, 10, 0.01)
    extra = dict(synthetic_env_type=1, reward_env_type=0) if int(g["mode"]) == 0 else {}
    extra.update(over)
    return orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, test_mode=1, **orc.hp_overrides(hp), **extra), cfgd, hp


def g12_tapes(g, i):
    pre = "a%d_" % i
    return [g[pre + "tape_" + k] for k in ("eps_uniform", "rand_action", "replay_idx", "train_reset", "test_reset")]


@pytest.mark.parametrize("name", G12)
def test_g12_oracle_reproduces_the_reference_train_test_agents(golden, name):
    g = golden(name)
    n_agents = int(g["agents_num"])
    assert n_agents == 2 and g["reward_list"].shape == (2, 10)
    for i in range(n_agents):
        pre = "a%d_" % i
        cfg, cfgd, hp = g12_oracle_cfg(g, i)
        assert cfg.test_mode == 1 and cfg.early_out_virtual_diff == 0.01
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
        np.testing.assert_allclose(losses, g[pre + "losses"], rtol=2e-3, atol=1e-6)
        # what the function returns: reward_list (the final test's returns), [sum(episode_length)], [len(reward_train)] -- the two counters EXACTLY
        e = int(g["episodes_needed"][i, 0])
        assert out["episodes_run"] == e and out["train_steps"] == int(g["train_steps_needed"][i, 0])
        assert np.array_equal(out["episode_len"][:e], g[pre + "episode_length"]) and not out["episode_len"][e:].any()
        np.testing.assert_allclose(out["episode_test_mean"][:e], g[pre + "reward_train"], rtol=0, atol=1e-4)
        assert np.isnan(out["episode_test_mean"][e:]).all()
        np.testing.assert_allclose(out["final_test_returns"], g["reward_list"][i], rtol=0, atol=1e-4)
        # no per-episode tests ran: the only real-env test steps are the final test's
        assert out["test_steps"] == int(np.sum(g["reward_list"][i]))          # CartPole: return == episode length


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
