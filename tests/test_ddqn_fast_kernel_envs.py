"""The register-resident DDQN kernel (csrc/ddqn_se_inner_loop.hip) beyond the VirtualEnv + calc_score case it was built for (round 6):
  * RENV instantiations: the agent trains on a RewardEnv over the real CartPole / Acrobot (default_config_cartpole_reward_env.yaml,
    envs/reward_env.py:61-133) or -- reward type 0 -- on the real env itself (experiments/syn_env_run_vary_hp.py:47-54, mode 0);
  * lenv_ddqn_cfg::test_mode 1 = BaseAgent.train without a test env (the evaluation harness).
A launch takes this kernel when its cfg carries an explicit gradient micro-chunk (config.pick_grad_chunk: what the product's configs get);
grad_chunk 0 = one sequential batch gradient = the GEMM-tiled kernel.  Everything here is bit-exact against the oracle run with the SAME
micro-chunk, and within the fixture tolerances of the reference's own runs; the GEMM-tiled kernel runs the same chains as a second witness
(equal up to the gradient's summation order, i.e. equal trajectories until rounding moves an argmax)."""
import ctypes as C
import json

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from learning_environments_amd import engine
    engine.require_device()
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cfgs(orc, cfgd, **over):
    """(oracle cfg, HIP cfg) with the micro-chunk config.pick_grad_chunk chooses for the shape."""
    from learning_environments_amd import _lib
    from learning_environments_amd.config import pick_grad_chunk
    o = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, **over)
    c = _lib.DdqnCfg()
    for f, _ in _lib.DdqnCfg._fields_:
        setattr(c, f, getattr(o, f, 0))
    chunk = pick_grad_chunk(c)
    assert chunk > 0, "the shape must fit the register-resident kernel"
    o.grad_chunk = c.grad_chunk = chunk
    assert _lib.lib().lenv_ddqn_se_lds_bytes(C.byref(c)) > 0
    return o, c


@pytest.mark.parametrize("name", ["g8r_calc_score_cartpole_ddqn_reward_env", "g8r6_calc_score_cartpole_ddqn_reward_env_t6"])
def test_reward_env_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    """The reference's DDQN-on-a-CartPole-RewardEnv runs (types 2 and 6) replayed in the RENV instantiation."""
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _cfgs(orc, cfgd, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.synthetic_env_type == 1
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))), rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))), test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    assert not il.dueling                                   # the register-resident kernel
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["state"][c, :n].cpu().numpy(), o["trace"]["state"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 1].cpu().numpy(), o["trace"]["done"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]) and float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        # the reference's own run
        assert np.array_equal(act & 0xFFFF, g["tr_action"]) and np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"])
        np.testing.assert_allclose(il.trace["reward_done"][c, :n, 0].cpu().numpy(), g["tr_reward"], rtol=0, atol=1e-6)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("env_name,rtype,act,q_act,test_mode", [
    ("CartPole-v0", 2, "prelu", "leakyrelu", 0), ("CartPole-v0", 0, "relu", "tanh", 0), ("CartPole-v0", 1, "tanh", "relu", 0),
    ("Acrobot-v1", 5, "leakyrelu", "tanh", 0), ("Acrobot-v1", 6, "identity", "relu", 0),
    ("CartPole-v0", 2, "prelu", "relu", 1), ("CartPole-v0", 0, "relu", "leakyrelu", 1), ("Acrobot-v1", 1, "tanh", "tanh", 1)])
def test_reward_env_counter_mode_vs_oracle(eng, orc, golden, env_name, rtype, act, q_act, test_mode):
    """Every info-free reward type on both real envs, perturbed reward nets, with per-episode tests (calc_score) and without a test env
    (test_mode 1: the shaped / real training rewards feed the meter, the real rule ends training): bit-exact against the oracle.  The
    early-out threshold sits inside the range of the runs' meters, so some chains stop early."""
    g = golden("g8r_calc_score_cartpole_ddqn_reward_env")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["gtn"].update(agent_name="DDQN", synthetic_env_type=1)
    cfgd["env_name"] = env_name
    base_env = dict(list(cfgd["envs"].values())[0])
    base_env.update(hidden_size=40, hidden_layer=1, activation_fn=act, reward_env_type=rtype, info_dim=0, max_steps=14,
                    solved_reward=(12.0 if env_name == "CartPole-v0" else -13.5) if rtype in (0, 2, 6) else 0.05)
    cfgd["envs"] = {env_name: base_env}
    cfgd["agents"]["ddqn"].update(batch_size=20, test_episodes=3, init_episodes=1, hidden_size=24, hidden_layer=1, activation_fn=q_act, early_out_num=2)
    ocfg, cfg = _cfgs(orc, cfgd, rng_mode=0, train_episodes=8, test_mode=test_mode)
    S = cfg.state_dim
    P_rn = orc.mlp_num_params(orc.mlp_desc(1 if rtype == 0 else S, 40, 1, 1, act))
    chains = 6
    rng = np.random.RandomState(81)
    theta = (rng.randn(P_rn) * 0.3).astype(np.float32)
    eps = (rng.randn(2, P_rn) * 0.1).astype(np.float32)
    worker, sign = np.array([0, 0, 0, 1, 1, 1], np.int32), np.array([0.0, 1.0, -1.0] * 2, np.float32)
    keys = np.array([orc.chain_key(41, 5, int(worker[c]), c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=120, want_final_online=True)
    assert not il.dueling
    agent_init = (rng.uniform(-0.3, 0.3, (chains, il.p_agent))).astype(np.float32)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    stops = []
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=120, want_final_online=True)
        m = o["trace"]["action"].size
        assert o["rc"] == 0 and o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :m].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.trace["reward_done"][c, :m, 0].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.trace["reward_done"][c, :m, 1].cpu().numpy(), o["trace"]["done"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]), c
        assert np.array_equal(il.final_online[c].cpu().numpy(), o["final_online"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        if test_mode == 1:
            assert o["test_steps"] == int(np.sum(np.abs(o["final_test_returns"])))      # only the final test touched the real env (|return| = length)
        stops.append(o["episodes_run"])
    assert min(stops) >= 2


@pytest.mark.parametrize("env_name,q_act", [("CartPole-v0", "tanh"), ("Acrobot-v1", "relu")])
def test_virtual_env_test_mode_1_vs_oracle_and_gemm_kernel(eng, orc, golden, env_name, q_act):
    """train(env) without a test env on a VirtualEnv in the register-resident kernel: the SE's episode rewards feed the meter, the virtual rule
    (early_out_virtual_diff) ends training at different episodes for different chains; bit-exact against the oracle.  The GEMM-tiled kernel
    (kernel route of grad_chunk 0) runs the same chains with the batch gradient summed in one piece: same counters and, here, the same stops."""
    g = golden("g8_calc_score_cartpole_a")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["env_name"] = env_name
    e = dict(list(cfgd["envs"].values())[0])
    e.update(max_steps=12, hidden_size=32)
    cfgd["envs"] = {env_name: e}
    cfgd["agents"]["ddqn"].update(batch_size=24, test_episodes=2, init_episodes=2, hidden_size=20, activation_fn=q_act, early_out_num=3,
                                  early_out_virtual_diff=0.08)
    ocfg, cfg = _cfgs(orc, cfgd, rng_mode=0, train_episodes=30, test_mode=1)
    S, A = cfg.state_dim, cfg.num_actions
    p_theta = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, 32, 1, e["activation_fn"]))
    chains = 6
    rng = np.random.RandomState(7)
    theta = (rng.randn(p_theta) * 0.25).astype(np.float32)
    eps = (rng.randn(chains, p_theta) * 0.05).astype(np.float32)
    worker, sign = np.arange(chains).astype(np.int32), np.ones(chains, np.float32)
    keys = np.array([orc.chain_key(3, 9, c, 1) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, want_final_online=True)
    assert not il.dueling
    agent_init = rng.uniform(-0.4, 0.4, (chains, il.p_agent)).astype(np.float32)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    stops = []
    for c in range(chains):
        o = orc.ddqn_se_chain(ocfg, (eps[c] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c]), want_final_online=True)
        assert o["rc"] == 0
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]), c
        assert np.array_equal(il.final_online[c].cpu().numpy(), o["final_online"]), c
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]) and float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        stops.append(o["episodes_run"])
    assert len(set(stops)) > 1 and min(stops) >= 2 + 3, stops
