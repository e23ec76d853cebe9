"""GPU parity tests proper: the HIP path (through the C-ABI of liblenv_hip.so) against the CPU oracle on the
same seeded inputs, and against the committed golden vectors produced by the reference.

Bar: BIT-EXACT against the oracle (the kernels implement the oracle's canonical fp32 order; integer/index
work is exact by construction); within the fixture tolerances of tests/test_oracle_golden.py against the
reference's own numbers (torch's summation order / tanh differ from the canonical order by O(1e-7)).
"""
import copy
import json

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

ACTS = ["identity", "relu", "leakyrelu", "tanh", "prelu"]


@pytest.fixture(scope="module")
def eng():
    from learning_environments_amd import engine
    engine.require_device()
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_native_library_loaded(eng):
    from learning_environments_amd import _lib
    assert _lib.lib().lenv_abi_version() == 7
    with open("/proc/self/maps") as f:
        assert "liblenv_hip.so" in f.read()


def test_se_step_population_golden_and_oracle(eng, orc, golden):
    g = golden("g1_virtual_env_step")
    rng = np.random.RandomState(0)
    for ci in range(int(g["n_cases"])):
        pre = "c%02d_" % ci
        S, A, H, L, act = [int(v) for v in g[pre + "meta"]]
        theta, state, action = g[pre + "theta"], g[pre + "state"], g[pre + "action"]
        n = state.shape[0]
        # (a) unperturbed, against the reference's outputs and the oracle
        ns, r, d = eng.se_step_population(eng.se_descs(S, A, H, L, ACTS[act]), dev(theta), None, None, None, dev(state), dev(action))
        ons, orr, od = orc.se_step_population(orc.se_descs(S, A, H, L, ACTS[act]), theta, None, None, None, state, action)
        assert np.array_equal(ns.cpu().numpy(), ons), pre
        assert np.array_equal(r.cpu().numpy(), orr), pre
        assert np.array_equal(d.cpu().numpy(), od), pre
        np.testing.assert_allclose(ns.cpu().numpy(), g[pre + "next_state"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(r.cpu().numpy(), g[pre + "reward"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(d.cpu().numpy(), g[pre + "done"], rtol=2e-6, atol=2e-6)
        # (b) perturbed population: chains = 3*pop over pop noise rows
        pop = 4
        eps = (rng.randn(pop, theta.size) * 0.05).astype(np.float32)
        worker = np.repeat(np.arange(pop), 3).astype(np.int32)
        sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
        st = np.tile(state[:1], (3 * pop, 1)) + rng.randn(3 * pop, S).astype(np.float32) * 0.1
        ac = rng.randint(0, A, 3 * pop).astype(np.int32)
        ns, r, d = eng.se_step_population(eng.se_descs(S, A, H, L, ACTS[act]), dev(theta), dev(eps), dev(worker), dev(sign), dev(st), dev(ac))
        ons, orr, od = orc.se_step_population(orc.se_descs(S, A, H, L, ACTS[act]), theta, eps, worker, sign, st, ac)
        assert np.array_equal(ns.cpu().numpy(), ons), pre
        assert np.array_equal(r.cpu().numpy(), orr) and np.array_equal(d.cpu().numpy(), od), pre


@pytest.mark.parametrize("L,H,act", [(2, 24, "relu"), (3, 33, "tanh"), (2, 96, "leakyrelu")])
def test_se_step_population_layer_norm(eng, orc, L, H, act):
    """VirtualEnv.step with `use_layer_norm: True` in the env's section (models/model_utils.py:22-37; envs/virtual_env.py:16-33): the three
    SE nets each carry ONE shared nn.LayerNorm behind their hidden Linear 2..L.  lenv_se_step_population against the oracle bit for bit
    (perturbed population included) and against torch modules from the package's builder (= the reference's, fixture G1LN) within 2e-6."""
    import torch as T
    from learning_environments_amd.models.model_utils import build_nn_from_config
    S, A = 6, 3
    T.manual_seed(5)
    nets = [build_nn_from_config(S + A, o, {"hidden_size": H, "hidden_layer": L, "activation_fn": act, "use_layer_norm": True}) for o in (S, 1, 1)]
    for n in nets:                                         # LayerNorm starts at 1 / 0: perturb so that both matter
        for m in n.modules():
            if isinstance(m, T.nn.LayerNorm):
                with T.no_grad():
                    m.weight.add_(0.1 * T.randn_like(m.weight)); m.bias.add_(0.05 * T.randn_like(m.bias))
    theta = np.concatenate([np.concatenate([p.detach().numpy().reshape(-1) for p in n.parameters()]) for n in nets]).astype(np.float32)
    descs_h = tuple(eng.mlp_desc(S + A, H, L, o, act, use_layer_norm=True) for o in (S, 1, 1))
    descs_o = tuple(orc.mlp_desc(S + A, H, L, o, act, use_layer_norm=True) for o in (S, 1, 1))
    assert theta.size == sum(orc.mlp_num_params(d) for d in descs_o)
    rng = np.random.RandomState(3)
    pop = 4
    eps = (rng.randn(pop, theta.size) * 0.05).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    st = rng.randn(3 * pop, S).astype(np.float32)
    ac = rng.randint(0, A, 3 * pop).astype(np.int32)
    ns, r, d = eng.se_step_population(descs_h, dev(theta), dev(eps), dev(worker), dev(sign), dev(st), dev(ac))
    ons, orr, od = orc.se_step_population(descs_o, theta, eps, worker, sign, st, ac)
    assert np.array_equal(ns.cpu().numpy(), ons) and np.array_equal(r.cpu().numpy(), orr) and np.array_equal(d.cpu().numpy(), od)
    # the unperturbed chains (sign 0) against torch
    x = T.from_numpy(np.concatenate([np.eye(A, dtype=np.float32)[ac], st], axis=1))
    with T.no_grad():
        ref = [n(x).numpy() for n in nets]
    rows = np.arange(0, 3 * pop, 3)
    np.testing.assert_allclose(ons[rows], ref[0][rows], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(orr[rows], ref[1][rows, 0], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(od[rows], ref[2][rows, 0], rtol=2e-6, atol=2e-6)


def test_se_step_batched_states(eng, orc, golden):
    # EnvWrapper.step(action, state=batched) (env_wrapper.py:33-40): several states per chain
    g = golden("g1_virtual_env_step")
    S, A, H, L, act = [int(v) for v in g["c00_meta"]]
    theta, state, action = g["c00_theta"], g["c00_state"], g["c00_action"]
    ns, r, d = eng.se_step_population(eng.se_descs(S, A, H, L, ACTS[act]), dev(theta), None, None, None,
                                      dev(state.reshape(2, 6, S)), dev(action.reshape(2, 6)))
    np.testing.assert_allclose(ns.cpu().numpy().reshape(-1, S), g["c00_next_state"], rtol=2e-6, atol=2e-6)
    ons, _, _ = orc.se_step_population(orc.se_descs(S, A, H, L, ACTS[act]), theta, None, None, None, state, action)
    assert np.array_equal(ns.cpu().numpy().reshape(-1, S), ons)


def test_qnet_td_forward(eng, orc, golden):
    g = golden("g4_ddqn_learn")
    for vi in (0, 2):
        pre = "v%d_" % vi
        S, A, H, L, act, B, _ = [int(v) for v in g[pre + "meta"]]
        gamma = float(g[pre + "hparams"][0])
        rows = g[pre + "rows"][0]
        chains, cap, stride = 3, B + 5, (2 * S + 3 + 3) & ~3
        rng = np.random.RandomState(vi)
        replay = np.zeros((chains, cap, stride), np.float32)
        idx = np.zeros((chains, B), np.int32)
        online = np.stack([g[pre + "online0"] + c * 0.01 for c in range(chains)]).astype(np.float32)
        target = np.stack([g[pre + "target0"] - c * 0.01 for c in range(chains)]).astype(np.float32)
        for c in range(chains):
            perm = rng.permutation(cap)[:B]
            replay[c, perm, :2 * S + 3] = rows
            idx[c] = perm
        q_sa, y = eng.qnet_td_forward(eng.mlp_desc(S, H, L, A, ACTS[act]), dev(online), dev(target), dev(replay), dev(idx), gamma)
        for c in range(chains):
            oq, oy, _ = orc.qnet_td_forward(orc.mlp_desc(S, H, L, A, ACTS[act]), online[c], target[c], rows, S, gamma)
            assert np.array_equal(q_sa[c].cpu().numpy(), oq)
            assert np.array_equal(y[c].cpu().numpy(), oy)


def _inner_cfg(orc, cfgd, **over):
    from learning_environments_amd import _lib
    o = orc.ddqn_cfg_from_config(cfgd, **over)
    c = _lib.DdqnCfg()
    for f, _ in _lib.DdqnCfg._fields_:
        setattr(c, f, getattr(o, f, 0))      # (team_size / kernel_variant exist only in the HIP cfg)
    return o, c


@pytest.mark.parametrize("name,chunk", [("g8_calc_score_cartpole_a", 17), ("g8_calc_score_cartpole_b", 17),
                                        ("g8w_calc_score_cartpole_ringwrap", 17),
                                        ("g8l2_calc_score_acrobot_ddqn_2layer", 0),    # Critic_DQN 6-128-128-3 -> GEMM-tiled kernel
                                        ("g8ln_calc_score_acrobot_ddqn_layernorm", 0),  # use_layer_norm: 6-40-40-3 with the LayerNorm behind its second Linear
                                        ("g8seln_calc_score_acrobot_ddqn_se_layernorm", 0),  # the ENV's use_layer_norm: SE nets 9-32-32-x with the LayerNorm
                                        ("g8m_calc_score_mountaincar_ddqn", 0)])       # MountainCar-v0 SE + DDQN 2-48-48-3
def test_inner_loop_tape_mode_vs_reference_and_oracle(eng, orc, golden, name, chunk):
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=chunk, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    chains = 2   # the same chain twice: checks chain indexing of tapes/workspace
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        # bit-exact against the oracle
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"])
        assert np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["state"][c, :n].cpu().numpy(), o["trace"]["state"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 1].cpu().numpy(), o["trace"]["done"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        st = il.stats[c].cpu().tolist()
        assert st == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        # against the reference's own trace / returns (tolerances of test_oracle_golden.py)
        assert np.array_equal(act & 0xFFFF, g["tr_action"])
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(il.episode_test_mean[c].cpu().numpy(), g["reward_list_train"], rtol=0, atol=1e-4)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4   # north_star: returns within 1e-4 of the reference


@pytest.mark.parametrize("env_name,chains,episodes,max_steps,batch,rb", [("CartPole-v0", 9, 4, 40, 199, None), ("CartPole-v0", 6, 3, 25, 64, None),
                                                                        ("Acrobot-v1", 6, 3, 30, 149, None),
                                                                        ("CartPole-v0", 3, 4, 40, 48, 53)])   # replay ring wraps
def test_inner_loop_counter_mode_vs_oracle(eng, orc, golden, env_name, chains, episodes, max_steps, batch, rb):
    g = golden("g8_calc_score_cartpole_a")
    cfgd = json.loads(str(g["config_json"]))
    if env_name == "Acrobot-v1":
        cfgd["env_name"] = env_name
        cfgd["envs"][env_name] = dict(cfgd["envs"]["CartPole-v0"], solved_reward=-100.0, hidden_size=64)
        cfgd["agents"]["ddqn"].update(hidden_size=112, activation_fn="leakyrelu")
    cfgd["agents"]["ddqn"]["batch_size"] = batch
    if rb:
        cfgd["agents"]["ddqn"]["rb_size"] = rb
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=episodes, max_steps=max_steps)
    if ocfg.grad_chunk == 0:
        from learning_environments_amd.config import pick_grad_chunk
        ocfg.grad_chunk = cfg.grad_chunk = pick_grad_chunk(cfg)
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(5)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, ocfg.q_hidden, 1, A, "tanh"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    pop = chains // 3
    eps = (rng.randn(pop, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.4, 0.4, (chains, P_q))).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(7, 3, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=episodes * max_steps)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=episodes * max_steps)
        n = o["trace"]["action"].size
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def test_inner_loop_early_out(eng, orc, golden):
    # solved_reward low enough that the real-env early-out (base_agent.py:141-148) fires after init_episodes
    g = golden("g8_calc_score_cartpole_a")
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=17, rng_mode=0, train_episodes=6, max_steps=20, solved_reward=5.0, early_out_num=2)
    key = orc.chain_key(1, 0, 0, 0)
    il = eng.InnerLoop(cfg, 1)
    il.run(dev(g["theta"]), None, None, None, dev(g["agent_init"][None]), rng_keys=dev(np.array([key], np.uint64).view(np.int64)))
    torch.cuda.synchronize()
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], rng_key=key)
    assert o["episodes_run"] < 6
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert float(il.score[0]) == o["score"]


def test_unsupported_shapes_raise(eng, orc, golden):
    g = golden("g8_calc_score_cartpole_a")
    cfgd = json.loads(str(g["config_json"]))
    _, cfg = _inner_cfg(orc, cfgd, se_layers=4)            # no fused kernel takes an SE with four hidden layers
    with pytest.raises(NotImplementedError):
        il = eng.InnerLoop(cfg, 1)
        il.run(dev(g["theta"]), None, None, None, dev(np.zeros((1, il.p_agent), np.float32)), rng_keys=dev(np.zeros(1, np.int64)))


@pytest.mark.parametrize("se_layers", [2, 3])
def test_inner_loop_multi_layer_se_vs_oracle(eng, orc, golden, se_layers):
    """Synthetic envs whose three nets have more than one hidden layer (envs/virtual_env.py:16-33 builds them with
    build_nn_from_config and the env section's hidden_layer): DDQN chains run in the GEMM-queue kernel's plain-DQN mode with the
    hidden-to-hidden SE layers in the arena; whole chains bit for bit against the oracle (counter mode)."""
    g = golden("g8_calc_score_cartpole_a")
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, se_layers=se_layers, se_hidden=24, train_episodes=4, max_steps=15, test_episodes=3)
    S, A = cfg.state_dim, cfg.num_actions
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, 24, se_layers, "leakyrelu"))
    rng = np.random.RandomState(17)
    chains = 3
    theta = (rng.randn(P_se) * 0.3).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(31, 2, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains)
    assert il.dueling                                        # the register-resident DDQN kernel keeps to one SE hidden layer
    init = rng.uniform(-0.4, 0.4, (chains, il.p_agent)).astype(np.float32)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, init[c], rng_key=int(keys[c]))
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def test_nes_worker_best_and_rank_update(eng, orc, golden):
    g = golden("g7_master")
    g6 = golden("g6_worker_noise")
    # calc_best_score (mirrored): cases from the reference
    cs = np.stack([np.zeros(3), g6["score_add"], g6["score_sub"]], axis=1)
    res = eng.nes_worker_best(dev(cs.reshape(-1)), 3, True).cpu().numpy()
    assert np.array_equal(res[:, 0], g6["score_best"])
    obest, osign = orc.worker_best(g6["score_add"], g6["score_sub"], True)
    assert np.array_equal(res[:, 2], osign.astype(np.float64))
    # score_transform all types + update_env against reference vectors
    from learning_environments_amd.agents.GTN_master import rank_table
    pop = g["scores"].size
    for scores, key in ((g["scores"], ""), (g["tied"], "_tied")):
        gathered = np.zeros((pop, 4))
        gathered[:, 0], gathered[:, 1], gathered[:, 2] = scores, g["scores_orig"], 1.0
        for t in range(8):
            w = eng.nes_rank_update(t, dev(gathered), dev(rank_table(t, pop)), None, None, 0.0).cpu().numpy()
            ow = orc.score_transform(t, scores, g["scores_orig"])
            np.testing.assert_allclose(w, ow, rtol=1e-15, atol=1e-15, err_msg="type %d%s" % (t, key))
            if key == "":
                np.testing.assert_allclose(w, g["tf%d" % t], rtol=1e-15, atol=1e-15)
    gathered = np.zeros((pop, 4))
    gathered[:, 0], gathered[:, 1], gathered[:, 2] = g["scores"], g["scores_orig"], 1.0
    theta = dev(g["theta0"].copy())
    eng.nes_rank_update(3, dev(gathered), dev(rank_table(3, pop)), theta, dev(g["eps"]), float(g["step_size"]))
    assert np.array_equal(theta.cpu().numpy(), g["theta1"])
    eng.nes_rank_update(3, dev(gathered), dev(rank_table(3, pop)), theta, dev(g["eps"]), float(g["step_size"]), True, 0.01)
    assert np.array_equal(theta.cpu().numpy(), g["theta2"])
    # mirrored sign flips eps
    gathered[:, 2] = np.array([1, -1, 1, -1, -1, 1, 1, -1.0])
    theta = dev(g["theta0"].copy())
    w = eng.nes_rank_update(3, dev(gathered), dev(rank_table(3, pop)), theta, dev(g["eps"]), float(g["step_size"])).cpu().numpy()
    ref = orc.update_env(g["theta0"], g["eps"], gathered[:, 2].astype(np.float32), w, float(g["step_size"]))
    assert np.array_equal(theta.cpu().numpy(), ref)


# ---------------------------------------------------------------------------------------------------------------
# config 4: Cliff gridworld + tabular QL + potential-shaped RewardEnv (integer-state path: bit-exact everywhere)
# ---------------------------------------------------------------------------------------------------------------
def _ql_cfgs(orc, cfgd, rng_mode, **over):
    from learning_environments_amd import _lib
    from learning_environments_amd.envs.gridworld import transition_tables
    tables = transition_tables(cfgd["env_name"])
    o = orc.ql_cfg_from_config(cfgd, tables, rng_mode=rng_mode, **over)
    c = _lib.QlCfg()
    for f, _ in _lib.QlCfg._fields_:
        setattr(c, f, getattr(o, f, 0))      # (team_size / kernel_variant exist only in the HIP cfg)
    return o, c, tables


@pytest.mark.parametrize("name", ["g9_calc_score_cliff_a", "g9_calc_score_cliff_b", "g9s_calc_score_cliff_sarsa", "g9c_calc_score_cliff_ql_cb",
                                  "g9sc_calc_score_cliff_sarsa_cb", "g9i_calc_score_cliff_ql_init2",
                                  "g9k_calc_score_cliff_ql_same_action_2", "g9ks_calc_score_cliff_sarsa_same_action_3",     # same_action_num 2 / 3
                                  "g9ln_calc_score_cliff_ql_reward_net_layernorm"])     # the ENV section's use_layer_norm: reward net 48-24-24-1 with the LayerNorm
def test_ql_rn_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    """The tabular agents of select_agent (QL, SARSA, count-based variants; init_episodes gate) against the reference's runs."""
    g = golden(name)
    ocfg, cfg, tables = _ql_cfgs(orc, json.loads(str(g["config_json"])), 1)
    n = g["tr_action"].size
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))), rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))))
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], np.zeros(0, np.int32), np.zeros((0, 4)), np.zeros((0, 4)))
    # (a) the reference's own shaped-reward table as input: trajectories, fp64 Q-table and returns EXACTLY the reference's
    il = eng.QlInnerLoop(cfg, chains, tables, trace_cap=n + 4)
    il.run(dev(g["theta"]), None, None, None, tapes=tapes, shaped_override=dev(g["shaped_ref"].reshape(-1)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, g["tr_action"])
        if cfg.agent_kind == 0:     # (for SARSA the fixture's explored flag also counts the draws of learn's next_action)
            assert np.array_equal(act >> 16, g["tr_explored"])
        assert np.array_equal(il.trace["state"][c, :n, 0].cpu().numpy(), g["tr_state"])
        assert np.array_equal(il.trace["state"][c, :n, 1].cpu().numpy(), g["tr_next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), g["tr_reward"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 1].cpu().numpy(), g["tr_done"])
        assert np.array_equal(il.q_table[c].cpu().numpy().reshape(48, 4), g["q_table"])
        ne = g["reward_list_train"].size
        assert np.array_equal(il.episode_test_mean[c, :ne].cpu().numpy(), g["reward_list_train"])
        assert np.array_equal(il.episode_len[c, :ne].cpu().numpy(), g["episode_length_train"])
        assert np.array_equal(il.final_returns[c].cpu().numpy(), g["reward_list_test"])
        assert float(il.score[c]) == float(g["score"])
    # (b) own reward-net evaluation: bit-exact against the oracle, phi within 2e-6 of the reference
    il2 = eng.QlInnerLoop(cfg, 1, tables, trace_cap=n + 200)
    il2.run(dev(g["theta"]), None, None, None, tapes={k: v[:1].contiguous() for k, v in tapes.items()})
    torch.cuda.synchronize()
    o = orc.ql_rn_chain(ocfg, g["theta"], tables, tapes=otapes, trace_cap=n + 200)
    _, oshaped = orc.rn_shaped_rewards(ocfg, g["theta"], tables)
    assert np.array_equal(il2.shaped[0].cpu().numpy().reshape(48, 4), oshaped)
    np.testing.assert_allclose(il2.shaped[0].cpu().numpy().reshape(48, 4), g["shaped_ref"], rtol=2e-6, atol=2e-6)
    m = o["trace"]["action"].size
    assert np.array_equal(il2.trace["action"][0, :m].cpu().numpy(), o["trace"]["action"])
    assert np.array_equal(il2.q_table[0].cpu().numpy().reshape(48, 4), o["q_table"])
    assert float(il2.score[0]) == o["score"] and int(il2.status[0]) == o["rc"]
    assert il2.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.parametrize("env_name,rtype", [("Cliff", 2), ("HoleRoomLarge", 1), ("WallRoom", 6), ("EmptyRoom33", 0), ("Cliff", 5)])
def test_ql_rn_counter_mode_population_vs_oracle(eng, orc, golden, env_name, rtype):
    g = golden("g9_calc_score_cliff_a")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["env_name"] = env_name
    cfgd["envs"][env_name] = dict(cfgd["envs"]["Cliff"], reward_env_type=rtype)
    cfgd["agents"]["ql"].update(eps_init=0.3, eps_min=0.05, eps_decay=0.9, alpha=0.7, train_episodes=30)
    ocfg, cfg, tables = _ql_cfgs(orc, cfgd, 0)
    N = tables["n_states"]
    P = N * ocfg.rn_hidden + 2 * ocfg.rn_hidden + 1
    rng = np.random.RandomState(3)
    pop = 4
    theta = (rng.randn(P) * 0.3).astype(np.float32)
    eps = (rng.randn(pop, P) * 0.1).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(5, 1, int(worker[c]), c % 3) for c in range(3 * pop)], np.uint64)
    il = eng.QlInnerLoop(cfg, 3 * pop, tables)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    for c in range(3 * pop):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ql_rn_chain(ocfg, w, tables, rng_key=int(keys[c]))
        assert np.array_equal(il.q_table[c].cpu().numpy().reshape(N, 4), o["q_table"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def test_ql_rn_draw_buffers_are_clamped_not_refused(eng, orc, golden):
    """SARSA with a minibatch draws 1 + batch_size exploration decisions per step; their per-episode LDS buffers (12 B each) are a
    speed-up only -- the kernel computes draws past the buffer inline -- so a long episode with a large batch (2 000 steps x 65 draws =
    1.5 MB of buffers, ten times the LDS) must run with clamped buffers, not be refused, and still equal the oracle exactly."""
    g = golden("g9s_calc_score_cliff_sarsa")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["envs"]["Cliff"]["max_steps"] = 2000
    cfgd["agents"]["sarsa"].update(batch_size=64, eps_init=0.6, eps_min=0.3, eps_decay=0.95, alpha=0.5, train_episodes=6)
    ocfg, cfg, tables = _ql_cfgs(orc, cfgd, 0)
    assert cfg.agent_kind == 1 and cfg.max_steps * (1 + cfg.batch_size) * 12 > 160 * 1024
    N = tables["n_states"]
    P = N * ocfg.rn_hidden + 2 * ocfg.rn_hidden + 1
    rng = np.random.RandomState(9)
    theta = (rng.randn(P) * 0.3).astype(np.float32)
    eps = (rng.randn(2, P) * 0.1).astype(np.float32)
    worker = np.repeat(np.arange(2), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    keys = np.array([orc.chain_key(6, 2, int(worker[c]), c % 3) for c in range(6)], np.uint64)
    il = eng.QlInnerLoop(cfg, 6, tables)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * 6
    for c in range(6):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ql_rn_chain(ocfg, w, tables, rng_key=int(keys[c]))
        assert np.array_equal(il.q_table[c].cpu().numpy().reshape(N, 4), o["q_table"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.parametrize("layers,env_ln", [(2, False), (3, False), (2, True), (3, True)])
def test_ql_rn_multi_layer_reward_net_vs_oracle(eng, orc, golden, layers, env_ln):
    """Grid reward nets with more than one hidden layer (default_config_gridworld_reward_env.yaml:108-115 ships HoleRoomLarge with
    hidden_layer 2): the shaped-reward table of every perturbation and whole QL chains, bit for bit against the oracle.  env_ln: with
    `use_layer_norm` in the env's section (cfg.rn_layer_norm; theta keeps its Linear-only size)."""
    g = golden("g9_calc_score_cliff_a")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["env_name"] = "HoleRoomLarge"
    cfgd["envs"]["HoleRoomLarge"] = dict(cfgd["envs"]["Cliff"], reward_env_type=2, hidden_layer=layers, hidden_size=32, max_steps=30, use_layer_norm=env_ln)
    cfgd["agents"]["ql"].update(eps_init=0.3, eps_min=0.05, eps_decay=0.9, alpha=0.7, train_episodes=20)
    ocfg, cfg, tables = _ql_cfgs(orc, cfgd, 0)
    assert cfg.rn_layers == layers and cfg.rn_layer_norm == ocfg.rn_layer_norm == int(env_ln)
    N, H = tables["n_states"], ocfg.rn_hidden
    P = N * H + H + (layers - 1) * (H * H + H) + H + 1
    rng = np.random.RandomState(13)
    pop = 2
    theta = (rng.randn(P) * 0.3).astype(np.float32)
    eps = (rng.randn(pop, P) * 0.1).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(15, 1, int(worker[c]), c % 3) for c in range(3 * pop)], np.uint64)
    il = eng.QlInnerLoop(cfg, 3 * pop, tables)
    assert il.p_theta == P
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), rng_keys=dev(keys.view(np.int64)))
    phi, shaped = eng.rn_shape_population(cfg, dev(theta), dev(eps), dev(worker), dev(sign), dev(tables["next_state"].astype(np.int32)),
                                          dev(tables["reward"].astype(np.float64)), 3 * pop)
    torch.cuda.synchronize()
    A = tables["n_actions"]
    for c in range(3 * pop):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        ophi, oshaped = orc.rn_shaped_rewards(ocfg, w, tables)
        assert np.array_equal(phi[c].cpu().numpy(), ophi) and np.array_equal(shaped[c].cpu().numpy().reshape(N, A), oshaped)
        o = orc.ql_rn_chain(ocfg, w, tables, rng_key=int(keys[c]))
        assert np.array_equal(il.q_table[c].cpu().numpy().reshape(N, A), o["q_table"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


# ---------------------------------------------------------------------------------------------------------------
# config 3: DuelingDDQN on a synthetic environment (LDS-tiled GEMM kernel, parameters in the HBM arena)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g8d_calc_score_acrobot_dueling", "g8df_calc_score_acrobot_dueling_fullshape",
                                  "g8dln_calc_score_acrobot_dueling_layernorm"])      # use_layer_norm: feature stream 6-24-24-24-16, ONE LayerNorm at two positions
def test_dueling_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.agent_kind == 1
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    assert il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        # the reference's own run
        assert np.array_equal(act & 0xFFFF, g["tr_action"])
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=1e-5, atol=1e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("env_name,layers,hidden,feat,batch,act", [("Acrobot-v1", 2, 128, 128, 128, "relu"), ("CartPole-v0", 1, 40, 24, 50, "tanh"),
                                                                   ("Acrobot-v1", 2, 33, 17, 77, "leakyrelu"),
                                                                   ("CartPole-v0", 1, 24, 16, 20, "relu")])      # rb_size 23: ring wraps
def test_dueling_counter_mode_vs_oracle(eng, orc, golden, env_name, layers, hidden, feat, batch, act):
    g = golden("g8d_calc_score_acrobot_dueling")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["env_name"] = env_name
    cfgd["envs"][env_name] = dict(cfgd["envs"]["Acrobot-v1"], hidden_size=48)
    cfgd["agents"]["duelingddqn"].update(hidden_size=hidden, hidden_layer=layers, feature_dim=feat, batch_size=batch, activation_fn=act,
                                         test_episodes=4)
    if batch == 20:
        cfgd["agents"]["duelingddqn"]["rb_size"] = 23
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=3, max_steps=12)
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(8)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.dueling_num_params(ocfg)
    chains = 3
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.15, 0.15, (chains, P_q))).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(9, 2, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=40, want_final_online=True)
    assert il.p_agent == P_q
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=40)
        n = o["trace"]["action"].size
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.parametrize("kind,layers,hidden,feat,batch,act", [("duelingddqn", 2, 48, 24, 40, "relu"), ("duelingddqn", 3, 33, 17, 77, "tanh"),
                                                                ("ddqn", 2, 64, 0, 50, "leakyrelu"), ("ddqn", 3, 40, 0, 32, "relu"),
                                                                ("ddqn", 1, 40, 0, 32, "relu")])      # one hidden layer: the flag changes nothing
def test_layer_norm_in_the_ddqn_loops_vs_oracle(eng, orc, golden, kind, layers, hidden, feat, batch, act):
    """`use_layer_norm: True` in the agent's section (models/model_utils.py:22-37): the shared LayerNorm behind hidden Linear 2..L of the
    Q-net / the DuelingDDQN feature stream, forward AND backward inside the fused loop (lenv_ln.cuh row routines between the queued layer
    products), its weight | bias trained by the same Adam pass.  Counter mode, three chains, against the oracle (which reproduces the
    reference runs G8LN / G8DLN): step traces, returns, counters and ALL final online parameters incl. the LayerNorm's, bit for bit."""
    g = golden("g8d_calc_score_acrobot_dueling")
    cfgd = json.loads(str(g["config_json"]))
    if kind == "ddqn":
        cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
        cfgd["agents"]["ddqn"] = dict(cfgd["agents"]["duelingddqn"])
        cfgd["agents"]["ddqn"].pop("feature_dim", None)
    cfgd["agents"][kind].update(hidden_size=hidden, hidden_layer=layers, batch_size=batch, activation_fn=act, test_episodes=3, use_layer_norm=True)
    if kind == "duelingddqn":
        cfgd["agents"][kind]["feature_dim"] = feat
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0 if (layers > 1 or kind == "duelingddqn") else 4, rng_mode=0, train_episodes=3, max_steps=14)
    assert cfg.q_layer_norm == 1 and ocfg.q_layer_norm == 1
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(28)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.dueling_num_params(ocfg) if kind == "duelingddqn" else orc.mlp_num_params(orc.mlp_desc(S, hidden, layers, A, act, use_layer_norm=1))
    plain_P = (S * hidden + hidden) + (layers - 1) * (hidden * hidden + hidden) + (A * hidden + A)
    if kind == "ddqn":
        assert P_q == plain_P + (2 * hidden if layers >= 2 else 0)
    chains = 3
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.15, 0.15, (chains, P_q))).astype(np.float32)
    if layers >= 2:                                        # nn.LayerNorm starts at weight 1 / bias 0; perturbed here so that both matter
        off = (S * hidden + hidden) + (hidden * hidden + hidden)
        agent_init[:, off:off + hidden] = 1.0 + 0.1 * rng.randn(chains, hidden).astype(np.float32)
        agent_init[:, off + hidden:off + 2 * hidden] = 0.05 * rng.randn(chains, hidden).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(19, 2, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=48, want_final_online=True)
    assert il.p_agent == P_q
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=48, want_final_online=True)
        n = o["trace"]["action"].size
        assert o["learn_steps"] >= 20
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.final_online[c].cpu().numpy(), o["final_online"]), c
        assert not np.array_equal(o["final_online"], agent_init[c])


@pytest.mark.parametrize("kind,se_layers,se_hidden,se_act,q_layers,q_ln", [("ddqn", 2, 32, "leakyrelu", 1, False), ("ddqn", 3, 24, "tanh", 2, True),
                                                                           ("duelingddqn", 2, 48, "relu", 1, False)])
def test_layer_norm_in_the_synthetic_env_of_the_ddqn_loops_vs_oracle(eng, orc, golden, kind, se_layers, se_hidden, se_act, q_layers, q_ln):
    """`use_layer_norm: True` in the ENV's section (virtual_env.py:16-33 builds the three SE nets with build_nn_from_config): the SE step
    inside the fused loop normalises behind hidden Linear 2..L.  NES perturbs nn.Linear modules only (GTN_worker.py:156-175), so theta is
    the Linear parameters and the module's weight 1 / bias 0 never move (the oracle reproduces the reference run G8SELN with exactly that).
    Counter mode, three chains with + / - / 0 perturbation: step traces, returns, counters and final parameters bit for bit."""
    g = golden("g8d_calc_score_acrobot_dueling")
    cfgd = json.loads(str(g["config_json"]))
    if kind == "ddqn":
        cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
        cfgd["agents"]["ddqn"] = dict(cfgd["agents"]["duelingddqn"])
        cfgd["agents"]["ddqn"].pop("feature_dim", None)
    cfgd["agents"][kind].update(hidden_size=40, hidden_layer=q_layers, batch_size=24, test_episodes=3, use_layer_norm=q_ln)
    cfgd["envs"]["Acrobot-v1"].update(hidden_size=se_hidden, hidden_layer=se_layers, activation_fn=se_act, use_layer_norm=True)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=3, max_steps=14)
    assert cfg.se_layer_norm == 1 and ocfg.se_layer_norm == 1 and cfg.q_layer_norm == int(q_ln)
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(61)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, se_hidden, se_layers, se_act))       # Linear parameters only
    K = S + A
    assert P_se == 3 * (K * se_hidden + se_hidden + (se_layers - 1) * (se_hidden * se_hidden + se_hidden)) + (S + 2) * (se_hidden + 1)
    chains = 3
    theta = (rng.randn(P_se) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    il = eng.InnerLoop(cfg, chains, trace_cap=48, want_final_online=True)
    agent_init = (rng.uniform(-0.15, 0.15, (chains, il.p_agent))).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(23, 1, 0, c) for c in range(chains)], np.uint64)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    ocfg_plain = copy.copy(ocfg)
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=48, want_final_online=True)
        n = o["trace"]["action"].size
        assert o["learn_steps"] >= 6
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.final_online[c].cpu().numpy(), o["final_online"]), c
    # and the normalisation is what made the difference: the same chain through plain SE nets walks elsewhere
    ocfg_plain.se_layer_norm = 0
    o0 = orc.ddqn_se_chain(ocfg_plain, theta, agent_init[0], rng_key=int(keys[0]), trace_cap=48)
    assert not np.array_equal(o0["trace"]["next_state"][:4], il.trace["next_state"][0, :4].cpu().numpy())


@pytest.mark.parametrize("env_name,layers,hidden,batch,act,T", [("Acrobot-v1", 2, 128, 128, "relu", 3), ("CartPole-v0", 2, 64, 64, "tanh", 4),
                                                                 ("CartPole-v0", 2, 33, 50, "leakyrelu", 70),      # test rows > batch rows
                                                                 ("Acrobot-v1", 2, 40, 20, "relu", 2),            # rb_size 23: ring wraps
                                                                 ("MountainCar-v0", 2, 256, 128, "relu", 3)])     # default_config_mountaincar.yaml's Q-net
def test_ddqn_multilayer_counter_mode_vs_oracle(eng, orc, golden, env_name, layers, hidden, batch, act, T):
    """DDQN whose Critic_DQN has hidden_layer >= 2 (default_config_acrobot.yaml: 6-128-128-3) runs in the GEMM-tiled kernel's
    plain-DQN mode: bit-exact against the oracle's DDQN with one sequential batch gradient."""
    from learning_environments_amd.config import ddqn_cfg_from_config
    g = golden("g8l2_calc_score_acrobot_ddqn_2layer")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["env_name"] = env_name
    cfgd["envs"][env_name] = dict(cfgd["envs"]["Acrobot-v1"], hidden_size=48)
    cfgd["agents"]["ddqn"].update(hidden_size=hidden, hidden_layer=layers, batch_size=batch, activation_fn=act, test_episodes=T)
    if batch == 20:
        cfgd["agents"]["ddqn"]["rb_size"] = 23
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=3, max_steps=12)
    assert cfg.agent_kind == 0 and cfg.q_layers == layers
    # the package's own cfg builder keeps grad_chunk 0 for these shapes (pick_grad_chunk's second probe)
    assert ddqn_cfg_from_config(cfgd, train_episodes=3, max_steps=12).grad_chunk == 0
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(18)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, hidden, layers, A, act))
    chains = 3
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.15, 0.15, (chains, P_q))).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(19, 2, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=40, want_final_online=True)
    assert il.dueling and il.p_agent == P_q
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=40)
        n = o["trace"]["action"].size
        assert o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


# ---------------------------------------------------------------------------------------------------------------
# synthetic_env_type 1 with a DDQN-family agent: RewardEnv over the real CartPole / Acrobot
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g8r_calc_score_cartpole_ddqn_reward_env", "g8r6_calc_score_cartpole_ddqn_reward_env_t6",
                                  "g8mr_calc_score_mountaincar_ddqn_reward_env",
                                  "g8rl_calc_score_cartpole_ddqn_reward_env_2layer",        # reward net 4-24-24-1
                                  "g8rln_calc_score_cartpole_ddqn_reward_env_layernorm"])   # the same with use_layer_norm in the env's section (type 1)
def test_ddqn_reward_env_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.synthetic_env_type == 1
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    assert il.dueling and il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 1].cpu().numpy(), o["trace"]["done"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(act & 0xFFFF, g["tr_action"]) and np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"])
        np.testing.assert_allclose(il.trace["reward_done"][c, :n, 0].cpu().numpy(), g["tr_reward"], rtol=0, atol=1e-6)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("name,k", [("g8k_calc_score_cartpole_ddqn_same_action_2", 2), ("g8kd_calc_score_acrobot_duelingddqn_same_action_3", 3),
                                    ("g8kr_calc_score_cartpole_ddqn_reward_env_same_action_2", 2)])
def test_ddqn_same_action_num_vs_reference_and_oracle(eng, orc, golden, name, k):
    """same_action_num > 1 in the DDQN family (GEMM-tiled kernel; the register-resident kernel and the wave-chain kernel refuse it and
    the engine routes): the reference's runs replayed (tape mode) and three counter-mode chains, bit-equal to the oracle."""
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.same_action_num == k
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    assert o["rc"] == 0
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    assert il.dueling and il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act & 0xFFFF, g["tr_action"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 1].cpu().numpy(), o["trace"]["done"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]) and np.array_equal(o["episode_len"], g["episode_length_train"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"] and abs(float(il.score[c]) - float(g["score"])) <= 1e-4
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        np.testing.assert_allclose(il.trace["reward_done"][c, :n, 0].cpu().numpy(), g["tr_reward"], rtol=0, atol=2e-6)
    # production RNG, perturbed theta, a step budget that cuts the run short
    ocfg2, cfg2 = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    keys = np.array([5, 6, 2 ** 61 + 7], np.uint64)
    rng = np.random.RandomState(9)
    eps = (rng.randn(1, g["theta"].size) * 0.03).astype(np.float32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    init = np.tile(g["agent_init"], (3, 1))
    il2 = eng.InnerLoop(cfg2, 3, trace_cap=64)
    il2.run(dev(g["theta"]), dev(eps), dev(np.zeros(3, np.int32)), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il2.status.cpu().tolist() == [0, 0, 0]
    for c in range(3):
        w = (np.float32(sign[c]) * eps[0] + g["theta"]).astype(np.float32)
        o2 = orc.ddqn_se_chain(ocfg2, w, init[c], rng_key=int(keys[c]), trace_cap=64)
        m = min(o2["trace"]["action"].size, 64)
        assert np.array_equal(il2.trace["action"][c, :m].cpu().numpy() & 0xFFFF, o2["trace"]["action"][:m])
        assert np.array_equal(il2.trace["reward_done"][c, :m, 0].cpu().numpy(), o2["trace"]["reward"][:m])
        assert np.array_equal(il2.episode_len[c].cpu().numpy(), o2["episode_len"])
        assert float(il2.score[c]) == o2["score"]
        assert il2.stats[c].cpu().tolist() == [o2["episodes_run"], o2["train_steps"], o2["learn_steps"], o2["test_steps"]]


@pytest.mark.parametrize("env_name,family,rtype,act,rn_layers,env_ln",
                         [("CartPole-v0", "ddqn", 2, "prelu", 1, False), ("Acrobot-v1", "duelingddqn", 1, "leakyrelu", 1, False),
                          ("CartPole-v0", "ddqn", 5, "tanh", 1, False), ("Acrobot-v1", "ddqn", 0, "relu", 1, False),
                          ("CartPole-v0", "duelingddqn", 6, "relu", 1, False), ("MountainCar-v0", "ddqn", 1, "leakyrelu", 1, False),
                          ("MountainCar-v0", "duelingddqn", 2, "tanh", 1, False),
                          # reward nets with two / three hidden layers (arena-resident), plain and with the env section's use_layer_norm
                          ("CartPole-v0", "ddqn", 2, "prelu", 2, False), ("Acrobot-v1", "duelingddqn", 1, "leakyrelu", 3, False),
                          ("CartPole-v0", "ddqn", 6, "tanh", 2, True), ("Acrobot-v1", "duelingddqn", 2, "relu", 3, True),
                          ("MountainCar-v0", "ddqn", 0, "relu", 2, True)])
def test_ddqn_reward_env_counter_mode_vs_oracle(eng, orc, golden, env_name, family, rtype, act, rn_layers, env_ln):
    """All info-free reward types, both real envs, both agent families, perturbed reward networks of one to three hidden layers (with
    and without their LayerNorm): bit-exact against the oracle."""
    g = golden("g8r_calc_score_cartpole_ddqn_reward_env" if family == "ddqn" else "g8ia_calc_score_acrobot_dueling_icm")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["gtn"].update(agent_name="DDQN" if family == "ddqn" else "DuelingDDQN", synthetic_env_type=1)
    cfgd["env_name"] = env_name
    base_env = dict(list(cfgd["envs"].values())[0])
    base_env.update(hidden_size=40, hidden_layer=rn_layers, activation_fn=act, reward_env_type=rtype, info_dim=0, max_steps=14,
                    solved_reward=1e9, use_layer_norm=env_ln)
    cfgd["envs"] = {env_name: base_env}
    cfgd["agents"][family].update(batch_size=20, test_episodes=3, init_episodes=1, hidden_size=24)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=3)
    S = cfg.state_dim
    assert cfg.se_layers == rn_layers and cfg.se_layer_norm == ocfg.se_layer_norm == int(env_ln)
    P_rn = orc.mlp_num_params(orc.mlp_desc(1 if rtype == 0 else S, 40, rn_layers, 1, act))
    chains = 3
    rng = np.random.RandomState(81)
    theta = (rng.randn(P_rn) * 0.3).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.1).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(41, 5, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=50)
    agent_init = (rng.uniform(-0.3, 0.3, (chains, il.p_agent))).astype(np.float32)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=50)
        m = o["trace"]["action"].size
        assert o["rc"] == 0 and o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :m].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.trace["reward_done"][c, :m, 0].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


# ---------------------------------------------------------------------------------------------------------------
# ICM agents: the Intrinsic Curiosity Module trained inside learn() (lenv_dueling_se_inner_loop_icm)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g8i_calc_score_cartpole_ddqn_icm", "g8ia_calc_score_acrobot_dueling_icm"])
def test_icm_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    """select_agent "ddqn_icm" / "duelingddqn_icm": the reference's run replayed on the GPU -- bit-exact against the oracle
    (trace, returns AND the ICM parameters after the last update), actions / score equal to the reference's own run."""
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.icm_enabled == 1
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8, icm_init=g["icm_init"])
    chains = 2
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8)
    assert il.icm and il.dueling and il.p_icm == g["icm_init"].size and il.p_agent == g["agent_init"].size
    il.icm_init.copy_(dev(np.tile(g["icm_init"], (chains, 1))))
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.icm_final[c].cpu().numpy(), o["icm_final"])           # every ICM update, bit for bit
        assert np.array_equal(act & 0xFFFF, g["tr_action"])
        np.testing.assert_allclose(il.icm_final[c].cpu().numpy(), g["icm_final"], rtol=0, atol=2e-7)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("family,env_name,batch,fdim,hid", [("ddqn", "CartPole-v0", 40, 32, 128), ("duelingddqn", "Acrobot-v1", 33, 20, 48)])
def test_icm_counter_mode_vs_oracle(eng, orc, golden, family, env_name, batch, fdim, hid):
    """ICM agents in counter mode at the shipped ICM shapes (feature_dim 32, hidden 128) and at odd ones: fresh ICM parameters
    from the chains' counter RNG (stream 12), three chains, everything bit-exact against the oracle."""
    from learning_environments_amd.agents.nes_common import linear_init_bounds
    from learning_environments_amd.config import icm_layer_dims
    g = golden("g8i_calc_score_cartpole_ddqn_icm" if family == "ddqn" else "g8ia_calc_score_acrobot_dueling_icm")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"][family].update(batch_size=batch, test_episodes=3)
    cfgd["agents"]["icm"].update(feature_dim=fdim, hidden_size=hid)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=3, max_steps=9)
    S, A = cfg.state_dim, cfg.num_actions
    chains = 3
    rng = np.random.RandomState(51)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(29, 3, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains, trace_cap=30)
    agent_init = (rng.uniform(-0.2, 0.2, (chains, il.p_agent))).astype(np.float32)
    keys_t = dev(keys.view(np.int64))
    bounds = torch.from_numpy(linear_init_bounds(icm_layer_dims(cfg))).cuda()
    icm_init = il.draw_icm_init(keys_t, bounds).cpu().numpy()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        oinit = orc.agent_init_from_key(int(keys[c]), orc.icm_layer_dims(ocfg), stream=orc.STREAM_ICM_INIT)
        assert np.array_equal(icm_init[c], oinit), c
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=30, icm_init=oinit)
        m = o["trace"]["action"].size
        assert o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy() & 0xFFFF, o["trace"]["action"]), c
        assert np.array_equal(il.icm_final[c].cpu().numpy(), o["icm_final"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def test_icm_vary_counter_mode_vs_oracle(eng, orc, golden):
    """select_agent "ddqn_icm_vary" = DDQN_vary(icm=True): per-chain hyper-parameters of the agent AND an ICM per chain in one
    launch; each chain equals the oracle chain with its draw, its fresh agent and its fresh ICM."""
    from learning_environments_amd.agents import vary
    from learning_environments_amd.agents.nes_common import linear_init_bounds
    from learning_environments_amd.config import agent_layer_dims, icm_layer_dims
    g = golden("g8i_calc_score_cartpole_ddqn_icm")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN_icm_vary"
    cfgd["agents"]["ddqn_vary"] = {"vary_hp": True}
    cfgd["agents"]["ddqn"].update(test_episodes=3)
    common = dict(rng_mode=0, train_episodes=3, max_steps=9)
    cfg = _vary_max_cfg(orc, cfgd, "ddqn", **common)
    assert cfg.icm_enabled == 1
    chains = 3
    keys = np.array([orc.chain_key(33, 2, 0, c) for c in range(chains)], np.uint64)
    hps = [vary.vary_hyperparameters(cfgd["agents"]["ddqn"], vary.chain_units(int(k))) for k in keys]
    rng = np.random.RandomState(61)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(cfg.state_dim, cfg.num_actions, cfg.se_hidden, 1, "leakyrelu"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    il = eng.InnerLoop(cfg, chains, trace_cap=30, vary=True)
    assert il.icm and il.vary
    il.set_hp([h["lr"] for h in hps], [h["batch_size"] for h in hps], [h["hidden_size"] for h in hps], [h["hidden_layer"] for h in hps])
    keys_t = dev(keys.view(np.int64))
    il.draw_agent_init(keys_t)
    il.draw_icm_init(keys_t, torch.from_numpy(linear_init_bounds(icm_layer_dims(cfg))).cuda())
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), None, rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, **common, **orc.hp_overrides(hps[c]))
        oinit = orc.agent_init_from_key(int(keys[c]), agent_layer_dims(_lib_cfg_copy(ocfg)))
        icm_init = orc.agent_init_from_key(int(keys[c]), orc.icm_layer_dims(ocfg), stream=orc.STREAM_ICM_INIT)
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, oinit, rng_key=int(keys[c]), trace_cap=30, icm_init=icm_init)
        m = o["trace"]["action"].size
        assert o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy() & 0xFFFF, o["trace"]["action"]), (c, hps[c])
        assert np.array_equal(il.icm_final[c].cpu().numpy(), o["icm_final"]), (c, hps[c])
        assert float(il.score[c]) == o["score"], (c, hps[c])


@pytest.mark.parametrize("name", ["g8ts_calc_score_cheetah_td3_virtual_env", "g8tseln_calc_score_cheetah_td3_virtual_env_layernorm"])
def test_td3_virtual_env_tape_and_counter_mode_vs_oracle(eng, orc, golden, name):
    """TD3 on a VirtualEnv (default_config_halfcheetah.yaml: synthetic_env_type 0): (a) the reference run G8TS, (b) counter mode
    with perturbed three-hidden-layer SEs of width 128 (the shipped SE shape); bit-exact against the oracle.  *_layernorm: the same with
    `use_layer_norm` in the ENV's section (cfg.rn_layer_norm: the SE nets normalise behind hidden Linear 2..L, theta stays Linear-only)."""
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _td3_cfgs(orc, cfgd, 1)
    assert cfg.virtual_env == 1 and cfg.rn_layer_norm == ocfg.rn_layer_norm == int(name.endswith("layernorm"))
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4)
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4, want_episode_stats=True)
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=0, atol=2e-6)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4
    # (b) the shipped SE shape: three hidden layers of 128, relu; a done head biased so that episodes end early
    cfgd["envs"]["HalfCheetah-v3"].update(hidden_size=128, hidden_layer=3, activation_fn="relu", max_steps=12)
    cfgd["agents"]["td3"].update(batch_size=24)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 3
    keys = np.array([orc.chain_key(43, 1, 0, c) for c in range(chains)], np.uint64)
    rng = np.random.RandomState(91)
    d_s = orc.mlp_desc(23, 128, 3, 17, "relu"); d_1 = orc.mlp_desc(23, 128, 3, 1, "relu")
    P = orc.mlp_num_params(d_s) + 2 * orc.mlp_num_params(d_1)
    theta = (rng.randn(P) * 0.08).astype(np.float32)
    theta[-1] = 0.45                                       # done_net output bias: the learned done flag hovers around 0.5
    eps = (rng.randn(1, P) * 0.02).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=40, want_episode_stats=True)
    agent_init = (rng.uniform(-0.2, 0.2, (chains, il.p_agent))).astype(np.float32)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    lens = []
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        oo = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=40)
        m = oo["trace"]["reward"].size
        assert oo["rc"] == 0 and oo["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy(), oo["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :m].cpu().numpy(), oo["trace"]["next_state"]), c
        assert np.array_equal(il.episode_len[c].cpu().numpy(), oo["episode_len"]), c
        assert float(il.score[c]) == oo["score"], c
        lens += oo["episode_len"].tolist()
    assert min(lens) < 12 or max(lens) == 12               # (informational) the learned done flag may cut episodes short


def test_td3_icm_tape_and_counter_mode_vs_oracle(eng, orc, golden):
    """TD3(icm=True): (a) the reference run G8TI replayed -- bit-exact against the oracle incl. the ICM parameters, reference
    within tolerance; (b) counter mode at the shipped ICM shapes (32 / 128) with fresh ICMs from the chains' RNG."""
    from learning_environments_amd.agents.nes_common import linear_init_bounds
    from learning_environments_amd.config import icm_layer_dims
    g = golden("g8ti_calc_score_cheetah_td3_icm")
    cfgd = json.loads(str(g["config_json"]))
    ocfg, cfg = _td3_cfgs(orc, cfgd, 1)
    assert cfg.icm_enabled == 1
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4, icm_init=g["icm_init"])
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4)
    assert il.icm and il.p_icm == g["icm_init"].size
    il.icm_init.copy_(dev(np.tile(g["icm_init"], (chains, 1))))
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.icm_final[c].cpu().numpy(), o["icm_final"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        np.testing.assert_allclose(il.icm_final[c].cpu().numpy(), g["icm_final"], rtol=0, atol=2e-7)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4
    # (b) counter mode, shipped ICM shapes
    cfgd["agents"]["icm"].update(feature_dim=32, hidden_size=128)
    cfgd["agents"]["td3"].update(batch_size=40)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 3
    keys = np.array([orc.chain_key(37, 1, 0, c) for c in range(chains)], np.uint64)
    rng = np.random.RandomState(71)
    theta = (rng.randn(g["theta"].size) * 0.2).astype(np.float32)
    eps = (rng.randn(1, theta.size) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=40)
    agent_init = (rng.uniform(-0.2, 0.2, (chains, il.p_agent))).astype(np.float32)
    keys_t = dev(keys.view(np.int64))
    icm_init = il.draw_icm_init(keys_t, torch.from_numpy(linear_init_bounds(icm_layer_dims(cfg))).cuda()).cpu().numpy()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        oinit = orc.agent_init_from_key(int(keys[c]), orc.icm_layer_dims(ocfg), stream=orc.STREAM_ICM_INIT)
        assert np.array_equal(icm_init[c], oinit), c
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        oo = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=40, icm_init=oinit)
        m = oo["trace"]["reward"].size
        assert oo["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy(), oo["trace"]["action"]), c
        assert np.array_equal(il.icm_final[c].cpu().numpy(), oo["icm_final"]), c
        assert float(il.score[c]) == oo["score"], c


@pytest.mark.parametrize("env_name,fx", [("Pendulum-v0", "g8pr_calc_score_pendulum_td3_reward_env"),
                                         ("MountainCarContinuous-v0", "g8cr_calc_score_cmc_td3_reward_env")])
def test_td3_icm_vary_on_the_other_continuous_envs_vs_oracle(eng, orc, golden, env_name, fx):
    """td3_icm (one-dimensional actions: the inverse model predicts a single torque / force) and per-chain hyper-parameters on
    Pendulum-v0 and MountainCarContinuous-v0 (same_action_num 2 there): bit-exact against the oracle, ICM parameters included."""
    from learning_environments_amd.config import icm_layer_dims
    from learning_environments_amd.agents.nes_common import linear_init_bounds
    cfgd = json.loads(str(golden(fx)["config_json"]))
    cfgd["agents"]["gtn"]["agent_name"] = "td3_icm"
    cfgd["agents"]["icm"] = {"lr": 1e-3, "beta": 0.2, "eta": 0.5, "feature_dim": 24, "hidden_size": 40}
    cfgd["agents"]["td3"].update(hidden_size=48, hidden_layer=2, batch_size=40, train_episodes=3, init_episodes=1, test_episodes=2, early_out_num=50)
    cfgd["envs"][env_name].update(max_steps=10, hidden_size=24, hidden_layer=1, reward_env_type=2, solved_reward=1e9)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert cfg.icm_enabled == 1
    chains = 3
    keys = np.array([orc.chain_key(39, 2, 0, c) for c in range(chains)], np.uint64)
    rng = np.random.RandomState(73)
    P_rn = orc.rn_num_params(2, cfg.state_dim, 0, 24, 1)
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    # chains 1 and 2 carry their own (smaller) hyper-parameters: the *_vary form of the launch
    hps = [dict(lr=cfg.lr, batch_size=40, hidden_size=48, hidden_layer=2), dict(lr=2e-3, batch_size=17, hidden_size=20, hidden_layer=1),
           dict(lr=5e-4, batch_size=33, hidden_size=31, hidden_layer=2)]
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=40, vary=True)
    il.set_hp([h["lr"] for h in hps], [h["batch_size"] for h in hps], [h["hidden_size"] for h in hps], [h["hidden_layer"] for h in hps])
    keys_t = dev(keys.view(np.int64))
    init = il.draw_agent_init(keys_t).cpu().numpy()
    icm_init = il.draw_icm_init(keys_t, torch.from_numpy(linear_init_bounds(icm_layer_dims(cfg))).cuda()).cpu().numpy()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), None, rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    from learning_environments_amd.config import td3_layer_dims
    for c in range(chains):
        h = hps[c]
        oc, pc = _td3_cfgs(orc, cfgd, 0, lr=float(h["lr"]), batch_size=int(h["batch_size"]), hidden=int(h["hidden_size"]),
                           layers=max(1, int(h["hidden_layer"])))
        oinit = orc.agent_init_from_key(int(keys[c]), td3_layer_dims(pc))
        assert np.array_equal(init[c, :oinit.size], oinit), c
        oicm = orc.agent_init_from_key(int(keys[c]), orc.icm_layer_dims(oc), stream=orc.STREAM_ICM_INIT)
        assert np.array_equal(icm_init[c], oicm), c
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        oo = orc.td3_rn_chain(oc, w, oinit, rng_key=int(keys[c]), trace_cap=40, icm_init=oicm)
        m = oo["trace"]["reward"].size
        assert oo["rc"] == 0 and oo["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy(), oo["trace"]["action"]), c
        assert np.array_equal(il.trace["reward"][c, :m].cpu().numpy(), oo["trace"]["reward"]), c
        assert np.array_equal(il.icm_final[c].cpu().numpy(), oo["icm_final"]), c
        assert float(il.score[c]) == oo["score"], c


# ---------------------------------------------------------------------------------------------------------------
# *_vary agents: per-chain lr / batch_size / hidden_size / hidden_layer in ONE launch (lenv_dueling_se_inner_loop_hp)
# ---------------------------------------------------------------------------------------------------------------
def _vary_max_cfg(orc, cfgd, agent_key, **over):
    from learning_environments_amd.agents import vary
    bd = vary.hp_bounds(cfgd["agents"][agent_key])
    return _inner_cfg(orc, cfgd, grad_chunk=0, batch_size=bd["batch_size"][1], q_hidden=bd["hidden_size"][1],
                      q_layers=max(1, bd["hidden_layer"][1]), **over)[1]


@pytest.mark.parametrize("name", ["g8v_calc_score_cartpole_ddqn_vary", "g8v2_calc_score_cartpole_ddqn_vary_wide",
                                  "g8vd_calc_score_acrobot_dueling_vary"])
def test_vary_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    """The reference's DDQN_vary / DuelingDDQN_vary runs (hyper-parameters drawn through the ConfigSpace stand-in, recorded in
    the fixture) replayed in a launch sized for the LARGEST possible draw: chain 0 carries the recorded draw, chain 1 the
    family's base hyper-parameters.  Products with more than 128 rows / columns run block by block."""
    g = golden(name)
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["hp_json"]))
    agent_key = cfgd["agents"]["gtn"]["agent_name"].lower()[:-5]
    base = cfgd["agents"][agent_key]
    common = dict(rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, **common, **orc.hp_overrides(hp))
    cfg = _vary_max_cfg(orc, cfgd, agent_key, **common)
    assert cfg.batch_size >= 3 * base["batch_size"] - 2 and cfg.q_layers == base["hidden_layer"] + 1
    n = g["tr_action"].size
    otapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.ddqn_se_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 8)
    chains = 2
    il = eng.InnerLoop(cfg, chains, trace_cap=n + 8, vary=True, want_final_online=True)
    il.set_hp([hp["lr"]] * 2, [hp["batch_size"]] * 2, [hp["hidden_size"]] * 2, [hp["hidden_layer"]] * 2)
    p_c = il.chain_num_params(hp["hidden_size"], hp["hidden_layer"])
    assert p_c == g["agent_init"].size and p_c <= il.p_agent
    init = np.full((chains, il.p_agent), np.nan, np.float32)            # the unused tail of a row is never read
    init[:, :p_c] = g["agent_init"]
    tapes = dict(eps_uniform=dev(np.tile(g["tape_eps_uniform"], (chains, 1))),
                 rand_action=dev(np.tile(g["tape_rand_action"], (chains, 1))),
                 replay_idx=dev(np.tile(g["tape_replay_idx"].reshape(1, -1), (chains, 1))),
                 train_reset=dev(np.tile(g["tape_train_reset"][None], (chains, 1, 1))),
                 test_reset=dev(np.tile(g["tape_test_reset"][None], (chains, 1, 1))))
    il.run(dev(g["theta"]), None, None, None, dev(init), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        act = il.trace["action"][c, :n].cpu().numpy()
        assert np.array_equal(act & 0xFFFF, o["trace"]["action"]) and np.array_equal(act >> 16, o["trace"]["explored"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward_done"][c, :n, 0].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(act & 0xFFFF, g["tr_action"])
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=1e-5, atol=1e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4
    # a chain whose draw exceeds the maxima the launch was sized for reports status -8 and leaves the others alone
    il.hp["q_hidden"][1] = cfg.q_hidden + 1
    il.run(dev(g["theta"]), None, None, None, dev(init), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0, -8] and float(il.score[0]) == o["score"]


@pytest.mark.parametrize("family", ["ddqn", "duelingddqn"])
def test_vary_counter_mode_heterogeneous_population_vs_oracle(eng, orc, golden, family):
    """Six chains with six different (lr, batch, width, depth) draws in one launch: each equals the oracle chain run with
    that chain's hyper-parameters and the same fresh agent (lenv_dueling_agent_init_hp == nn.Linear default init per shape)."""
    from learning_environments_amd.agents import vary
    from learning_environments_amd.config import agent_layer_dims
    g = golden("g8v_calc_score_cartpole_ddqn_vary" if family == "ddqn" else "g8vd_calc_score_acrobot_dueling_vary")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"][family].update(test_episodes=3)
    common = dict(rng_mode=0, train_episodes=3, max_steps=10)
    cfg = _vary_max_cfg(orc, cfgd, family, **common)
    S, A = cfg.state_dim, cfg.num_actions
    chains = 6
    keys = np.array([orc.chain_key(21, 4, c // 3, c % 3) for c in range(chains)], np.uint64)
    hps = [vary.vary_hyperparameters(cfgd["agents"][family], vary.chain_units(int(k))) for k in keys]
    for k, h in zip(keys, hps):                            # the package's draw == the oracle's restatement, key by key
        o = orc.vary_chain_hp(cfgd["agents"][family], int(k))
        assert all(h[n] == o[n] for n in ("batch_size", "hidden_size", "hidden_layer")) and abs(h["lr"] / o["lr"] - 1) < 1e-14
    assert len({(h["batch_size"], h["hidden_size"]) for h in hps}) == chains
    rng = np.random.RandomState(31)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, "leakyrelu"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(2, P_se) * 0.05).astype(np.float32)
    worker = np.repeat(np.arange(2), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    il = eng.InnerLoop(cfg, chains, trace_cap=30, vary=True)
    il.set_hp([h["lr"] for h in hps], [h["batch_size"] for h in hps], [h["hidden_size"] for h in hps], [h["hidden_layer"] for h in hps])
    keys_t = dev(keys.view(np.int64))
    init = il.draw_agent_init(keys_t).cpu().numpy()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), None, rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        h = hps[c]
        ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, **common, **orc.hp_overrides(h))
        pc = _lib_cfg_copy(ocfg)
        dims = agent_layer_dims(pc)
        oinit = orc.agent_init_from_key(int(keys[c]), dims)
        assert oinit.size == il.chain_num_params(h["hidden_size"], h["hidden_layer"])
        assert np.array_equal(init[c, :oinit.size], oinit), c
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, oinit, rng_key=int(keys[c]), trace_cap=30)
        n = o["trace"]["action"].size
        assert o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy() & 0xFFFF, o["trace"]["action"]), (c, h)
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), (c, h)
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), (c, h)
        assert float(il.score[c]) == o["score"], (c, h)
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def _lib_cfg_copy(ocfg):
    from learning_environments_amd import _lib
    c = _lib.DdqnCfg()
    for f, _ in _lib.DdqnCfg._fields_:
        setattr(c, f, getattr(ocfg, f, 0))
    return c


# ---------------------------------------------------------------------------------------------------------------
# config 5: TD3 on a continuous-state RewardEnv (HalfCheetah stand-in)
# ---------------------------------------------------------------------------------------------------------------
def _td3_cfgs(orc, cfgd, rng_mode, **over):
    from learning_environments_amd import _lib
    o = orc.td3_cfg_from_config(cfgd, rng_mode=rng_mode, **over)
    c = _lib.Td3Cfg()
    for f, _ in _lib.Td3Cfg._fields_:
        setattr(c, f, getattr(o, f, 0))      # (team_size / kernel_variant exist only in the HIP cfg)
    return o, c


@pytest.mark.parametrize("name", ["g8t_calc_score_cheetah_td3", "g8tf_calc_score_cheetah_td3_fullshape",
                                  "g8tln_calc_score_cheetah_td3_layernorm", "g8tln3_calc_score_cheetah_td3_layernorm_3layer"])   # use_layer_norm in the td3 section
def test_td3_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    g = golden(name)
    ocfg, cfg = _td3_cfgs(orc, json.loads(str(g["config_json"])), 1)
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4)
    assert o["rc"] == 0
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4)
    assert il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        # bit-exact against the oracle
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        # against the reference's own run (continuous control: tolerances of tests/test_oracle_golden.py)
        np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(il.trace["reward"][c, :n].cpu().numpy(), g["tr_reward"], rtol=0, atol=5e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4        # north_star bar (measured 2.4e-7)


def test_td3_vary_tape_and_counter_mode_vs_oracle(eng, orc, golden):
    """TD3_vary: (a) the reference run of fixture G8TV (batch 145, width 108, 3 hidden layers, its own lr) replayed in a launch
    sized for the largest possible draw; (b) three chains with their own counter-RNG draws, each equal to the oracle chain
    with that chain's hyper-parameters and fresh agent."""
    from learning_environments_amd.agents import vary
    from learning_environments_amd.config import td3_layer_dims
    g = golden("g8tv_calc_score_cheetah_td3_vary")
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["hp_json"]))
    base = cfgd["agents"]["td3"]
    bd = vary.hp_bounds(base)
    mx = dict(batch_size=bd["batch_size"][1], hidden=bd["hidden_size"][1], layers=bd["hidden_layer"][1])
    ocfg, _ = _td3_cfgs(orc, cfgd, 1, lr=float(hp["lr"]), batch_size=int(hp["batch_size"]), hidden=int(hp["hidden_size"]),
                        layers=max(1, int(hp["hidden_layer"])))
    _, cfg = _td3_cfgs(orc, cfgd, 1, **mx)
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4)
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4, vary=True)
    il.set_hp([hp["lr"]] * 2, [hp["batch_size"]] * 2, [hp["hidden_size"]] * 2, [hp["hidden_layer"]] * 2)
    p_c = il.chain_num_params(hp["hidden_size"], hp["hidden_layer"])
    assert p_c == g["agent_init"].size and p_c < il.p_agent
    init = np.full((chains, il.p_agent), np.nan, np.float32)
    init[:, :p_c] = g["agent_init"]
    il.run(dev(g["theta"]), None, None, None, dev(init), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4
    # (b) counter mode, heterogeneous chains
    chains = 3
    _, cfg = _td3_cfgs(orc, cfgd, 0, **mx)
    keys = np.array([orc.chain_key(23, 1, 0, c) for c in range(chains)], np.uint64)
    hps = [vary.vary_hyperparameters(base, vary.chain_units(int(k))) for k in keys]
    assert len({h["batch_size"] for h in hps}) >= 2          # heterogeneous chains in one launch
    rng = np.random.RandomState(41)
    theta = (rng.randn(g["theta"].size) * 0.2).astype(np.float32)
    eps = (rng.randn(1, theta.size) * 0.05).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=40, vary=True, want_episode_stats=True)
    il.set_hp([h["lr"] for h in hps], [h["batch_size"] for h in hps], [h["hidden_size"] for h in hps], [h["hidden_layer"] for h in hps])
    keys_t = dev(keys.view(np.int64))
    init = il.draw_agent_init(keys_t).cpu().numpy()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), None, rng_keys=keys_t)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        h = hps[c]
        oc, pc = _td3_cfgs(orc, cfgd, 0, lr=float(h["lr"]), batch_size=int(h["batch_size"]), hidden=int(h["hidden_size"]),
                           layers=max(1, int(h["hidden_layer"])))
        oinit = orc.agent_init_from_key(int(keys[c]), td3_layer_dims(pc))
        assert np.array_equal(init[c, :oinit.size], oinit), c
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        oo = orc.td3_rn_chain(oc, w, oinit, rng_key=int(keys[c]), trace_cap=40)
        m = oo["trace"]["reward"].size
        assert oo["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :m].cpu().numpy(), oo["trace"]["action"]), (c, h)
        assert np.array_equal(il.trace["reward"][c, :m].cpu().numpy(), oo["trace"]["reward"]), (c, h)
        assert float(il.score[c]) == oo["score"], (c, h)
        assert il.stats[c].cpu().tolist() == [oo["episodes_run"], oo["train_steps"], oo["learn_steps"], oo["test_steps"]]


@pytest.mark.parametrize("hidden,layers,batch,act,delay,rtype", [(128, 2, 192, "relu", 1, 2), (40, 1, 50, "tanh", 2, 1), (33, 2, 130, "leakyrelu", 3, 6),
                                                                 (24, 1, 32, "relu", 1, 3), (24, 1, 32, "relu", 1, 4), (24, 1, 32, "relu", 2, 7),
                                                                 (24, 1, 32, "tanh", 1, 8), (24, 1, 32, "relu", 1, 101), (24, 1, 32, "relu", 1, 102),
                                                                 (24, 1, 32, "relu", 1, 0), (24, 1, 32, "relu", 1, 5)])
def test_td3_counter_mode_vs_oracle(eng, orc, golden, hidden, layers, batch, act, delay, rtype):
    g = golden("g8t_calc_score_cheetah_td3")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["td3"].update(hidden_size=hidden, hidden_layer=layers, batch_size=batch, activation_fn=act, policy_delay=delay,
                                 train_episodes=3, init_episodes=1, test_episodes=3)
    cfgd["envs"]["HalfCheetah-v3"].update(max_steps=6, hidden_size=128 if hidden == 128 else 24, reward_env_type=rtype)
    if rtype == 5:
        cfgd["agents"]["td3"]["rb_size"] = 11                   # 18 env steps: the replay ring wraps
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_rn = max(1, orc.rn_num_params(rtype, 17, 4, ocfg.rn_hidden, 1))
    assert ocfg.info_dim == 4
    rng = np.random.RandomState(11)
    chains = 3
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.2, 0.2, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(21, 4, 0, c) for c in range(chains)], np.uint64)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=24)
    assert (il.p_actor, il.p_critic) == (Pa, Pc)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=24)
        n = o["trace"]["reward"].size
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"]), c
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.parametrize("hidden,layers,batch,act,delay,env", [(24, 2, 16, "relu", 1, "HalfCheetah-v3"), (40, 3, 33, "tanh", 2, "HalfCheetah-v3"),
                                                               (128, 2, 192, "leakyrelu", 1, "Pendulum-v0"), (24, 1, 16, "relu", 1, "HalfCheetah-v3")])
def test_layer_norm_in_the_td3_loop_vs_oracle(eng, orc, golden, hidden, layers, batch, act, delay, env):
    """`use_layer_norm: True` in the td3 section (models/model_utils.py:22-37): one shared LayerNorm per net (actor, critic_1, critic_2)
    behind its hidden Linear 2..L -- forward (incl. the one-row actor of the env / test steps), backward and the optimizer step inside
    td3_rn_inner_kernel (lenv_ln.cuh row routines between the queued products).  Counter mode, three chains, against the oracle (which
    reproduces the reference runs G8TLN / G8TLN3): step traces, returns, counters and ALL final parameters, bit for bit.  The Pendulum case
    is default_config_pendulum_reward_env.yaml's shape, which without the flag takes the wave-chain kernel; one hidden layer: no position."""
    from learning_environments_amd import configs
    if env == "Pendulum-v0":
        cfgd = configs.pendulum_reward_env_td3(2)
        cfgd["agents"]["td3"].update(train_episodes=3, init_episodes=1, test_episodes=10, use_layer_norm=True)
        cfgd["envs"][env].update(max_steps=8)
        rtype, S, info = 2, 3, 0
    else:
        g = golden("g8t_calc_score_cheetah_td3")
        cfgd = json.loads(str(g["config_json"]))
        cfgd["agents"]["td3"].update(hidden_size=hidden, hidden_layer=layers, batch_size=batch, activation_fn=act, policy_delay=delay,
                                     train_episodes=3, init_episodes=1, test_episodes=3, use_layer_norm=True)
        cfgd["envs"][env].update(max_steps=8, hidden_size=24, reward_env_type=2)
        rtype, S, info = 2, 17, 4
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert cfg.use_layer_norm == 1 and ocfg.use_layer_norm == 1
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_rn = orc.rn_num_params(rtype, S, info, ocfg.rn_hidden, ocfg.rn_layers)
    rng = np.random.RandomState(12)
    chains = 3
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.2, 0.2, (chains, Pa + 2 * Pc)).astype(np.float32)
    if layers >= 2:                                        # the three LayerNorm blocks: around weight 1 / bias 0
        A = ocfg.action_dim
        for base, n_in in ((0, S), (Pa, S + A), (Pa + Pc, S + A)):
            off = base + (n_in * hidden + hidden) + (hidden * hidden + hidden)
            agent_init[:, off:off + hidden] = 1.0 + 0.1 * rng.randn(chains, hidden).astype(np.float32)
            agent_init[:, off + hidden:off + 2 * hidden] = 0.05 * rng.randn(chains, hidden).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(23, 4, 0, c) for c in range(chains)], np.uint64)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=30, want_final_params=True, want_episode_stats=True)
    assert (il.p_actor, il.p_critic) == (Pa, Pc)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=30, want_final_params=True)
        n = o["trace"]["reward"].size
        assert o["learn_steps"] >= 8                       # (the Pendulum case stops early: 8 steps of cost stay above solved_reward)
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"]), c
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.final_params[c].cpu().numpy(), o["final_params"]), c


@pytest.mark.parametrize("name", ["g8pf_calc_score_pendulum_td3_virtual_env_fullshape", "g8hf_calc_score_cheetah_td3_virtual_env_fullshape"])
def test_td3_full_shape_virtual_env_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    """Reference runs of the td3 sections of default_config_pendulum.yaml / default_config_halfcheetah.yaml at their real shapes (batch 256,
    policy_delay 2, VirtualEnv, ten test episodes) replayed in tape mode: bit-equal to the oracle incl. all final parameters, within the
    fixture tolerances of the reference's own numbers."""
    g = golden(name)
    ocfg, cfg = _td3_cfgs(orc, json.loads(str(g["config_json"])), 1)
    assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.policy_delay, cfg.virtual_env, cfg.test_episodes) == (128, 2, 256, 2, 1, 10)
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=cfg.action_dim, S=orc.TD3_STATE_WORDS[cfg.env_id])
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4, want_final_params=True)
    assert o["rc"] == 0
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4, want_episode_stats=True, want_final_params=True)
    assert il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert np.array_equal(il.final_params[c].cpu().numpy(), o["final_params"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(il.final_params[c].cpu().numpy(), g["final_params"], rtol=0, atol=2e-6)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


# ---------------------------------------------------------------------------------------------------------------
# Pendulum-v0 behind the TD3 path (default_config_pendulum.yaml / default_config_pendulum_reward_env.yaml)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g8p_calc_score_pendulum_td3_virtual_env", "g8pr_calc_score_pendulum_td3_reward_env",
                                  "g8trnln_calc_score_pendulum_td3_reward_net_layernorm"])     # the ENV section's use_layer_norm: reward net 3-20-20-1
def test_td3_pendulum_tape_mode_vs_reference_and_oracle(eng, orc, golden, name):
    g = golden(name)
    ocfg, cfg = _td3_cfgs(orc, json.loads(str(g["config_json"])), 1)
    assert (cfg.env_id, cfg.state_dim, cfg.action_dim, cfg.max_action) == (4, 3, 1, 2.0)
    n = g["tr_reward"].size
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=1, S=2)
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4)
    assert o["rc"] == 0
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4, want_episode_stats=True)
    assert il.p_agent == g["agent_init"].size
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains):
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"])
        assert np.array_equal(il.trace["state"][c, :n].cpu().numpy(), o["trace"]["state"])
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"])
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"])
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        # against the reference's own run
        np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(il.trace["reward"][c, :n].cpu().numpy(), g["tr_reward"], rtol=0, atol=5e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("virtual,rtype,rn_layers,act,hidden,layers,batch,env_ln",
                         [(False, 2, 1, "prelu", 128, 2, 256, False), (False, 1, 1, "tanh", 24, 1, 20, False),
                          (False, 2, 2, "prelu", 24, 1, 20, False), (False, 5, 3, "leakyrelu", 24, 2, 20, False),
                          (False, 0, 1, "relu", 24, 1, 20, False), (False, 6, 1, "relu", 33, 2, 40, False),
                          (True, 0, 2, "leakyrelu", 128, 2, 256, False), (True, 0, 1, "tanh", 24, 1, 20, False),
                          # `use_layer_norm` in the ENV's section (cfg.rn_layer_norm): reward nets with 2 / 3 hidden layers, the shipped SE shape
                          (False, 2, 2, "prelu", 24, 1, 20, True), (False, 5, 3, "leakyrelu", 24, 2, 20, True), (True, 0, 2, "leakyrelu", 24, 2, 20, True)])
def test_td3_pendulum_counter_mode_vs_oracle(eng, orc, golden, virtual, rtype, rn_layers, act, hidden, layers, batch, env_ln):
    """Pendulum-v0, RewardEnv types without an info vector and the VirtualEnv (the shipped 4-32-32-x SE shape among them),
    perturbed synthetic envs, full-length 200-step episodes in the first case: bit-exact against the oracle.  env_ln: the env nets carry
    the (never perturbed) LayerNorm behind hidden Linear 2..L; theta / eps keep their Linear-only size."""
    g = golden("g8pr_calc_score_pendulum_td3_reward_env")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["envs"]["Pendulum-v0"]["use_layer_norm"] = env_ln
    cfgd["agents"]["gtn"]["synthetic_env_type"] = 0 if virtual else 1
    long_run = hidden == 128 and not virtual
    cfgd["agents"]["td3"].update(hidden_size=hidden, hidden_layer=layers, batch_size=batch, train_episodes=3, init_episodes=1, test_episodes=3,
                                 early_out_num=50)
    cfgd["envs"]["Pendulum-v0"].update(max_steps=200 if long_run else 12, hidden_size=32, hidden_layer=rn_layers, activation_fn=act,
                                       reward_env_type=rtype, solved_reward=1e9)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert cfg.rn_layer_norm == ocfg.rn_layer_norm == int(env_ln)
    Pa, Pc = orc.td3_param_counts(ocfg)
    if virtual:
        P_rn = orc.mlp_num_params(orc.mlp_desc(4, 32, rn_layers, 3, act)) + 2 * orc.mlp_num_params(orc.mlp_desc(4, 32, rn_layers, 1, act))
    else:
        P_rn = max(1, orc.rn_num_params(rtype, 3, 0, 32, rn_layers))
    rng = np.random.RandomState(13 + rtype)
    chains = 3
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.2, 0.2, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(23, 6, 0, c) for c in range(chains)], np.uint64)
    cap = 3 * (200 if long_run else 12)
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=cap, want_episode_stats=True)
    assert (il.p_actor, il.p_critic) == (Pa, Pc)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains if not long_run else 2):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=cap)
        n = o["trace"]["reward"].size
        assert o["rc"] == 0 and o["learn_steps"] > 0
        assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"]), c
        assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
        assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"]), c
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]), c
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    # reward types that need the step's info vector do not exist on Pendulum
    if not virtual and rtype == 0:
        cfg.reward_env_type, cfg.info_dim = 3, 4
        with pytest.raises(Exception):
            eng.Td3InnerLoop(cfg, chains)


# ---------------------------------------------------------------------------------------------------------------
# MountainCarContinuous-v0 behind the TD3 path (default_config_cmc.yaml / default_config_cmc_reward_env.yaml), same_action_num
# ---------------------------------------------------------------------------------------------------------------
def _td3_compare(il, o, c, n, with_state=True):
    assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"]), c
    if with_state:
        assert np.array_equal(il.trace["state"][c, :n].cpu().numpy(), o["trace"]["state"]), c
    assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"]), c
    assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"]), c
    assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"]), c
    assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True), c
    assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"]), c
    assert float(il.score[c]) == o["score"], c
    assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]], c


@pytest.mark.parametrize("name,near_flag", [("g8c_calc_score_cmc_td3_virtual_env", False), ("g8cr_calc_score_cmc_td3_reward_env", False),
                                            ("g8c_calc_score_cmc_td3_virtual_env", True), ("g8cr_calc_score_cmc_td3_reward_env", True),
                                            ("g8cf_calc_score_cmc_td3_virtual_env_fullshape", False)])      # default_config_cmc.yaml at its real shapes
def test_td3_cmc_tape_mode_vs_reference_and_oracle(eng, orc, golden, name, near_flag):
    """The reference's runs (same_action_num 2) replayed; and the same tapes with every episode reset next to the flag, so that
    training episodes (RewardEnv mode) and test episodes end at the env's own done flag after a step or two -- with the +100."""
    g = golden(name)
    ocfg, cfg = _td3_cfgs(orc, json.loads(str(g["config_json"])), 1)
    assert (cfg.env_id, cfg.state_dim, cfg.action_dim, cfg.same_action_num) == (5, 2, 1, 2)
    n = g["tr_reward"].size
    train_reset, test_reset = g["tape_train_reset"].copy(), g["tape_test_reset"].copy()
    if near_flag:
        train_reset[:] = np.array([0.40, 0.045]) + np.arange(train_reset.shape[0])[:, None] * np.array([0.004, 0.0])
        test_reset[:] = np.array([0.36, 0.05]) + np.arange(test_reset.shape[0])[:, None] * np.array([0.011, 0.001])
    replay_idx = g["tape_replay_idx"] % 2 if near_flag else g["tape_replay_idx"]     # short episodes: keep the recorded indices inside the buffer
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                replay_idx, train_reset, test_reset, A=1, S=2)
    o = orc.td3_rn_chain(ocfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4)
    assert o["rc"] == 0
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(replay_idx.reshape(-1)),
                 train_reset=rep(train_reset), test_reset=rep(test_reset))
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=n + 4, want_episode_stats=True)
    il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    m = o["trace"]["reward"].size
    for c in range(chains):
        _td3_compare(il, o, c, m)
        if not near_flag:
            np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(il.trace["next_state"][c, :n].cpu().numpy(), g["tr_next_state"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(il.trace["reward"][c, :n].cpu().numpy(), g["tr_reward"], rtol=0, atol=5e-5)
            assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4
    if near_flag:
        assert o["final_test_returns"].max() > 90.0                        # test episodes reached the flag
        if not cfg.virtual_env:
            assert o["trace"]["reward"].max() > 50.0 and o["episode_len"].min() < cfg.max_steps      # so did training episodes on the real env


@pytest.mark.parametrize("env_name,virtual,k,rtype", [("MountainCarContinuous-v0", False, 2, 2), ("MountainCarContinuous-v0", True, 2, 0),
                                                      ("MountainCarContinuous-v0", False, 1, 6), ("MountainCarContinuous-v0", False, 3, 1),
                                                      ("Pendulum-v0", False, 2, 2), ("Pendulum-v0", True, 3, 0), ("HalfCheetah-v3", False, 2, 4),
                                                      ("HalfCheetah-v3", True, 2, 0), ("MountainCarContinuous-v0", False, 2, 5)])
def test_td3_same_action_num_counter_mode_vs_oracle(eng, orc, golden, env_name, virtual, k, rtype):
    """same_action_num 1..3 on all three continuous envs, RewardEnv and VirtualEnv: bit-exact against the oracle; max_steps odd, so
    the last action of an episode is cut short by TimeLimit on the real env."""
    fx = {"MountainCarContinuous-v0": "g8cr_calc_score_cmc_td3_reward_env", "Pendulum-v0": "g8pr_calc_score_pendulum_td3_reward_env",
          "HalfCheetah-v3": "g8t_calc_score_cheetah_td3"}[env_name]
    cfgd = json.loads(str(golden(fx)["config_json"]))
    cfgd["agents"]["gtn"]["synthetic_env_type"] = 0 if virtual else 1
    cfgd["agents"]["td3"].update(hidden_size=24, hidden_layer=1, batch_size=20, train_episodes=3, init_episodes=1, test_episodes=3,
                                 same_action_num=k, early_out_num=50)
    ms = 999 if rtype == 5 else 11                    # one case at the shipped episode length of MountainCarContinuous (999 env steps)
    cfgd["envs"][env_name].update(max_steps=ms, hidden_size=24, hidden_layer=1, activation_fn="relu", reward_env_type=rtype, solved_reward=1e9,
                                  info_dim=4 if env_name == "HalfCheetah-v3" else 0)
    if ms == 999:
        cfgd["agents"]["td3"].update(train_episodes=2, test_episodes=2)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert cfg.same_action_num == k
    S, A = cfg.state_dim, cfg.action_dim
    Pa, Pc = orc.td3_param_counts(ocfg)
    if virtual:
        P_rn = orc.mlp_num_params(orc.mlp_desc(S + A, 24, 1, S, "relu")) + 2 * orc.mlp_num_params(orc.mlp_desc(S + A, 24, 1, 1, "relu"))
    else:
        P_rn = max(1, orc.rn_num_params(rtype, S, ocfg.info_dim, 24, 1))
    rng = np.random.RandomState(29 + k)
    chains = 3
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.2, 0.2, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([0.0, 1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(27, 8, 0, c) for c in range(chains)], np.uint64)
    cap = 40 if ms == 11 else 1100
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=cap, want_episode_stats=True)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in range(chains if ms == 11 else 2):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=cap)
        assert o["rc"] == 0 and o["learn_steps"] > 0
        _td3_compare(il, o, c, o["trace"]["reward"].size)
        if not virtual and ms == 11:
            assert o["episode_len"].tolist() == [k * ((11 + k - 1) // k)] * 3 and o["test_steps"] == 4 * 3 * 11


@pytest.mark.parametrize("which", ["pendulum_reward_env", "cmc", "cmc_reward_env"])
def test_td3_other_published_shapes_specialised_vs_generic(eng, orc, which):
    """default_config_pendulum_reward_env.yaml, default_config_cmc.yaml and default_config_cmc_reward_env.yaml have their own
    shape-specialised TD3 instantiations (network and batch shapes, reward type, policy delay, same_action_num as literals): a
    launch without a trace (specialised) and one asking for a trace (generic) agree on every output, one chain equals the oracle."""
    from learning_environments_amd import configs
    make = {"pendulum_reward_env": configs.pendulum_reward_env_td3, "cmc": configs.cmc_syn_env_td3, "cmc_reward_env": configs.cmc_reward_env_td3}[which]
    cfgd = make(num_workers=2, max_iterations=1)
    env_name = cfgd["env_name"]
    cfgd["envs"][env_name]["max_steps"] = 10
    cfgd["agents"]["td3"].update(train_episodes=4, init_episodes=1, early_out_num=50)
    cfgd["envs"][env_name]["solved_reward"] = 1e9
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    chains, pop = 6, 2
    rng = np.random.RandomState(43)
    Pa, Pc = orc.td3_param_counts(ocfg)
    S, A = cfg.state_dim, cfg.action_dim
    e = cfgd["envs"][env_name]
    if cfg.virtual_env:
        P_rn = orc.mlp_num_params(orc.mlp_desc(S + A, e["hidden_size"], e["hidden_layer"], S, e["activation_fn"])) + \
            2 * orc.mlp_num_params(orc.mlp_desc(S + A, e["hidden_size"], e["hidden_layer"], 1, e["activation_fn"]))
    else:
        P_rn = orc.rn_num_params(cfg.reward_env_type, S, 0, e["hidden_size"], e["hidden_layer"])
    theta = (rng.randn(P_rn) * 0.1).astype(np.float32)
    eps = (rng.randn(pop, P_rn) * 0.05).astype(np.float32)
    agent_init = rng.uniform(-0.1, 0.1, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(83, 2, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    outs = []
    for trace_cap in (0, 2):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        outs.append([t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.episode_len, il.final_returns)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    c = 4
    w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
    o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
    assert o["rc"] == 0 and o["learn_steps"] > 0
    assert float(outs[0][0][c]) == o["score"] and outs[0][1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(outs[0][2][c], o["episode_test_mean"], equal_nan=True) and np.array_equal(outs[0][4][c], o["final_test_returns"])


@pytest.mark.parametrize("rtype", [0, 1, 2, 3, 4, 5, 6, 7, 8, 101, 102])
def test_rn_shape_rows_vs_oracle(eng, orc, rtype):
    """RewardEnv._calc_reward for rows of a vector-state env, every reward type: bit-exact against the oracle."""
    rng = np.random.RandomState(rtype)
    S, nI, H, n = 17, 4, 37, 33
    P = max(1, orc.rn_num_params(rtype, S, nI, H, 1))
    theta = (rng.randn(P) * 0.3).astype(np.float32)
    s, s2 = rng.randn(n, S).astype(np.float32), rng.randn(n, S).astype(np.float32)
    info, r = rng.randn(n, nI).astype(np.float32), rng.randn(n).astype(np.float32)
    for act in ("prelu", "tanh"):
        desc = eng.mlp_desc(S + nI if rtype in (3, 4, 7, 8) else S, H, 1, 1, act, 0.25) if 1 <= rtype <= 8 else None
        got = eng.rn_shape_rows(rtype, desc, S, nI, 0.98, dev(theta), dev(s), dev(s2), dev(info), dev(r)).cpu().numpy()
        want = orc.rn_shape_rows(rtype, S, nI, H, 1, act, 0.25, 0.98, theta, s, s2, info, r)
        assert np.array_equal(got, want), (rtype, act)
    if rtype in (3, 4, 7, 8, 101, 102):
        with pytest.raises(ValueError):                       # reward_env.py:96: 'No info dict provided by environment'
            eng.rn_shape_rows(rtype, desc, S, 0, 0.98, dev(theta), dev(s), dev(s2), None, dev(r))
    if 1 <= rtype <= 8:
        # deeper reward nets (default_config_pendulum_reward_env.yaml ships hidden_layer 2) and `use_layer_norm` nets: the LayerNorm's
        # weight | bias sit behind the second Linear (lenv_mlp_desc layout)
        D = S + nI if rtype in (3, 4, 7, 8) else S
        for layers, ln, act in ((2, False, "relu"), (3, False, "leakyrelu"), (2, True, "tanh"), (4, True, "prelu")):
            d_h, d_o = eng.mlp_desc(D, H, layers, 1, act, 0.25, use_layer_norm=ln), orc.mlp_desc(D, H, layers, 1, act, 0.25, use_layer_norm=ln)
            Pl = orc.mlp_num_params(d_o)
            assert Pl == D * H + H + (layers - 1) * (H * H + H) + H + 1 + (2 * H if ln else 0) == eng.mlp_num_params(d_h)
            th = (rng.randn(Pl) * 0.25).astype(np.float32)
            got = eng.rn_shape_rows(rtype, d_h, S, nI, 0.98, dev(th), dev(s), dev(s2), dev(info), dev(r)).cpu().numpy()
            want = orc.rn_shape_rows(rtype, S, nI, H, layers, act, 0.25, 0.98, th, s, s2, info, r, use_layer_norm=ln)
            assert np.array_equal(got, want), (rtype, layers, ln)


def test_rn_unknown_type_raises(eng):
    with pytest.raises(NotImplementedError):                  # reward_env.py:49,58
        eng.rn_num_params(9, 17, 4, 8, 1)


def test_inner_loop_full_size_properties(eng, orc):
    """BASELINE configs[1] at full size (pop 64 = 192 chains, B=199, 200-step episodes, the bench's 20 train episodes):
    size-independent properties of the fused kernel + a spot check of whole chains against the oracle.
      * determinism: two launches are bit-identical;
      * chains are independent: permuting the chain order permutes every output;
      * antithetic symmetry: (sign=+1, eps) == (sign=-1, -eps) bit for bit (GTN_worker.py:165-198)."""
    from learning_environments_amd import configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.cartpole_syn_env_ddqn(64), 20)
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=cfg.grad_chunk, rng_mode=0)
    S, A, pop = cfg.state_dim, cfg.num_actions, 64
    chains = 3 * pop
    rng = np.random.RandomState(11)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, cfg.q_hidden, 1, A, "tanh"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(pop, P_se) * 0.0124).astype(np.float32)
    agent_init = rng.uniform(-0.4, 0.4, (chains, P_q)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(1234, 5, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(theta_, eps_, worker_, sign_, init_, keys_, trace_cap=0):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap)
        il.run(dev(theta_), dev(eps_), dev(worker_), dev(sign_), dev(init_), rng_keys=dev(keys_.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return (il.score.cpu().numpy().copy(), il.stats.cpu().numpy().copy(), il.episode_test_mean.cpu().numpy().copy(),
                il.final_returns.cpu().numpy().copy())

    base = run(theta, eps, worker, sign, agent_init, keys)
    assert base[1][:, 2].min() > 0                                     # every chain trained
    again = run(theta, eps, worker, sign, agent_init, keys)
    for a, b in zip(base, again):
        assert np.array_equal(a, b, equal_nan=True)
    perm = rng.permutation(chains)
    permuted = run(theta, eps, worker[perm], sign[perm], agent_init[perm], keys[perm])
    for a, b in zip(base, permuted):
        assert np.array_equal(a[perm], b, equal_nan=True)
    flipped = run(theta, -eps, worker, -sign, agent_init, keys)          # sign 0 stays 0, +1 <-> -1 with eps negated
    for a, b in zip(base, flipped):
        assert np.array_equal(a, b, equal_nan=True)
    # this shape runs in the shape-specialised instantiation (literal layout, counter RNG, no trace); a launch that asks for a
    # step trace takes the generic instantiation: every output of all 192 chains must agree bit for bit
    generic = run(theta, eps, worker, sign, agent_init, keys, trace_cap=2)
    for a, b in zip(base, generic):
        assert np.array_equal(a, b, equal_nan=True)
    for c in (1, 95):                                                   # whole chains against the oracle at full size
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
        assert float(base[0][c]) == o["score"]
        assert base[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(base[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(base[3][c], o["final_test_returns"])


@pytest.mark.parametrize("env_name,hq,hse,batch,act", [("Acrobot-v1", 112, 64, 149, "leakyrelu"), ("CartPole-v0", 33, 40, 64, "relu"),
                                                      ("CartPole-v0", 57, 83, 199, "tanh")])
def test_inner_loop_with_and_without_trace_agree(eng, orc, golden, env_name, hq, hse, batch, act):
    """A launch without a step trace and the same launch asking for one must agree on every output, chain for chain, for shapes
    other than the published one too (the third case has the published widths but 40-step episodes and 4 test episodes, so it
    does NOT take the shape-specialised instantiation).  A production-mode instantiation without the tape / trace plumbing but
    with the run-time layout was measured for these shapes and brought nothing (9.30 vs 9.30 us per learn step), so it does not exist."""
    cfgd = json.loads(str(golden("g8_calc_score_cartpole_a")["config_json"]))
    if env_name == "Acrobot-v1":
        cfgd["env_name"] = env_name
        cfgd["envs"][env_name] = dict(cfgd["envs"]["CartPole-v0"], solved_reward=-100.0)
    cfgd["envs"][env_name]["hidden_size"] = hse
    cfgd["agents"]["ddqn"].update(hidden_size=hq, batch_size=batch, activation_fn=act, test_episodes=4)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=5, max_steps=40)
    from learning_environments_amd.config import pick_grad_chunk
    ocfg.grad_chunk = cfg.grad_chunk = pick_grad_chunk(cfg)
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(31)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, hse, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, hq, 1, A, act))
    chains, pop = 9, 3
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(pop, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.4, 0.4, (chains, P_q))).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(9, 3, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    outs = []
    for trace_cap in (0, 2):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        assert not il.dueling
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        outs.append([t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.episode_len, il.final_returns, il.final_online)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    c = 5
    w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
    o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
    assert float(outs[0][0][c]) == o["score"] and outs[0][1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.parametrize("env_name,hq,batch,act,split", [("CartPole-v0", 16, 180, "tanh", (28, 4)), ("CartPole-v0", 33, 213, "relu", (127, 2)),
                                                         ("CartPole-v0", 24, 200, "leakyrelu", (88, 2)), ("Acrobot-v1", 40, 190, "tanh", (58, 4)),
                                                         ("CartPole-v0", 57, 199, "relu", (85, 3)), ("CartPole-v0", 10, 199, "tanh", (0, 0))])
def test_inner_loop_split_forward_layouts(eng, orc, golden, env_name, hq, batch, act, split):
    """Minibatches of more than 170 samples spill forward items into the third wave of SIMDs 0 / 1; the kernel cuts those items
    into 2, 3 or 4 parts over the hidden-unit pairs, shares their activations between the four third waves and runs their output
    layers from LDS rows (docs/notebook_r01_r04.md, split layout).  Every cut (and the fall-back for nets too narrow to cut) must
    leave the bits alone: whole chains against the oracle."""
    cfgd = json.loads(str(golden("g8_calc_score_cartpole_a")["config_json"]))
    if env_name == "Acrobot-v1":
        cfgd["env_name"] = env_name
        cfgd["envs"][env_name] = dict(cfgd["envs"]["CartPole-v0"], solved_reward=-100.0)
    cfgd["envs"][env_name]["hidden_size"] = 64
    cfgd["agents"]["ddqn"].update(hidden_size=hq, batch_size=batch, activation_fn=act, test_episodes=3)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=4, max_steps=30)
    from learning_environments_amd.config import pick_grad_chunk
    ocfg.grad_chunk = cfg.grad_chunk = pick_grad_chunk(cfg)
    import ctypes as C
    from learning_environments_amd import _lib
    items, parts = C.c_int32(), C.c_int32()
    assert _lib.lib().lenv_ddqn_se_forward_split(C.byref(cfg), C.byref(items), C.byref(parts)) == 0
    assert (items.value, parts.value) == split                      # the layout this launch will really use
    S, A = ocfg.state_dim, ocfg.num_actions
    hse = cfgd["envs"][env_name]["hidden_size"]
    rng = np.random.RandomState(47)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, hse, 1, cfgd["envs"][env_name]["activation_fn"]))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, hq, 1, A, act))
    chains = 6
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(2, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.4, 0.4, (chains, P_q))).astype(np.float32)
    worker = np.repeat(np.arange(2), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    keys = np.array([orc.chain_key(11, 5, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains)
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    for c in (1, 5):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
        assert o["learn_steps"] > 10
        assert float(il.score[c]) == o["score"]
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.episode_len[c].cpu().numpy()[:o["episode_len"].size], o["episode_len"])
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])


@pytest.mark.parametrize("case", ["published", "acrobot", "narrow"])
def test_ddqn_chain_on_a_team_of_workgroups(eng, orc, golden, case):
    """A DDQN chain on G co-resident workgroups (launches that leave most of the GPU idle): the members deal the minibatch by whole
    gradient micro-chunks, exchange the chunk partials once per learn step and each applies the same Adam step.  Every team size
    must give the bits of the one-workgroup launch (and of the oracle): the published CartPole shape (its own instantiation), an
    Acrobot net with three actions, and a narrow net whose chunking leaves only five micro-chunks to deal."""
    import ctypes as C
    from learning_environments_amd import _lib
    from learning_environments_amd.config import pick_grad_chunk
    cfgd = json.loads(str(golden("g8_calc_score_cartpole_a")["config_json"]))
    env_name = "CartPole-v0"
    if case == "published":
        hq, batch, act, T, max_steps, episodes = 57, 199, "tanh", 10, 200, 3
    elif case == "acrobot":
        env_name = "Acrobot-v1"
        cfgd["env_name"] = env_name
        cfgd["envs"][env_name] = dict(cfgd["envs"]["CartPole-v0"], solved_reward=-100.0)
        hq, batch, act, T, max_steps, episodes = 40, 150, "leakyrelu", 3, 40, 5
    else:
        hq, batch, act, T, max_steps, episodes = 24, 64, "relu", 4, 50, 5
    cfgd["envs"][env_name]["hidden_size"] = 83 if case == "published" else 64
    cfgd["envs"][env_name]["max_steps"] = max_steps
    cfgd["agents"]["ddqn"].update(hidden_size=hq, batch_size=batch, activation_fn=act, test_episodes=T)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=episodes, max_steps=max_steps)
    ocfg.grad_chunk = cfg.grad_chunk = pick_grad_chunk(cfg) if case != "narrow" else 13          # 64 samples in chunks of 13: five chunks
    S, A = ocfg.state_dim, ocfg.num_actions
    rng = np.random.RandomState(53)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfgd["envs"][env_name]["hidden_size"], 1, cfgd["envs"][env_name]["activation_fn"]))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, hq, 1, A, act))
    chains = 11                                           # not a multiple of 8: the grid is rounded up and the extra blocks leave
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(4, P_se) * 0.05).astype(np.float32)
    agent_init = (rng.uniform(-0.4, 0.4, (chains, P_q))).astype(np.float32)
    worker = (np.arange(chains) % 4).astype(np.int32)
    sign = np.array([0.0, 1.0, -1.0] * 4, np.float32)[:chains]
    keys = np.array([orc.chain_key(13, 2, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(G):
        cfg.team_size = G
        got = _lib.lib().lenv_ddqn_se_team_size(C.byref(cfg), chains)
        il = eng.InnerLoop(cfg, chains, want_final_online=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains, (G, il.status.cpu().tolist())
        return got, [t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.episode_len, il.final_returns, il.final_online)]

    g1, base = run(1)
    assert g1 == 1
    n_chunks = -(-batch // cfg.grad_chunk)
    for G in (2, 3, 4, 6):
        got, outs = run(G)
        assert got == max(x for x in (1, 2, 3, 4, 6) if x <= G and x <= n_chunks), (G, got, n_chunks)
        for a, b in zip(base, outs):
            assert np.array_equal(a, b, equal_nan=True), (case, G)
    # teams of four / six cut every forward item over the idle lanes where that fits (the TWIDE instantiation); the same launches with whole
    # items per lane (kernel_variant TEAM_NARROW) must give the same bits
    cfg.kernel_variant = _lib.VARIANT_TEAM_NARROW
    for G in (4, 6):
        _, outs = run(G)
        for a, b in zip(base, outs):
            assert np.array_equal(a, b, equal_nan=True), (case, G, "narrow")
    cfg.kernel_variant = 0
    cfg.team_size = 0
    assert _lib.lib().lenv_ddqn_se_team_size(C.byref(cfg), chains) == 1          # fewer than 16 chains: no team unless asked for
    assert _lib.lib().lenv_ddqn_se_team_size(C.byref(cfg), 96) == 2 and _lib.lib().lenv_ddqn_se_team_size(C.byref(cfg), 192) == 1
    for c in (2, 10):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
        assert o["learn_steps"] > 10
        assert float(base[0][c]) == o["score"]
        assert base[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(base[2][c], o["episode_test_mean"], equal_nan=True)


@pytest.mark.parametrize("pop,team", [(32, 2), (16, 4), (8, 6)])
def test_ddqn_team_full_size_generations_bit_equal(pop, team):
    """The strong-scaling shards of BASELINE configs[1] at full length (20 x 200 train steps = 3 800 learn steps per chain, through
    GTN_Master and its HIP graph): two generations with one workgroup per chain and with the automatic team size must leave the same
    scores, counters, per-episode test means, final returns and theta, bit for bit."""
    import ctypes as C
    import bench
    from learning_environments_amd import _lib
    outs = []
    # "narrow": the automatic team with whole forward items per lane (kernel_variant TEAM_NARROW) -- teams of four and six otherwise cut
    # every item over the idle lanes (the TWIDE instantiation: three parts / six parts; the launch bench.py's strong-scaling shard record times)
    for mode in ("1", "auto") + (("narrow",) if team >= 4 else ()):
        m, _ = bench.build_master(pop, team_size=1 if mode == "1" else 0)
        if mode == "narrow":
            m.cfg.kernel_variant = _lib.VARIANT_TEAM_NARROW
        assert _lib.lib().lenv_ddqn_se_team_size(C.byref(m.cfg), 3 * pop) == (1 if mode == "1" else team)
        m.step(0)
        m.step(1)
        torch.cuda.synchronize()
        assert m.inner.status.cpu().abs().max().item() == 0
        outs.append([t.cpu().numpy().copy() for t in (m.inner.score, m.inner.stats, m.inner.episode_test_mean, m.inner.final_returns, m.theta)])
    assert int(outs[0][1][0][2]) == 3800
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("which", ["cartpole", "acrobot_syn_env", "mountaincar"])
def test_inner_loop_other_published_shapes_specialised_vs_generic(eng, orc, which):
    """default_config_cartpole.yaml (Critic_DQN 4-64-2 relu, batch 32, SE hidden 128, one test episode) and
    default_config_acrobot_syn_env.yaml (6-112-3 leakyrelu, batch 149, SE hidden 167 prelu, 500-step episodes) have their own
    shape-specialised instantiations: outputs equal to the generic build's (a launch asking for a trace) and to the oracle's."""
    from learning_environments_amd import configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.cartpole_syn_env_ddqn(4)
    if which == "cartpole":
        cfgd["envs"]["CartPole-v0"].update(hidden_size=128)
        cfgd["agents"]["ddqn"].update(hidden_size=64, batch_size=32, activation_fn="relu", test_episodes=1, train_episodes=12, init_episodes=1)
        expect = (64, 128, 32, 1, 200)
    elif which == "mountaincar":                           # default_config_mountaincar.yaml: DDQN 2-256-256-3 in the GEMM-tiled kernel
        cfgd = configs.mountaincar_syn_env_ddqn(4)
        cfgd["agents"]["ddqn"].update(train_episodes=3, init_episodes=1)
        cfgd["envs"]["MountainCar-v0"]["max_steps"] = 60
        expect = (256, 128, 128, 10, 60)
    else:
        cfgd["env_name"] = "Acrobot-v1"
        cfgd["envs"] = {"Acrobot-v1": {"solved_reward": -100.0, "max_steps": 500, "activation_fn": "prelu", "hidden_size": 167, "hidden_layer": 1,
                                       "info_dim": 0, "reward_env_type": 0}}
        cfgd["agents"]["ddqn"].update(hidden_size=112, batch_size=149, activation_fn="leakyrelu", test_episodes=10, train_episodes=3, init_episodes=1)
        expect = (112, 167, 149, 10, 500)
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=cfg.grad_chunk, rng_mode=0)
    assert (cfg.q_hidden, cfg.se_hidden, cfg.batch_size, cfg.test_episodes, cfg.max_steps) == expect
    S, A, pop = cfg.state_dim, cfg.num_actions, 4
    chains = 3 * pop
    rng = np.random.RandomState(19)
    se_act = "prelu" if which == "acrobot_syn_env" else "leakyrelu"
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, se_act))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, cfg.q_hidden, cfg.q_layers, A, cfgd["agents"]["ddqn"]["activation_fn"]))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(pop, P_se) * 0.02).astype(np.float32)
    agent_init = rng.uniform(-0.3, 0.3, (chains, P_q)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(79, 4, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    outs = []
    for trace_cap in (0, 2):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        assert il.dueling == (which == "mountaincar")
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        outs.append([t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.episode_len, il.final_returns, il.final_online)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    c = 7
    w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
    o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
    assert o["learn_steps"] > 0
    assert float(outs[0][0][c]) == o["score"] and outs[0][1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(outs[0][2][c], o["episode_test_mean"], equal_nan=True) and np.array_equal(outs[0][4][c], o["final_test_returns"])


@pytest.mark.parametrize("budget", [0, 9000])
def test_inner_loop_specialised_vs_generic_instantiation(eng, orc, budget):
    """The published CartPole shape in its production form (early-out on, as default_config_cartpole_syn_env.yaml ships it; with and
    without a step budget): the shape-specialised instantiation (no trace) and the generic one (a launch that asks for a trace)
    must agree on every output of every chain, and one chain is checked against the oracle."""
    from learning_environments_amd import configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.cartpole_syn_env_ddqn(8)
    cfgd["agents"]["ddqn"].update(train_episodes=30)
    if budget:
        cfgd["agents"]["ddqn"]["step_budget"] = budget
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=cfg.grad_chunk, rng_mode=0)
    assert (cfg.q_hidden, cfg.se_hidden, cfg.batch_size, cfg.test_episodes, cfg.max_steps, cfg.grad_chunk) == (57, 83, 199, 10, 200, 17)
    S, A, pop = cfg.state_dim, cfg.num_actions, 8
    chains = 3 * pop
    rng = np.random.RandomState(17)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, cfg.q_hidden, 1, A, "tanh"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    eps = (rng.randn(pop, P_se) * 0.02).astype(np.float32)
    agent_init = rng.uniform(-0.4, 0.4, (chains, P_q)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(77, 3, int(worker[c]), c % 3) for c in range(chains)], np.uint64)
    outs = []
    for trace_cap in (0, 2):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        outs.append([t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.episode_len, il.final_returns, il.final_online)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    c = 4
    w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
    o = orc.ddqn_se_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]))
    assert float(outs[0][0][c]) == o["score"] and outs[0][1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(outs[0][2][c], o["episode_test_mean"], equal_nan=True) and np.array_equal(outs[0][4][c], o["final_test_returns"])


def _oracle_chains_parallel(fn, jobs):
    """Run the (GIL-releasing) oracle on several whole chains at once."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        return list(ex.map(lambda j: fn(*j), jobs))


@pytest.mark.timeout(900)
def test_dueling_full_size_properties(eng, orc):
    """BASELINE configs[2] at its real shapes -- Acrobot SE 9-128-{6,1,1}, DuelingDDQN 6-128-128-128 / 128-128-{1,3}
    (67 460 parameters), B = 128, 500-step episodes, 10 real-env test episodes -- on 96 chains (32 workers = one 8-GPU rank's
    share of pop 256): one full-length 500-step exploration episode + one full-length 500-step learning episode.
      * determinism, chain independence (permutation), antithetic symmetry -- bit for bit;
      * two whole chains against the oracle: score, counters, per-episode test means, final returns -- bit for bit."""
    from learning_environments_amd import configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(32), 2)
    cfgd["agents"]["duelingddqn"]["init_episodes"] = 1
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=0)
    assert (cfg.q_hidden, cfg.q_layers, cfg.feature_dim, cfg.batch_size, cfg.max_steps, cfg.se_hidden) == (128, 2, 128, 128, 500, 128)
    S, A, pop = cfg.state_dim, cfg.num_actions, 32
    chains = 3 * pop
    rng = np.random.RandomState(21)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.dueling_num_params(ocfg)
    assert P_q == 67460
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0                                   # done-net output bias: the SE never terminates -> full-length episodes
    eps = (rng.randn(pop, P_se) * 0.05).astype(np.float32)
    agent_init = rng.uniform(-0.08, 0.08, (chains, P_q)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(77, 3, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(theta_, eps_, worker_, sign_, init_, keys_, trace_cap=0):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap)
        il.run(dev(theta_), dev(eps_), dev(worker_), dev(sign_), dev(init_), rng_keys=dev(keys_.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return (il.score.cpu().numpy().copy(), il.stats.cpu().numpy().copy(), il.episode_test_mean.cpu().numpy().copy(),
                il.final_returns.cpu().numpy().copy())

    base = run(theta, eps, worker, sign, agent_init, keys)
    assert (base[1][:, 1] == 1000).all() and (base[1][:, 2] == 500).all()      # 2 x 500 env steps, 500 learn steps per chain
    again = run(theta, eps, worker, sign, agent_init, keys)
    for a, b in zip(base, again):
        assert np.array_equal(a, b, equal_nan=True)
    perm = rng.permutation(chains)
    permuted = run(theta, eps, worker[perm], sign[perm], agent_init[perm], keys[perm])
    for a, b in zip(base, permuted):
        assert np.array_equal(a[perm], b, equal_nan=True)
    flipped = run(theta, -eps, worker, -sign, agent_init, keys)
    for a, b in zip(base, flipped):
        assert np.array_equal(a, b, equal_nan=True)
    # this shape takes the shape-specialised instantiation; a launch that asks for a step trace takes the generic one
    generic = run(theta, eps, worker, sign, agent_init, keys, trace_cap=2)
    for a, b in zip(base, generic):
        assert np.array_equal(a, b, equal_nan=True)
    picks = (4, 95)
    outs = _oracle_chains_parallel(
        lambda c: orc.ddqn_se_chain(ocfg, (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c])),
        [(c,) for c in picks])
    for c, o in zip(picks, outs):
        assert float(base[0][c]) == o["score"]
        assert base[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(base[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(base[3][c], o["final_test_returns"])


@pytest.mark.timeout(1200)
def test_td3_full_size_properties(eng, orc):
    """BASELINE configs[4] at its real shapes -- RewardEnv 17-128-1 (type 2) on the HalfCheetah stand-in, TD3 actor
    17-128-128-6 + twin critics 23-128-128-1 (59 016 parameters), B = 192, 1000-step episodes -- on 96 chains: one full-length
    1000-step random-action episode + one full-length 1000-step learning episode.  Same properties as the cfg-3 test."""
    from learning_environments_amd import configs
    cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(32), 2)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.max_steps, cfg.rn_hidden) == (128, 2, 192, 1000, 128)
    Pa, Pc = orc.td3_param_counts(ocfg)
    assert Pa + 2 * Pc == 59016
    P_rn = orc.rn_num_params(2, 17, 4, ocfg.rn_hidden, 1)
    pop = 32
    chains = 3 * pop
    rng = np.random.RandomState(31)
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(pop, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(78, 1, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(theta_, eps_, worker_, sign_, init_, keys_, trace_cap=0):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap)
        il.run(dev(theta_), dev(eps_), dev(worker_), dev(sign_), dev(init_), rng_keys=dev(keys_.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return (il.score.cpu().numpy().copy(), il.stats.cpu().numpy().copy(), il.episode_test_mean.cpu().numpy().copy(),
                il.final_returns.cpu().numpy().copy())

    base = run(theta, eps, worker, sign, agent_init, keys)
    assert (base[1][:, 1] == 2000).all() and (base[1][:, 2] == 1000).all()
    again = run(theta, eps, worker, sign, agent_init, keys)
    for a, b in zip(base, again):
        assert np.array_equal(a, b, equal_nan=True)
    perm = rng.permutation(chains)
    permuted = run(theta, eps, worker[perm], sign[perm], agent_init[perm], keys[perm])
    for a, b in zip(base, permuted):
        assert np.array_equal(a[perm], b, equal_nan=True)
    flipped = run(theta, -eps, worker, -sign, agent_init, keys)
    for a, b in zip(base, flipped):
        assert np.array_equal(a, b, equal_nan=True)
    # this shape takes the shape-specialised instantiation; a launch that asks for a step trace takes the generic one
    generic = run(theta, eps, worker, sign, agent_init, keys, trace_cap=2)
    for a, b in zip(base, generic):
        assert np.array_equal(a, b, equal_nan=True)
    picks = (7, 92)
    outs = _oracle_chains_parallel(
        lambda c: orc.td3_rn_chain(ocfg, (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c])),
        [(c,) for c in picks])
    for c, o in zip(picks, outs):
        assert float(base[0][c]) == o["score"]
        assert base[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(base[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(base[3][c], o["final_test_returns"])


@pytest.mark.parametrize("foreign_cus", [128, 200])
def test_team_launch_next_to_a_foreign_kernel_gives_up_cleanly(eng, orc, foreign_cus):
    """Team launches assume the device to themselves (members wait for each other).  With a foreign kernel holding 128 / 200 of
    the 256 CUs for seconds on another stream, the 144 workgroups of a 24-chain TD3 team launch cannot all be resident.  Either the
    teams still assemble one after the other as chains finish (clean result), or some cannot (a member waits 0.25 s, gives up, the
    whole launch drains: status -10, clean refusal) -- in both cases the launch returns within a second, nothing hangs, and
    engine.run_checked then delivers the bits of the one-workgroup launch (repeating a refused launch with team_size 1) while the
    foreign kernel is still there.  (With fewer free CUs than ONE team needs the refusal is as prompt -- every started member gives up
    after 0.25 s -- but the rest of the grid only drains when the foreign kernel frees its shader engines: the hardware dispatcher
    deals workgroups to the shader engines in turn and waits for room on the one whose turn it is; tools/diag/foreign_kernel.py.)"""
    import ctypes as C
    import time
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from tools import diag
    cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(8), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["HalfCheetah-v3"]["max_steps"] = 20
    _, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 24
    rng = np.random.RandomState(91)
    P_rn = 17 * 128 + 128 + 128 + 1
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(8, P_rn) * 0.1).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 8)
    keys = chain_keys(80, 3, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, 59016)).astype(np.float32)
    args = (dev(theta), dev(eps), dev(worker), dev(sign), dev(init))
    kw = dict(rng_keys=dev(keys.view(np.int64)))

    cfg.team_size = 1
    il = eng.Td3InnerLoop(cfg, chains, want_final_params=True)
    il.run(*args, **kw)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    ref = [t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    cfg.team_size = 0
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 6
    il = eng.Td3InnerLoop(cfg, chains, want_final_params=True)      # (allocations first: a hipMalloc would wait for the foreign kernel)
    # a side stream that really runs next to the current one (HIP maps streams onto a few hardware queues; two streams on the same
    # queue serialise, and then there is no foreign kernel to speak of)
    probe = torch.ones(1024, device="cuda")
    side = None
    for _ in range(16):
        cand = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(cand):
            diag.occupy_cus(8, 150 * 1024, 20_000_000, cand.cuda_stream)
        time.sleep(0.02)
        probe.add_(1.0)
        torch.cuda.current_stream().synchronize()
        concurrent = not cand.query()
        cand.synchronize()
        if concurrent:
            side = cand
            break
    assert side is not None, "no stream runs concurrently with the current one"
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                          # the foreign kernel: `foreign_cus` CUs for 3 s
        diag.occupy_cus(foreign_cus, 150 * 1024, 3 * 100_000_000, side.cuda_stream)
    time.sleep(0.05)                                       # let it take its CUs first
    t0 = time.time()
    il.run(*args, **kw)
    torch.cuda.current_stream().synchronize()
    dt = time.time() - t0
    st = il.status.cpu().tolist()
    # Three outcomes, all clean: the teams assembled next to the foreign kernel (status 0), some could not and the launch drained after
    # the 0.25 s give-up (status -10) -- both well within a second --, or the hardware dispatcher held the rest of the grid back until the
    # foreign kernel freed its shader engines (it deals workgroups to the engines in turn and waits for room on the one whose turn it
    # is: then the launch ends with the foreign kernel).  Never a hang, never a half-finished chain taken for good.
    assert dt < 3.0 + 1.0, dt
    assert set(st) <= {0, -10}, st
    if dt < 1.0:                                           # (the prompt outcomes really happened NEXT TO the foreign kernel)
        assert not side.query(), dt
    he = eng.HipNesEngine()
    he.run_checked(il, *args, **kw)                        # a refused launch is repeated with one workgroup per chain
    assert cfg.team_size == (1 if min(st) < 0 else 0)
    out = [t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]
    for a, b in zip(ref, out):
        assert np.array_equal(a, b, equal_nan=True)
    side.synchronize()


@pytest.mark.timeout(1800)
def test_td3_bench_launch_vs_oracle_and_one_workgroup(eng, orc):
    """The launch bench.py times for BASELINE configs[4], exactly as bench.secondary_configs builds it: 8 workers = 24 chains,
    5 x 1000 train steps with init_episodes 1 (4 000 learn steps per chain: the replay ring holds 5 000 rows, every minibatch index is
    drawn against a size past B, six-way exchange and 24 000 team barriers), automatic team size = SIX workgroups per chain.
      * two whole chains against the oracle: score, counters, per-episode test means, final returns, all 59 016 final parameters;
      * every chain of the team launch against the one-workgroup launch (team_size 1): all outputs, bit for bit."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(8), 5)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.max_steps, cfg.train_episodes, cfg.init_episodes) == (128, 2, 192, 1000, 5, 1)
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_rn = orc.rn_num_params(2, 17, 4, ocfg.rn_hidden, 1)
    pop = 8
    chains = 3 * pop
    rng = np.random.RandomState(41)
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(pop, P_rn) * 0.1).astype(np.float32)
    agent_init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(1234, 1, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(team_size):
        cfg.team_size = team_size
        il = eng.Td3InnerLoop(cfg, chains, want_final_params=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    cfg.team_size = 0
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 6
    team = run(0)
    assert (team[1][:, 1] == 5000).all() and (team[1][:, 2] == 4000).all() and (team[1][:, 3] == 6000).all()
    picks = (5, 22)
    import threading
    outs = {}
    th = threading.Thread(target=lambda: outs.update(o=_oracle_chains_parallel(
        lambda c: orc.td3_rn_chain(ocfg, (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c]),
                                   want_final_params=True),
        [(c,) for c in picks])))
    th.start()                                             # the oracle's two chains (minutes of CPU) run while the GPU does the second launch
    one = run(1)
    for a, b in zip(team, one):
        assert np.array_equal(a, b, equal_nan=True)
    th.join()
    for c, o in zip(picks, outs["o"]):
        assert float(team[0][c]) == o["score"]
        assert team[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(team[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(team[3][c], o["final_test_returns"])
        assert np.array_equal(team[4][c], o["final_params"])


@pytest.mark.timeout(1800)
def test_dueling_bench_launch_vs_oracle_and_one_workgroup(eng, orc):
    """The launch bench.py times for BASELINE configs[2]: 32 workers = 96 chains, init_episodes 10 as published, 12 x 500 train steps
    (two learning episodes = 1 000 learn steps; bench.py runs 20 episodes of the same), automatic team size = two workgroups per chain.
    Two whole chains against the oracle incl. all 67 460 final parameters; all chains against the one-workgroup launch."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(32), 12)
    assert cfgd["agents"]["duelingddqn"]["init_episodes"] == 10
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=0)
    S, A, pop = cfg.state_dim, cfg.num_actions, 32
    chains = 3 * pop
    rng = np.random.RandomState(43)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.dueling_num_params(ocfg)
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0                                   # done-net output bias: the SE never terminates -> full-length episodes
    eps = (rng.randn(pop, P_se) * 0.05).astype(np.float32)
    agent_init = rng.uniform(-0.08, 0.08, (chains, P_q)).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(1234, 2, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(team_size):
        cfg.team_size = team_size
        il = eng.InnerLoop(cfg, chains, want_final_online=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy().copy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_online)]

    cfg.team_size = 0
    assert _lib.lib().lenv_dueling_team_size(C.byref(cfg), chains) == 2
    team = run(0)
    assert (team[1][:, 1] == 6000).all() and (team[1][:, 2] == 1000).all()
    picks = (9, 94)
    import threading
    outs = {}
    th = threading.Thread(target=lambda: outs.update(o=_oracle_chains_parallel(
        lambda c: orc.ddqn_se_chain(ocfg, (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32), agent_init[c], rng_key=int(keys[c]),
                                    want_final_online=True),
        [(c,) for c in picks])))
    th.start()
    one = run(1)
    for a, b in zip(team, one):
        assert np.array_equal(a, b, equal_nan=True)
    th.join()
    for c, o in zip(picks, outs["o"]):
        assert float(team[0][c]) == o["score"]
        assert team[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(team[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(team[3][c], o["final_test_returns"])
        assert np.array_equal(team[4][c], o["final_online"])


@pytest.mark.parametrize("budget", [1, 40, 100, 150, 10 ** 6])
def test_inner_loop_step_budget_vs_oracle(eng, orc, budget):
    """The deterministic time-out (lenv_ddqn_cfg::step_budget, standing in for base_agent.py:30-47 time_is_up): training
    stops at the first episode start with elapsed env steps > budget, the reward list is padded with its minimum so far
    (-1e9 if empty), and the final test is cut / padded against the remaining budget -- bit-exact against the oracle."""
    from learning_environments_amd import configs
    cfgd = configs.fixed_work(configs.cartpole_syn_env_ddqn(2), 5)
    cfgd["envs"]["CartPole-v0"]["max_steps"] = 14
    cfgd["agents"]["ddqn"].update(test_episodes=4, batch_size=24, hidden_size=20, step_budget=budget)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=2, rng_mode=0)
    assert cfg.step_budget == budget and ocfg.step_budget == budget
    S, A = 4, 2
    rng = np.random.RandomState(budget % 97)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    P_q = orc.mlp_num_params(orc.mlp_desc(S, ocfg.q_hidden, 1, A, "tanh"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    theta[-1] = -10.0
    chains = 3
    init = rng.uniform(-0.4, 0.4, (chains, P_q)).astype(np.float32)
    keys = np.array([orc.chain_key(5, 0, 0, c) for c in range(chains)], np.uint64)
    il = eng.InnerLoop(cfg, chains)
    il.run(dev(theta), None, None, None, dev(init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    timed_out = 0
    for c in range(chains):
        o = orc.ddqn_se_chain(ocfg, theta, init[c], rng_key=int(keys[c]))
        assert float(il.score[c]) == o["score"], (budget, c)
        assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
        assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
        timed_out += o["episodes_run"] < 5
    if budget == 1:
        # time-out before the second episode: the list is padded with the first episode's test mean and the final test scores -1e9
        assert timed_out == chains and float(il.score[0]) == -1e9
        etm = il.episode_test_mean[0].cpu().numpy()
        assert np.all(etm[1:] == etm[0])
    if budget == 10 ** 6:
        assert timed_out == 0


@pytest.mark.parametrize("budget", [1, 60, 200, 10 ** 6])
def test_step_budget_other_kernels_vs_oracle(eng, orc, golden, budget):
    """The env-step time-out in the DuelingDDQN, TD3 and QL kernels: counters, padded per-episode lists, cut final test and
    score bit-exact against the oracle, for a budget that stops training at once, in the middle, inside the final test, never."""
    from learning_environments_amd import configs
    from learning_environments_amd.envs.gridworld import transition_tables
    rng = np.random.RandomState(5)
    key = orc.chain_key(6, 0, 0, 0)
    kt = dev(np.array([key], np.uint64).view(np.int64))
    # DuelingDDQN on an Acrobot SE
    g = golden("g8d_calc_score_acrobot_dueling")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["duelingddqn"].update(hidden_size=24, feature_dim=16, batch_size=16, test_episodes=3, init_episodes=1, step_budget=budget)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=4, max_steps=9)
    S, A = ocfg.state_dim, ocfg.num_actions
    theta = (rng.randn(sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))) * 0.15).astype(np.float32)
    init = rng.uniform(-0.15, 0.15, (1, orc.dueling_num_params(ocfg))).astype(np.float32)
    il = eng.InnerLoop(cfg, 1)
    il.run(dev(theta), None, None, None, dev(init), rng_keys=kt)
    torch.cuda.synchronize()
    o = orc.ddqn_se_chain(ocfg, theta, init[0], rng_key=key)
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert np.array_equal(il.episode_len[0].cpu().numpy(), o["episode_len"])
    assert np.array_equal(il.final_returns[0].cpu().numpy(), o["final_test_returns"]) and float(il.score[0]) == o["score"]
    if budget == 1:
        assert o["episodes_run"] == 1 and o["score"] == -1e9
    # TD3 on the stand-in RewardEnv
    g = golden("g8t_calc_score_cheetah_td3")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["td3"].update(hidden_size=24, hidden_layer=1, batch_size=16, train_episodes=4, init_episodes=1, test_episodes=3, step_budget=budget)
    cfgd["envs"]["HalfCheetah-v3"].update(max_steps=6, hidden_size=24)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert cfg.step_budget == budget
    Pa, Pc = orc.td3_param_counts(ocfg)
    theta = (rng.randn(orc.rn_num_params(ocfg.reward_env_type, 17, 4, 24, 1)) * 0.2).astype(np.float32)
    init = rng.uniform(-0.2, 0.2, (1, Pa + 2 * Pc)).astype(np.float32)
    il = eng.Td3InnerLoop(cfg, 1)
    il.run(dev(theta), None, None, None, dev(init), rng_keys=kt)
    torch.cuda.synchronize()
    o = orc.td3_rn_chain(ocfg, theta, init[0], rng_key=key)
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert np.array_equal(il.episode_len[0].cpu().numpy(), o["episode_len"])
    assert np.array_equal(il.final_returns[0].cpu().numpy(), o["final_test_returns"]) and float(il.score[0]) == o["score"]
    # QL on the Cliff RewardEnv
    cfgd = configs.cliff_reward_env_ql(num_workers=1)
    cfgd["agents"]["ql"].update(train_episodes=8, test_episodes=3, step_budget=budget)
    cfgd["envs"]["Cliff"]["solved_reward"] = 1e9
    tables = transition_tables("Cliff")
    from learning_environments_amd.config import ql_cfg_from_config
    cfg = ql_cfg_from_config(cfgd, tables)
    ocfg = orc.ql_cfg_from_config(cfgd, tables)
    assert cfg.step_budget == budget and ocfg.step_budget == budget
    theta = (rng.randn(48 * 32 + 32 + 32 + 1) * 0.3).astype(np.float32)
    il = eng.QlInnerLoop(cfg, 1, tables)
    il.run(dev(theta), None, None, None, rng_keys=kt)
    torch.cuda.synchronize()
    o = orc.ql_rn_chain(ocfg, theta, tables, rng_key=key)
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert np.array_equal(il.episode_len[0].cpu().numpy(), o["episode_len"])
    assert np.array_equal(il.final_returns[0].cpu().numpy(), o["final_test_returns"]) and float(il.score[0]) == o["score"]


def test_dueling_and_td3_early_out(eng, orc, golden):
    """BaseAgent.train's early-out (base_agent.py:141-148: mean of the last early_out_num real-env test means >= solved_reward,
    only once learning has started) in the big-net kernels: fewer episodes than train_episodes, same as the oracle."""
    # DuelingDDQN on an Acrobot SE: returns are >= -max_steps, so solved_reward = -1000 fires at the first learning episode
    g = golden("g8d_calc_score_acrobot_dueling")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["duelingddqn"].update(hidden_size=24, feature_dim=16, batch_size=16, test_episodes=2, init_episodes=2, early_out_num=2)
    ocfg, cfg = _inner_cfg(orc, cfgd, grad_chunk=0, rng_mode=0, train_episodes=6, max_steps=10, solved_reward=-1000.0)
    rng = np.random.RandomState(3)
    S, A = ocfg.state_dim, ocfg.num_actions
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, ocfg.se_hidden, 1, "leakyrelu"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    init = rng.uniform(-0.15, 0.15, (1, orc.dueling_num_params(ocfg))).astype(np.float32)
    key = orc.chain_key(4, 0, 0, 0)
    il = eng.InnerLoop(cfg, 1)
    il.run(dev(theta), None, None, None, dev(init), rng_keys=dev(np.array([key], np.uint64).view(np.int64)))
    torch.cuda.synchronize()
    o = orc.ddqn_se_chain(ocfg, theta, init[0], rng_key=key)
    assert o["episodes_run"] == 3                       # 2 init episodes + the first learning episode
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert float(il.score[0]) == o["score"]
    # TD3 on the stand-in RewardEnv
    g = golden("g8t_calc_score_cheetah_td3")
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["td3"].update(hidden_size=24, hidden_layer=1, batch_size=16, train_episodes=6, init_episodes=2, test_episodes=2, early_out_num=2)
    cfgd["envs"]["HalfCheetah-v3"].update(max_steps=5, hidden_size=24, solved_reward=-1e6)
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    Pa, Pc = orc.td3_param_counts(ocfg)
    theta = (rng.randn(orc.rn_num_params(ocfg.reward_env_type, 17, 4, 24, 1)) * 0.2).astype(np.float32)
    init = rng.uniform(-0.2, 0.2, (1, Pa + 2 * Pc)).astype(np.float32)
    il = eng.Td3InnerLoop(cfg, 1)
    il.run(dev(theta), None, None, None, dev(init), rng_keys=dev(np.array([key], np.uint64).view(np.int64)))
    torch.cuda.synchronize()
    o = orc.td3_rn_chain(ocfg, theta, init[0], rng_key=key)
    assert o["episodes_run"] == 3
    assert il.stats[0].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(il.episode_test_mean[0].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert float(il.score[0]) == o["score"]


def test_nes_draw_vs_oracle(eng, orc):
    """lenv_nes_draw (noise + fresh agents + chain keys of a generation, one launch) against its CPU twin: bit-exact, for a
    rank that owns workers [2, 5) of a population of 7 with 5 chains per worker."""
    from learning_environments_amd.agents.nes_common import chain_keys
    pop, P, cpw, w_lo, n_local = 7, 2247, 5, 2, 3
    bounds = np.linspace(0.1, 0.5, 401).astype(np.float32)
    eps, init, keys = eng.nes_draw(99, 4, pop, P, 0.0124, cpw * n_local, cpw, w_lo, dev(bounds))
    oeps, oinit, okeys = orc.nes_draw(99, 4, pop, P, 0.0124, cpw * n_local, cpw, w_lo, bounds)
    assert np.array_equal(eps.cpu().numpy(), oeps) and np.array_equal(init.cpu().numpy(), oinit)
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), okeys)
    workers = np.repeat(np.arange(w_lo, w_lo + n_local), cpw)
    assert np.array_equal(okeys, chain_keys(99, 4, workers, np.tile(np.arange(cpw), n_local)))
    assert np.all(np.abs(oinit) <= bounds[None, :])
    # another generation / seed gives other numbers; QL-style call without agents
    e2, i2, k2 = eng.nes_draw(99, 5, pop, P, 0.0124, cpw * n_local, cpw, w_lo, None)
    assert i2 is None and not np.array_equal(e2.cpu().numpy(), oeps)


def test_nes_worker_best_multi(eng, orc, golden):
    """num_grad_evals = 3: device calc_best_score == the reference's (fixture G6M) for 'mean' / 'minmax', mirrored or not."""
    g = golden("g6m_worker_best_multi")
    add, sub = g["score_add"], g["score_sub"]
    pop = add.shape[0]
    orig = np.arange(pop, dtype=np.float64)
    cs = np.concatenate([orig[:, None], add, sub], axis=1)
    for gt in ("mean", "minmax"):
        for m in (True, False):
            res = eng.nes_worker_best(dev(cs.reshape(-1)), pop, m, 3, gt).cpu().numpy()
            assert np.array_equal(res[:, 0], g["best_%s_%d" % (gt, int(m))]), (gt, m)
            assert np.array_equal(res[:, 1], orig)
            assert np.array_equal(res[:, 2].astype(np.float32), g["sign_%s_%d" % (gt, int(m))]), (gt, m)
    with pytest.raises(NotImplementedError):
        eng.nes_worker_best(dev(cs.reshape(-1)), pop, True, 3, "median")


def test_ql_full_size_population_vs_oracle_and_properties(eng, orc):
    """BASELINE configs[3] at full size (Cliff RewardEnv + QL, pop 128 = 384 chains, 100 episodes): EVERY chain against the
    oracle (integer path: fp64 Q-tables bit-exact), plus determinism and chain-order independence."""
    from learning_environments_amd import configs
    from learning_environments_amd.config import ql_cfg_from_config
    from learning_environments_amd.envs.gridworld import transition_tables
    cfgd = configs.cliff_reward_env_ql(128)
    tables = transition_tables("Cliff")
    cfg = ql_cfg_from_config(cfgd, tables)
    ocfg = orc.ql_cfg_from_config(cfgd, tables)
    N, pop = tables["n_states"], 128
    chains = 3 * pop
    P = N * cfg.rn_hidden + 2 * cfg.rn_hidden + 1
    rng = np.random.RandomState(17)
    theta = (rng.randn(P) * 0.3).astype(np.float32)
    eps = (rng.randn(pop, P) * 0.1).astype(np.float32)
    worker = np.repeat(np.arange(pop), 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), pop)
    keys = np.array([orc.chain_key(9, 4, int(worker[c]), c % 3) for c in range(chains)], np.uint64)

    def run(worker_, sign_, keys_):
        il = eng.QlInnerLoop(cfg, chains, tables)
        il.run(dev(theta), dev(eps), dev(worker_), dev(sign_), rng_keys=dev(keys_.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return il.score.cpu().numpy().copy(), il.q_table.cpu().numpy().copy(), il.stats.cpu().numpy().copy()

    base = run(worker, sign, keys)
    again = run(worker, sign, keys)
    perm = rng.permutation(chains)
    permuted = run(worker[perm], sign[perm], keys[perm])
    for a, b, c in zip(base, again, permuted):
        assert np.array_equal(a, b) and np.array_equal(a[perm], c)
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ql_rn_chain(ocfg, w, tables, rng_key=int(keys[c]))
        assert np.array_equal(base[1][c].reshape(N, 4), o["q_table"]), c
        assert float(base[0][c]) == o["score"]
        assert base[2][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


def _wavechain_pair(run):
    """(production launch = wave-chain kernel, launch with a step trace = GEMM-queue kernel) of the same inputs."""
    return run(0), run(2)


def test_wavechain_dueling_kernel_equals_gemm_queue_kernel(eng):
    """dueling_wavechain.hip (BASELINE configs[2]'s shape, production launches) against dueling_se_inner_kernel on the same inputs:
    scores, counters, per-episode test means AND all 67 460 online parameters after 80 learn steps, bit for bit.  (The full-size
    property test above adds the oracle.)"""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(2), 3)
    cfgd["agents"]["duelingddqn"]["init_episodes"] = 1
    cfgd["envs"]["Acrobot-v1"]["max_steps"] = 40
    cfg = ddqn_cfg_from_config(cfgd)
    chains = 6
    rng = np.random.RandomState(5)
    P_se = 3 * (9 * 128 + 128) + (6 + 1 + 1) * 128 + 8
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0
    eps = (rng.randn(2, P_se) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    keys = chain_keys(77, 3, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, 67460)).astype(np.float32)

    def run(trace_cap):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_online)]

    a, b = _wavechain_pair(run)
    assert a[1][:, 2].min() == 80
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    assert not np.array_equal(a[4], init)


def test_wavechain_plain_dqn_kernel_equals_gemm_queue_kernel_and_oracle(eng, orc):
    """The wave-chain kernel's plain-DQN shape -- default_config_acrobot.yaml's ddqn section: Critic_DQN 6-128-128-3 relu, B = 128, on the
    Acrobot SE (dueling_wavechain.hip, kWcShapes[2]: layers 1 and 2, then the output layer where the dueling net has its advantage head) --
    in production launches (teams of two workgroups per chain, and one) against (i) the GEMM-queue kernel's plain-DQN mode on the same inputs (a launch that asks for a step trace) and
    (ii) the oracle's DDQN on two whole chains: scores, counters, per-episode test means, final returns AND all 17 795 online parameters
    after 80 learn steps, bit for bit."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.acrobot_syn_env_ddqn(2), 3)
    cfgd["envs"]["Acrobot-v1"]["max_steps"] = 40
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=0)
    assert (cfg.agent_kind, cfg.q_hidden, cfg.q_layers, cfg.batch_size, cfg.init_episodes, cfg.test_episodes) == (0, 128, 2, 128, 1, 10)
    assert _lib.lib().lenv_dueling_team_size(C.byref(cfg), 6) == 2          # teams of two, as the dueling shape
    chains = 6
    P_q = 6 * 128 + 128 + 128 * 128 + 128 + 3 * 128 + 3
    rng = np.random.RandomState(6)
    P_se = 3 * (9 * 128 + 128) + (6 + 1 + 1) * 128 + 8
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0
    eps = (rng.randn(2, P_se) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    keys = chain_keys(78, 3, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, P_q)).astype(np.float32)

    def run(trace_cap):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        assert il.p_agent == P_q
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_online)]

    a, b = _wavechain_pair(run)                             # (teams of two | the GEMM-queue kernel)
    assert a[1][:, 2].min() == 80
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    assert not np.array_equal(a[4], init)
    cfg.team_size = 1                                       # one workgroup per chain
    for x, y in zip(run(0), b):
        assert np.array_equal(x, y, equal_nan=True)
    cfg.team_size = 0
    for c in (1, 5):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.ddqn_se_chain(ocfg, w, init[c], rng_key=int(keys[c]), want_final_online=True)
        assert float(a[0][c]) == o["score"]
        assert a[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(a[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(a[3][c], o["final_test_returns"])
        assert np.array_equal(a[4][c], o["final_online"])


@pytest.mark.parametrize("family", ["duelingddqn", "ddqn_mountaincar"])
def test_published_big_net_shapes_with_a_two_layer_layer_norm_se_take_the_generic_kernel(eng, orc, family):
    """the shape-specialised GEMM-queue instantiations (kDuelShapes[1] / [2]) hard-code a one-hidden-layer SE without LayerNorm.
    A production launch of the published Acrobot DuelingDDQN / MountainCar DDQN agent shape on an SE built with `hidden_layer: 2,
    use_layer_norm: True` fails the wave-chain predicate and must NOT match those instantiations either: its scores, counters, test means
    and final parameters equal the launch with a step trace (always the generic kernel) and the oracle's chain."""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.config import ddqn_cfg_from_config
    if family == "duelingddqn":
        cfgd = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(2), 2)
        env, agent = "Acrobot-v1", "duelingddqn"
    else:
        cfgd = configs.fixed_work(configs.mountaincar_syn_env_ddqn(2), 2)
        env, agent = "MountainCar-v0", "ddqn"
    cfgd["agents"][agent]["init_episodes"] = 1
    cfgd["envs"][env].update(max_steps=24, hidden_layer=2, use_layer_norm=True)
    cfg = ddqn_cfg_from_config(cfgd)
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=0)
    assert (cfg.se_layers, cfg.se_layer_norm) == (2, 1)
    S, A = cfg.state_dim, cfg.num_actions
    chains = 3
    rng = np.random.RandomState(9)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(S, A, cfg.se_hidden, 2, {0: "identity", 1: "relu", 2: "leakyrelu", 3: "tanh", 4: "prelu"}[cfg.se_act]))
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0                               # done_net's output bias: the SE never ends an episode
    eps = (rng.randn(1, P_se) * 0.05).astype(np.float32)
    worker = np.zeros(chains, np.int32)
    sign = np.array([0.0, 1.0, -1.0], np.float32)
    keys = chain_keys(79, 1, worker, np.arange(chains))

    def run(trace_cap, init=None):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        if init is None:
            init = rng.uniform(-0.08, 0.08, (chains, il.p_agent)).astype(np.float32)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return init, [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_online)]

    init, a = run(0)
    _, b = run(2, init)
    assert a[1][:, 2].min() >= 20
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    w = (np.float32(sign[1]) * eps[0] + theta).astype(np.float32)
    o = orc.ddqn_se_chain(ocfg, w, init[1], rng_key=int(keys[1]), want_final_online=True)
    assert float(a[0][1]) == o["score"]
    assert a[1][1].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(a[4][1], o["final_online"])


@pytest.mark.parametrize("chains", [5, 10])
def test_wavechain_dueling_team_agrees(eng, chains):
    """The DuelingDDQN wave-chain kernel with a chain on a team of two workgroups (blocks on four waves each, weight gradients dealt by
    layer, four agent-scope barriers per learn step) against the one-workgroup launch and the GEMM-queue kernel: same bits."""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.config import ddqn_cfg_from_config
    cfgd = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(2), 3)
    cfgd["agents"]["duelingddqn"]["init_episodes"] = 1
    cfgd["envs"]["Acrobot-v1"]["max_steps"] = 30
    cfg = ddqn_cfg_from_config(cfgd)
    rng = np.random.RandomState(50 + chains)
    P_se = 3 * (9 * 128 + 128) + (6 + 1 + 1) * 128 + 8
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    theta[-1] = -10.0
    eps = (rng.randn(4, P_se) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 4)[:chains].copy()
    keys = chain_keys(76, 4, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, 67460)).astype(np.float32)

    def run(trace_cap):
        il = eng.InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_online=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_online)]

    ref = run(2)                                            # GEMM-queue kernel
    assert ref[1][:, 2].min() == 60
    for G in (1, 2):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G


def test_wavechain_td3_kernel_equals_gemm_queue_kernel(eng, orc):
    """td3_wavechain.hip (BASELINE configs[4]'s shape) against td3_rn_inner_kernel: scores, counters, test means and all 59 016
    parameters (actor | critic_1 | critic_2) after 120 learn steps, bit for bit."""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(2), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["HalfCheetah-v3"]["max_steps"] = 60
    _, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 6
    rng = np.random.RandomState(7)
    P_rn = 17 * 128 + 128 + 128 + 1
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(2, P_rn) * 0.1).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)
    keys = chain_keys(78, 1, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, 59016)).astype(np.float32)

    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    a, b = _wavechain_pair(run)
    assert a[1][:, 2].min() == 120
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    assert not np.array_equal(a[4], init)

def test_wavechain_td3_pendulum_shape_every_team_size(eng, orc):
    """The TD3 wave-chain kernel's second shape -- default_config_pendulum_reward_env.yaml: Pendulum-v0, actor 3-128-128-1, twin critics
    4-128-128-1, leakyrelu, batch 192, TEN test episodes per test phase, a reward net with TWO hidden layers (PReLU, potential shaping) --
    in production launches with G = 1, 2, 3, 6 workgroups per chain against (i) the GEMM-queue kernel on the same inputs and (ii) the
    oracle on two whole chains: scores, counters, per-episode test means, the ten final returns and all 50 947 parameters, bit for bit."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.pendulum_reward_env_td3(2), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["Pendulum-v0"]["max_steps"] = 30
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert (cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.layers, cfg.batch_size, cfg.test_episodes, cfg.rn_layers, cfg.policy_delay) == (3, 1, 128, 2, 192, 10, 2, 1)
    chains = 5
    Pa, Pc = orc.td3_param_counts(ocfg)
    assert Pa + 2 * Pc == (3 * 128 + 128 + 128 * 128 + 128 + 128 + 1) + 2 * (4 * 128 + 128 + 128 * 128 + 128 + 128 + 1)
    P_rn = orc.rn_num_params(2, 3, 0, 128, 2)
    assert P_rn == 3 * 128 + 128 + 128 * 128 + 128 + 128 + 1
    rng = np.random.RandomState(17)
    theta = (rng.randn(P_rn) * 0.1).astype(np.float32)
    eps = (rng.randn(2, P_rn) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)[:chains].copy()
    keys = chain_keys(81, 2, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)

    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    ref = run(2)                                            # GEMM-queue kernel (a launch with a step trace)
    assert ref[1][:, 2].min() == 60 and ref[3].shape == (chains, 10)
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 6
    for G in (1, 2, 3, 6):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G
    assert not np.array_equal(ref[4], init)
    for c in (0, 4):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, init[c], rng_key=int(keys[c]), want_final_params=True)
        assert float(ref[0][c]) == o["score"]
        assert ref[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(ref[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(ref[3][c], o["final_test_returns"])
        assert np.array_equal(ref[4][c], o["final_params"])


def test_wavechain_td3_cmc_shape_every_team_size(eng, orc):
    """The TD3 wave-chain kernel's third shape -- default_config_cmc_reward_env.yaml: MountainCarContinuous-v0 (an env that TERMINATES: the
    episode ends at the flag), actor 2-128-128-1, twin critics 3-128-128-1, leakyrelu, batch 192, one test episode, tanh reward net with one
    hidden layer, same_action_num 2 (every chosen action applied twice or until done, the shaped rewards summed) -- in production launches
    with G = 1, 2, 3, 6 workgroups per chain against the GEMM-queue kernel and, on two whole chains, the oracle: all outputs + the 51 331
    final parameters, bit for bit."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.cmc_reward_env_td3(2), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["MountainCarContinuous-v0"]["max_steps"] = 41       # odd: the last action of a full-length episode is applied once
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    assert (cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.layers, cfg.batch_size, cfg.test_episodes, cfg.rn_layers, cfg.policy_delay,
            cfg.same_action_num) == (2, 1, 128, 2, 192, 1, 1, 1, 2)
    chains = 5
    Pa, Pc = orc.td3_param_counts(ocfg)
    assert Pa + 2 * Pc == 51331
    P_rn = orc.rn_num_params(2, 2, 0, 128, 1)
    rng = np.random.RandomState(18)
    theta = (rng.randn(P_rn) * 0.1).astype(np.float32)
    eps = (rng.randn(2, P_rn) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)[:chains].copy()
    keys = chain_keys(82, 2, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)

    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params, il.episode_len)]

    ref = run(2)                                            # GEMM-queue kernel (a launch with a step trace)
    assert ref[1][:, 2].min() >= 20 and ref[5].max() == 42  # 21 agent steps; episode_length += same_action_num per agent step (base_agent.py:122)
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 6
    for G in (1, 2, 3, 6):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G
    for c in (0, 4):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, init[c], rng_key=int(keys[c]), want_final_params=True)
        assert float(ref[0][c]) == o["score"]
        assert ref[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(ref[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(ref[3][c], o["final_test_returns"])
        assert np.array_equal(ref[4][c], o["final_params"])


@pytest.mark.parametrize("which", ["cmc", "pendulum", "halfcheetah"])
def test_wavechain_td3_virtual_env_shapes_every_team_size(eng, orc, which):
    """The TD3 wave-chain kernel's VirtualEnv shapes (batch 256 = eight sample blocks, policy_delay 2: actor step and soft updates every
    second learn step; the training env is the synthetic env itself -- three nets on cat(action, state), the learned done output ends an
    episode -- and the test episodes run on the real env):
      cmc          default_config_cmc.yaml: actor 2-128-128-1 / critics 3-128-128-1 relu, SE nets 3-96-96-x leakyrelu, same_action_num 2, one test episode
      pendulum     default_config_pendulum.yaml's td3 section: actor 3-128-128-1, SE nets 4-32-32-x leakyrelu, ten lock-step test episodes
      halfcheetah  default_config_halfcheetah.yaml's td3 section: actor 17-128-128-6 / critics 23-128-128-1, SE nets 23-128-128-128-x relu, ten test episodes
    in production launches with G = 1, 2, 4, 8 workgroups per chain against the GEMM-queue kernel and, on two whole chains, the oracle: all
    outputs + all final parameters, bit for bit."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    make = {"cmc": configs.cmc_syn_env_td3, "pendulum": configs.pendulum_syn_env_td3, "halfcheetah": configs.halfcheetah_syn_env_td3}[which]
    cfgd = configs.fixed_work(make(2), 3)
    env_name = cfgd["env_name"]
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"][env_name]["max_steps"] = {"cmc": 41, "pendulum": 33, "halfcheetah": 25}[which]     # cmc, odd: range(0, 41, 2) = 21 agent steps per training episode
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    e = cfgd["envs"][env_name]
    S, A = cfg.state_dim, cfg.action_dim
    assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.policy_delay, cfg.virtual_env) == (128, 2, 256, 2, 1)
    assert (S, A, cfg.test_episodes, cfg.rn_hidden, cfg.rn_layers, cfg.same_action_num) == \
        {"cmc": (2, 1, 1, 96, 2, 2), "pendulum": (3, 1, 10, 32, 2, 1), "halfcheetah": (17, 6, 10, 128, 3, 1)}[which]
    chains = 5
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_se = orc.mlp_num_params(orc.mlp_desc(S + A, e["hidden_size"], e["hidden_layer"], S, e["activation_fn"])) + \
        2 * orc.mlp_num_params(orc.mlp_desc(S + A, e["hidden_size"], e["hidden_layer"], 1, e["activation_fn"]))
    rng = np.random.RandomState(28)
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    eps = (rng.randn(2, P_se) * 0.05).astype(np.float32)
    if which == "pendulum":
        theta[-1] = 0.484                                   # done_net's output bias: the learned done flag hovers around 0.5 and ends some episodes early
    if which == "cmc":
        theta[-1] = 0.4                                     # ... here for the last chain only (episodes of 2 / 40 / 18 env steps; both repeats of an action run whatever the flag says)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)[:chains].copy()
    keys = chain_keys(84, 2, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)

    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params, il.episode_len)]

    ref = run(2)                                            # GEMM-queue kernel (a launch with a step trace)
    if which == "pendulum":
        # episodes of different lengths: chain 0 runs 10 / 1 / 33 steps, chain 2 runs 4 / 2 / 10 (learning episodes cut short by the done net), chains 1 and 4 full length
        assert ref[5][0].tolist() == [10, 1, 33] and ref[5][2].tolist() == [4, 2, 10] and ref[5][4].tolist() == [33, 33, 33] and ref[1][:, 2].min() == 12
    else:
        assert ref[1][:, 2].min() >= 20                     # learn steps
    if which == "cmc":
        assert ref[5].max() == 42 and ref[5][4].tolist() == [2, 40, 18]      # episode_length += same_action_num per agent step (base_agent.py:122)
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 8
    for G in (1, 2, 4, 8):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G
    for c in (0, 2, 4) if which == "pendulum" else (0, 4):
        w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, init[c], rng_key=int(keys[c]), want_final_params=True)
        assert float(ref[0][c]) == o["score"]
        assert ref[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(ref[2][c], o["episode_test_mean"], equal_nan=True)
        assert np.array_equal(ref[3][c], o["final_test_returns"])
        assert np.array_equal(ref[4][c], o["final_params"])


def test_wavechain_td3_cmc_virtual_env_shard_launch(eng, orc):
    """default_config_cmc.yaml as one 8-GPU shard of its population (16 workers = 48 chains): the automatic launch puts every chain on a team
    of FOUR workgroups (192 of the 256 CUs).  Longer episodes than the team-size test (3 x 150 env steps = 3 x 75 agent steps, 150 learn steps, 75 delayed policy
    updates): every chain equal to the one-workgroup launch, one whole chain equal to the oracle incl.
    all final parameters."""
    import ctypes as C
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.cmc_syn_env_td3(16), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["MountainCarContinuous-v0"]["max_steps"] = 150
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 48
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_se = orc.mlp_num_params(orc.mlp_desc(3, 96, 2, 2, "leakyrelu")) + 2 * orc.mlp_num_params(orc.mlp_desc(3, 96, 2, 1, "leakyrelu"))
    rng = np.random.RandomState(29)
    theta = (rng.randn(P_se) * 0.1).astype(np.float32)
    eps = (rng.randn(16, P_se) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 16)
    keys = chain_keys(85, 3, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, Pa + 2 * Pc)).astype(np.float32)
    assert _lib.lib().lenv_td3_rn_team_size(C.byref(cfg), chains) == 4

    def run(team):
        cfg.team_size = team
        il = eng.Td3InnerLoop(cfg, chains, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params, il.episode_len)]

    auto, one = run(0), run(1)
    for x, y in zip(auto, one):
        assert np.array_equal(x, y, equal_nan=True)
    assert auto[1][:, 2].min() >= 100
    c = 7
    w = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)
    o = orc.td3_rn_chain(ocfg, w, init[c], rng_key=int(keys[c]), want_final_params=True)
    assert float(auto[0][c]) == o["score"]
    assert auto[1][c].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(auto[2][c], o["episode_test_mean"], equal_nan=True)
    assert np.array_equal(auto[4][c], o["final_params"])


def test_wavechain_td3_cmc_episodes_that_end_at_the_flag(eng, orc):
    """MountainCarContinuous-v0 terminates: with an actor that pushes in the direction of the velocity (hand-built weights: tanh(k v)) the
    car reaches the flag long before max_steps.  No learn step touches the actor here (init_episodes = train_episodes: random-action
    training episodes), so every test episode of the wave-chain kernel's one-row test loop ends at the env's own done flag -- in the middle
    of an action's two repeats or not --, with the +100 of the last step in its return: lengths, returns and scores equal to the GEMM-queue
    kernel's and the oracle's, bit for bit, for one workgroup per chain and for teams."""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.cmc_reward_env_td3(2), 2)
    cfgd["agents"]["td3"]["init_episodes"] = 2
    cfgd["envs"]["MountainCarContinuous-v0"]["max_steps"] = 400
    ocfg, cfg = _td3_cfgs(orc, cfgd, 0)
    chains = 4
    Pa, Pc = orc.td3_param_counts(ocfg)
    P_rn = orc.rn_num_params(2, 2, 0, 128, 1)
    rng = np.random.RandomState(19)
    theta = (rng.randn(P_rn) * 0.1).astype(np.float32)
    eps = (rng.randn(2, P_rn) * 0.05).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 2)[:chains].copy()
    keys = chain_keys(83, 1, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.01, 0.01, (chains, Pa + 2 * Pc)).astype(np.float32)
    # actor 2-128-128-1 (leakyrelu): h1_0 = lrelu(c v), h1_1 = lrelu(-c v); h2_0 = h1_0, h2_1 = h1_1; out = K (h2_0 - h2_1) -> tanh(~K c v)
    H = 128
    a = np.zeros(Pa, np.float32)
    W0, b0 = a[:2 * H].reshape(H, 2), a[2 * H:3 * H]
    W1 = a[3 * H:3 * H + H * H].reshape(H, H)
    Wo = a[3 * H + H * H + H:3 * H + H * H + H + H]
    W0[0, 1], W0[1, 1] = 40.0, -40.0
    W1[0, 0], W1[1, 1] = 1.0, 1.0
    Wo[0], Wo[1] = 30.0, -30.0
    init[:, :Pa] = a[None, :] + 0.0 * b0.sum()
    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    ref = run(2)                                            # GEMM-queue kernel
    assert (ref[1][:, 2] == 0).all()                        # no learn step
    assert ref[0].min() > 50.0                              # every final test episode reached the flag (+100 minus the action costs)
    assert ref[1][:, 3].max() < 3 * 400                     # ... well before the time limit (three test episodes per chain)
    for G in (1, 3, 6):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G
    o = orc.td3_rn_chain(ocfg, (np.float32(sign[1]) * eps[worker[1]] + theta).astype(np.float32), init[1], rng_key=int(keys[1]))
    assert float(ref[0][1]) == o["score"] and ref[1][1].tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
    assert np.array_equal(ref[2][1], o["episode_test_mean"], equal_nan=True)


@pytest.mark.parametrize("chains", [5, 11])
def test_wavechain_td3_team_sizes_agree(eng, orc, chains):
    """A chain run by a team of G = 2, 3, 6 workgroups (sample blocks and gradient tiles dealt over the team, six agent-scope
    barriers per learn step) gives the same bits as the one-workgroup launch (G = 1) and as the GEMM-queue kernel: scores, counters,
    test means and all 59 016 parameters."""
    from learning_environments_amd import configs
    from learning_environments_amd.agents.nes_common import chain_keys
    cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(2), 3)
    cfgd["agents"]["td3"]["init_episodes"] = 1
    cfgd["envs"]["HalfCheetah-v3"]["max_steps"] = 30
    _, cfg = _td3_cfgs(orc, cfgd, 0)
    rng = np.random.RandomState(70 + chains)
    P_rn = 17 * 128 + 128 + 128 + 1
    theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
    eps = (rng.randn(4, P_rn) * 0.1).astype(np.float32)
    worker = (np.arange(chains) // 3).astype(np.int32)
    sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 4)[:chains].copy()
    keys = chain_keys(79, 2, worker, np.arange(chains) % 3)
    init = rng.uniform(-0.08, 0.08, (chains, 59016)).astype(np.float32)

    def run(trace_cap):
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        return [t.cpu().numpy() for t in (il.score, il.stats, il.episode_test_mean, il.final_returns, il.final_params)]

    ref = run(2)                                            # GEMM-queue kernel
    assert ref[1][:, 2].min() == 60
    for G in (1, 2, 3, 6):
        cfg.team_size = G
        out = run(0)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y, equal_nan=True), G

