"""The DIRECT instantiations of the TD3 GEMM-queue kernel (td3_rn_inner_loop.hip: one-hidden-layer agent nets of at most 64 units, batch <= 256,
no LayerNorm / ICM -- default_config_cmc_syn_env_opt.yaml's TD3 and the narrow draws of td3_vary): TD3.learn (agents/TD3.py:63-110) with a thread
per minibatch sample instead of the product queue.  Every shape class of the routines (hidden sizes that are / are not multiples of eight: ds_read_b128
blocks against clamped reads; batches below / at 256; the three continuous envs = three (S, A) instantiations; every activation) against the oracle
chain AND against the queued products of the same kernel (`LENV_VARIANT_NO_DIRECT`), final parameters included: bit for bit."""
import json

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ENVS = {"HalfCheetah-v3": ("g8t_calc_score_cheetah_td3", 17, 6), "Pendulum-v0": ("g8pr_calc_score_pendulum_td3_reward_env", 3, 1),
        "MountainCarContinuous-v0": ("g8cr_calc_score_cmc_td3_reward_env", 2, 1)}


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


def _cfgs(orc, cfgd, variant):
    from learning_environments_amd import _lib
    o = orc.td3_cfg_from_config(cfgd, rng_mode=0)
    c = _lib.Td3Cfg()
    for f, _ in _lib.Td3Cfg._fields_:
        setattr(c, f, getattr(o, f, 0))
    c.kernel_variant = variant
    return o, c


@pytest.mark.parametrize("env_name,hidden,batch,act,delay,virtual",
                         [("MountainCarContinuous-v0", 64, 256, "leakyrelu", 2, True),     # the cmc_syn_env_opt shape
                          ("MountainCarContinuous-v0", 13, 37, "relu", 1, False), ("Pendulum-v0", 8, 16, "leakyrelu", 2, False),
                          ("Pendulum-v0", 20, 255, "tanh", 1, True), ("Pendulum-v0", 64, 64, "relu", 3, True),
                          ("HalfCheetah-v3", 24, 32, "leakyrelu", 2, False), ("HalfCheetah-v3", 61, 48, "tanh", 1, False),
                          ("HalfCheetah-v3", 56, 128, "relu", 1, True), ("HalfCheetah-v3", 3, 5, "relu", 1, False)])
def test_direct_learn_step_vs_oracle_and_queued_products(eng, orc, golden, env_name, hidden, batch, act, delay, virtual):
    from learning_environments_amd import _lib
    fx, S, A = ENVS[env_name]
    g = golden(fx)
    cfgd = json.loads(str(g["config_json"]))
    cfgd["agents"]["gtn"]["synthetic_env_type"] = 0 if virtual else 1
    steps = max(8, (batch + 1) // 2 + 3)                      # the second episode learns on a full minibatch's worth of distinct rows at least once
    cfgd["agents"]["td3"].update(hidden_size=hidden, hidden_layer=1, batch_size=batch, activation_fn=act, policy_delay=delay, train_episodes=3,
                                 init_episodes=1, test_episodes=1, early_out_num=50, same_action_num=1)
    env_h, env_l = (128, 3) if (hidden, batch) in ((64, 256), (56, 128)) else (24, 1)        # 128-wide SE nets: the env step's LDS-row path
    cfgd["envs"][env_name].update(max_steps=min(steps, 40), hidden_size=env_h, hidden_layer=env_l, reward_env_type=0 if virtual else 2, solved_reward=1e9)
    ocfg, _ = _cfgs(orc, cfgd, 0)
    Pa, Pc = orc.td3_param_counts(ocfg)
    assert (Pa, Pc) == (S * hidden + hidden + A * hidden + A, (S + A) * hidden + 2 * hidden + 1)
    if virtual:
        P_rn = orc.mlp_num_params(orc.mlp_desc(S + A, env_h, env_l, S, cfgd["envs"][env_name]["activation_fn"])) + \
            2 * orc.mlp_num_params(orc.mlp_desc(S + A, env_h, env_l, 1, cfgd["envs"][env_name]["activation_fn"]))
    else:
        P_rn = max(1, orc.rn_num_params(2, S, ocfg.info_dim, env_h, env_l))
    rng = np.random.RandomState(hidden * 7 + batch)
    chains = 2
    scale = 0.2 * (24.0 / env_h) ** 0.5                       # (wide random SE nets would blow the states up to inf / nan within a few steps)
    theta = (rng.randn(P_rn) * scale).astype(np.float32)
    eps = (rng.randn(1, P_rn) * 0.5 * scale).astype(np.float32)
    agent_init = rng.uniform(-0.3, 0.3, (chains, Pa + 2 * Pc)).astype(np.float32)
    worker, sign = np.zeros(chains, np.int32), np.array([1.0, -1.0], np.float32)
    keys = np.array([orc.chain_key(29, 2, 0, c) for c in range(chains)], np.uint64)
    cap = 3 * 40
    runs = {}
    for variant in (0, _lib.VARIANT_NO_DIRECT):
        _, cfg = _cfgs(orc, cfgd, variant)
        il = eng.Td3InnerLoop(cfg, chains, trace_cap=cap, want_final_params=True)
        il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(agent_init), rng_keys=dev(keys.view(np.int64)))
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        runs[variant] = dict(final=il.final_params.cpu().numpy().copy(), action=il.trace["action"].cpu().numpy().copy(),
                             reward=il.trace["reward"].cpu().numpy().copy(), score=il.score.cpu().numpy().copy(), stats=il.stats.cpu().tolist(),
                             etm=il.episode_test_mean.cpu().numpy().copy())
    d, q = runs[0], runs[_lib.VARIANT_NO_DIRECT]
    assert d["stats"] == q["stats"] and d["stats"][0][2] > 0, d["stats"]
    for k in ("final", "action", "reward", "score", "etm"):
        assert np.array_equal(d[k], q[k], equal_nan=True), k             # thread-per-sample chains == the queued products' chains
    for c in range(chains):
        w = (np.float32(sign[c]) * eps[0] + theta).astype(np.float32)
        o = orc.td3_rn_chain(ocfg, w, agent_init[c], rng_key=int(keys[c]), trace_cap=cap, want_final_params=True)
        n = o["trace"]["reward"].size
        assert o["rc"] == 0 and o["learn_steps"] > 0
        assert np.array_equal(d["action"][c, :n], o["trace"]["action"]), c
        assert np.array_equal(d["reward"][c, :n], o["trace"]["reward"]), c
        assert float(d["score"][c]) == o["score"]
        assert d["stats"][c] == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
        assert np.array_equal(d["final"][c], o["final_params"]) and np.isfinite(d["final"][c]).all(), c


def test_direct_is_not_taken_where_it_does_not_apply(eng, orc, golden):
    """two hidden layers / a 65-wide layer: the queued products whatever the variant word says -- NO_DIRECT and the default agree bit for bit there
    too (such launches equal the oracle in the tests of test_gpu_parity.py)."""
    from learning_environments_amd import _lib
    g = golden("g8t_calc_score_cheetah_td3")
    for hidden, layers in ((65, 1), (24, 2)):
        cfgd = json.loads(str(g["config_json"]))
        cfgd["agents"]["td3"].update(hidden_size=hidden, hidden_layer=layers, batch_size=16, train_episodes=2, init_episodes=1, test_episodes=1)
        cfgd["envs"]["HalfCheetah-v3"].update(max_steps=10, hidden_size=24)
        ocfg, _ = _cfgs(orc, cfgd, 0)
        P_rn = max(1, orc.rn_num_params(ocfg.reward_env_type, 17, ocfg.info_dim, 24, 1))
        rng = np.random.RandomState(5)
        theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
        keys = np.array([orc.chain_key(3, 1, 0, 0)], np.uint64)
        outs, init = [], None
        for variant in (0, _lib.VARIANT_NO_DIRECT):
            _, cfg = _cfgs(orc, cfgd, variant)
            il = eng.Td3InnerLoop(cfg, 1, want_final_params=True)
            if init is None:
                init = rng.uniform(-0.3, 0.3, (1, il.p_agent)).astype(np.float32)
            il.run(dev(theta), None, None, None, dev(init), rng_keys=dev(keys.view(np.int64)))
            torch.cuda.synchronize()
            assert il.status.cpu().tolist() == [0]
            outs.append(il.final_params.cpu().numpy().copy())
        assert np.array_equal(outs[0], outs[1], equal_nan=True)


def test_direct_replays_the_reference_run_of_the_syn_env_opt_shape(eng, orc, golden):
    """Fixture G8CO: the reference's own TD3 run at default_config_cmc_syn_env_opt.yaml's REAL shapes (actor 2-64-1, critics 3-64-1 leakyrelu,
    batch 256, policy_delay 2, same_action_num 2, SE nets 3-128-128-128-x relu; 40 learn steps, 20 policy updates) replayed in tape mode by
    the DIRECT instantiation (thread-per-sample learn step + the LDS-row SE step): bit-equal to the oracle and to the queued products incl.
    all final parameters, within the fixture tolerances of the reference's own numbers."""
    from learning_environments_amd import _lib
    g = golden("g8co_calc_score_cmc_td3_syn_env_opt_fullshape")
    cfgd = json.loads(str(g["config_json"]))
    n = g["tr_reward"].size
    o_cfg = orc.td3_cfg_from_config(cfgd, rng_mode=1)
    assert (o_cfg.hidden, o_cfg.layers, o_cfg.batch_size, o_cfg.policy_delay, o_cfg.rn_hidden, o_cfg.rn_layers, o_cfg.virtual_env) == (64, 1, 256, 2, 128, 3, 1)
    otapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                                g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=1, S=2)
    o = orc.td3_rn_chain(o_cfg, g["theta"], g["agent_init"], tapes=otapes, trace_cap=n + 4, want_final_params=True)
    assert o["rc"] == 0 and o["learn_steps"] == 40
    chains = 2
    rep = lambda a: dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    tapes = dict(rand_action=rep(g["tape_rand_action"]), act_noise=rep(g["tape_act_noise"]), test_noise=rep(g["tape_test_noise"]),
                 policy_noise=rep(g["tape_policy_noise"]), replay_idx=rep(g["tape_replay_idx"].reshape(-1)),
                 train_reset=rep(g["tape_train_reset"]), test_reset=rep(g["tape_test_reset"]))
    finals = {}
    for variant in (0, _lib.VARIANT_NO_WAVECHAIN, _lib.VARIANT_NO_WAVECHAIN | _lib.VARIANT_NO_DIRECT):
        c = _lib.Td3Cfg()
        for f, _ in _lib.Td3Cfg._fields_:
            setattr(c, f, getattr(o_cfg, f, 0))
        c.kernel_variant = variant
        il = eng.Td3InnerLoop(c, chains, trace_cap=n + 4, want_episode_stats=True, want_final_params=True)
        assert il.p_agent == g["agent_init"].size
        il.run(dev(g["theta"]), None, None, None, dev(np.tile(g["agent_init"], (chains, 1))), tapes=tapes)
        torch.cuda.synchronize()
        assert il.status.cpu().tolist() == [0] * chains
        for ch in range(chains):
            assert np.array_equal(il.trace["action"][ch, :n].cpu().numpy(), o["trace"]["action"])
            assert np.array_equal(il.trace["next_state"][ch, :n].cpu().numpy(), o["trace"]["next_state"])
            assert np.array_equal(il.trace["reward"][ch, :n].cpu().numpy(), o["trace"]["reward"])
            assert np.array_equal(il.episode_len[ch].cpu().numpy(), o["episode_len"])
            assert np.array_equal(il.final_params[ch].cpu().numpy(), o["final_params"])
            assert float(il.score[ch]) == o["score"]
            assert il.stats[ch].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]
            np.testing.assert_allclose(il.trace["action"][ch, :n].cpu().numpy(), g["tr_action"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(il.trace["next_state"][ch, :n].cpu().numpy(), g["tr_next_state"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(il.trace["reward"][ch, :n].cpu().numpy(), g["tr_reward"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(il.final_params[ch].cpu().numpy(), g["final_params"], rtol=0, atol=2e-6)
            assert abs(float(il.score[ch]) - float(g["score"])) <= 1e-4
        finals[variant] = il.final_params.cpu().numpy().copy()
    assert len({f.tobytes() for f in finals.values()}) == 1
