"""TD3_discrete_vary (agents/TD3_discrete_vary.py: TD3 on a discrete action space through a Gumbel-softmax actor, optionally with the
shared LayerNorm of models/model_utils.py:22-37).

CPU part: the oracle (oracle/lenv_oracle_td3d.inc) against runs of the REFERENCE (fixtures G4TD: learn calls on a fixed buffer;
G8TD*: whole GTN_Worker.calc_score runs on a CartPole / Acrobot VirtualEnv with every random draw recorded).
GPU part: the fused HIP kernel against the oracle, bit for bit, in tape and counter mode."""
import json

import numpy as np
import pytest

CHAIN_FIXTURES = ["g8td_calc_score_cartpole_td3_discrete", "g8tdl_calc_score_acrobot_td3_discrete_layer_norm",
                  "g8td3_calc_score_cartpole_td3_discrete_3_layers", "g8tdv_calc_score_cartpole_td3_discrete_vary",
                  "g8tdseln_calc_score_cartpole_td3_discrete_se_layernorm"]       # the ENV section's use_layer_norm (SE nets 6-20-20-x)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _learn_cfg(orc, g, pre):
    S, A, H, L, act, B, pd, nsteps, ln, hard, it0 = [int(x) for x in g[pre + "meta"]]
    gamma, lr, tau, pstd, pclip, ma, temp = [float(x) for x in g[pre + "hparams"]]
    cfg = orc.Td3dCfg(env_id={4: 0, 6: 1}[S], state_dim=S, action_dim=A, max_steps=5, se_hidden=8, se_layers=1, se_act=1, se_prelu=0.25,
                      hidden=H, layers=L, act=act, prelu=0.25, use_layer_norm=ln, gumbel_hard=hard, batch_size=B, rb_size=100,
                      train_episodes=1, test_episodes=1, init_episodes=0, early_out_num=1, policy_delay=pd, rng_mode=1, solved_reward=1e9,
                      gamma=gamma, lr=lr, tau=tau, action_std=0.1, policy_std=pstd, policy_std_clip=pclip, max_action=ma, gumbel_temp=temp,
                      adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, step_budget=0)
    return cfg, nsteps, it0


def test_temperature_schedule_is_numpy_linspace(orc):
    for temp in (1.0, 0.7, 2.5):
        cfg = orc.Td3dCfg(gumbel_temp=temp)
        steps = np.linspace(temp, temp / 20, 2000)              # TD3_discrete_vary.py:59
        got = np.array([orc.td3d_temperature(cfg, i) for i in range(2000)], np.float32)
        assert np.array_equal(got, steps.astype(np.float32))
        assert orc.td3d_temperature(cfg, 123456) == np.float32(steps[-1])


def test_learn_calls_vs_reference(orc, golden):
    """Actor_TD3_discrete forward and TD3_discrete_vary.learn: LayerNorm (one and two positions sharing one module), soft and hard
    Gumbel softmax, policy_delay 1 and 2, a temperature off the schedule's first entry."""
    g = golden("g4td_td3_discrete_learn")
    for vi in range(int(g["n_variants"])):
        pre = "v%d_" % vi
        cfg, nsteps, it0 = _learn_cfg(orc, g, pre)
        P, Pa, Pc = orc.td3d_num_params(cfg)
        assert P == g[pre + "params0"].size
        y = orc.td3d_actor_forward(cfg, g[pre + "params0"][:Pa], g[pre + "fwd_s"], g[pre + "fwd_gumbel"], 0.8)
        np.testing.assert_allclose(y, g[pre + "fwd_actor"], rtol=0, atol=5e-7)
        p, t = g[pre + "params0"].copy(), g[pre + "targets0"].copy()
        m, v, pows = np.zeros_like(p), np.zeros_like(p), [1.0] * 4
        for st in range(nsteps):
            p, t, m, v, pows = orc.td3d_learn(cfg, p, t, m, v, pows, it0 + st + 1, g[pre + "rows"][st], g[pre + "policy_noise"][st],
                                              g[pre + "gumbel_target"][st], g[pre + "gumbel_actor"][st])
            np.testing.assert_allclose(p, g[pre + "params"][st], rtol=0, atol=2e-6)      # measured 7e-7
            np.testing.assert_allclose(t, g[pre + "targets"][st], rtol=0, atol=5e-7)
        assert np.abs(p[:Pa] - g[pre + "params0"][:Pa]).max() > 1e-3                    # the actor did train


def chain_inputs(orc, g, rng_mode=1, **over):
    cfgd = json.loads(str(g["config_json"]))
    hp = json.loads(str(g["hp_json"]))
    cfg = orc.td3d_cfg_from_config(cfgd, rng_mode=rng_mode, hp=hp, **over)
    tapes = {k: g["tape_" + k] for k in orc.TD3D_TAPE_KEYS}
    return cfg, tapes


@pytest.mark.parametrize("name", CHAIN_FIXTURES)
def test_calc_score_vs_reference(orc, golden, name):
    g = golden(name)
    cfg, tapes = chain_inputs(orc, g)
    assert orc.td3d_num_params(cfg)[0] == g["agent_init"].size
    n = g["tr_reward"].size
    r = orc.td3d_chain(cfg, g["theta"], g["agent_init"], tapes=orc.make_td3d_tapes(cfg.action_dim, **tapes), trace_cap=n + 4)
    assert r["rc"] == 0
    assert r["train_steps"] == n and len(r["trace"]["reward"]) == n
    E = g["reward_list_train"].size                              # the reference's lists stop at the early-out
    assert r["episodes_run"] == E
    assert np.array_equal(r["episode_len"][:E], g["episode_length_train"])
    np.testing.assert_allclose(r["episode_test_mean"][:E], g["reward_list_train"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["final_test_returns"], g["reward_list_test"], rtol=0, atol=1e-9)
    assert abs(r["score"] - float(g["score"])) <= 1e-4           # north_star bar
    # the action vectors the replay buffer holds (Gumbel softmax + Gaussian noise), the env's view of them, the SE's answers
    np.testing.assert_allclose(r["trace"]["action"], g["rb_action"][:n], rtol=0, atol=2e-6)
    assert np.array_equal(r["trace"]["action"].argmax(1), g["tr_action"])
    np.testing.assert_allclose(r["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["trace"]["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["final_params"], g["final_params"], rtol=0, atol=2e-5)     # measured <= 2.1e-6


def test_counter_mode_runs_and_is_deterministic(orc, golden):
    g = golden(CHAIN_FIXTURES[1])
    cfg, _ = chain_inputs(orc, g, rng_mode=0)
    a = orc.td3d_chain(cfg, g["theta"], g["agent_init"], rng_key=77)
    b = orc.td3d_chain(cfg, g["theta"], g["agent_init"], rng_key=77)
    c = orc.td3d_chain(cfg, g["theta"], g["agent_init"], rng_key=78)
    assert a["rc"] == 0 and np.array_equal(a["final_params"], b["final_params"]) and a["score"] == b["score"]
    assert not np.array_equal(a["final_params"], c["final_params"])
    # the Gumbel draw: -log(-log(u)) is finite and has the Gumbel(0,1) mean (Euler-Mascheroni) / variance (pi^2 / 6)
    gs = np.array([orc.gumbel(5, 13, i) for i in range(20000)])
    assert np.isfinite(gs).all() and abs(gs.mean() - 0.5772) < 0.03 and abs(gs.var() - np.pi ** 2 / 6) < 0.08


# ------------------------------------------------------------------------------------------------------------------
# GPU part: the fused kernel (lenv_td3d_inner_loop) against the oracle, bit for bit
# ------------------------------------------------------------------------------------------------------------------
def _dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _hip_cfg(ocfg):
    from learning_environments_amd import _lib
    c = _lib.Td3dCfg()
    for f, _ in _lib.Td3dCfg._fields_:
        setattr(c, f, getattr(ocfg, f))
    return c


def _assert_chain_equals_oracle(il, c, o, n):
    assert np.array_equal(il.trace["action"][c, :n].cpu().numpy(), o["trace"]["action"][:n])
    assert np.array_equal(il.trace["next_state"][c, :n].cpu().numpy(), o["trace"]["next_state"][:n])
    assert np.array_equal(il.trace["reward"][c, :n].cpu().numpy(), o["trace"]["reward"][:n])
    assert np.array_equal(il.episode_test_mean[c].cpu().numpy(), o["episode_test_mean"], equal_nan=True)
    assert np.array_equal(il.episode_len[c].cpu().numpy(), o["episode_len"])
    assert np.array_equal(il.final_returns[c].cpu().numpy(), o["final_test_returns"])
    assert float(il.score[c]) == o["score"]
    assert il.stats[c].cpu().tolist() == [o["episodes_run"], o["train_steps"], o["learn_steps"], o["test_steps"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CHAIN_FIXTURES)
def test_hip_tape_mode_vs_reference_and_oracle(orc, golden, name):
    """The reference's own run replayed from its recorded draws: HIP == oracle bit for bit, and both within the fixture tolerances
    of the reference.  The *_vary fixture runs in a launch sized for the largest possible draw."""
    import torch
    from learning_environments_amd import engine
    engine.require_device()
    g = golden(name)
    ocfg, tapes = chain_inputs(orc, g)
    hp = json.loads(str(g["hp_json"]))
    n = g["tr_reward"].size
    o = orc.td3d_chain(ocfg, g["theta"], g["agent_init"], tapes=orc.make_td3d_tapes(ocfg.action_dim, **tapes), trace_cap=n + 4)
    assert o["rc"] == 0
    chains = 2
    cfg = _hip_cfg(ocfg)
    if hp:                                                  # launch sized for the maxima of the draw
        cfg.batch_size, cfg.hidden, cfg.layers = 3 * ocfg.batch_size, 3 * ocfg.hidden, min(3, ocfg.layers + 2)
    il = engine.Td3DiscreteInnerLoop(cfg, chains, trace_cap=n + 4, want_final_params=True, vary=bool(hp))
    if hp:
        il.set_hp([hp["lr"]] * chains, [hp["batch_size"]] * chains, [hp["hidden_size"]] * chains, [hp["hidden_layer"]] * chains)
        assert il.chain_num_params(hp["hidden_size"], hp["hidden_layer"]) == g["agent_init"].size
    else:
        assert il.p_agent == g["agent_init"].size
    init = np.zeros((chains, il.p_agent), np.float32)
    init[:, :g["agent_init"].size] = g["agent_init"]
    rep = lambda a: _dev(np.tile(np.ascontiguousarray(a)[None], (chains,) + (1,) * np.ndim(a)))
    dt = {k: rep(tapes[k]) for k in orc.TD3D_TAPE_KEYS}
    il.run(_dev(g["theta"]), None, None, None, _dev(init), tapes=dt)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    P = g["agent_init"].size
    for c in range(chains):
        _assert_chain_equals_oracle(il, c, o, n)
        assert np.array_equal(il.final_params[c, :P].cpu().numpy(), o["final_params"])
        np.testing.assert_allclose(il.trace["action"][c, :n].cpu().numpy(), g["rb_action"][:n], rtol=0, atol=2e-6)
        np.testing.assert_allclose(il.final_params[c, :P].cpu().numpy(), g["final_params"], rtol=0, atol=2e-5)
        assert abs(float(il.score[c]) - float(g["score"])) <= 1e-4        # north_star bar


@pytest.mark.gpu
@pytest.mark.parametrize("name,over", [
    (CHAIN_FIXTURES[0], dict()),
    (CHAIN_FIXTURES[0], dict(use_layer_norm=1, layers=3, hidden=33, batch_size=21, policy_delay=2, test_episodes=4)),
    (CHAIN_FIXTURES[1], dict()),
    (CHAIN_FIXTURES[1], dict(gumbel_hard=0, layers=1, act=1, batch_size=130, hidden=40, init_episodes=1)),
    (CHAIN_FIXTURES[2], dict(act=2, gumbel_hard=1, gumbel_temp=0.5, step_budget=60)),
])
def test_hip_counter_mode_vs_oracle(orc, golden, name, over):
    """Production RNG: every chain's draws come from its key (Gumbel = -log(-log(u)), Box-Muller Gaussians, replay indices,
    resets); three chains with different keys, each equal to the oracle chain with that key."""
    import torch
    from learning_environments_amd import engine
    engine.require_device()
    g = golden(name)
    ocfg, _ = chain_inputs(orc, g, rng_mode=0, **over)
    P = orc.td3d_num_params(ocfg)[0]
    cfg = _hip_cfg(ocfg)
    chains = 3
    keys = np.array([11, 2 ** 62 + 5, 123456789], np.uint64)
    il = engine.Td3DiscreteInnerLoop(cfg, chains, trace_cap=512, want_final_params=True)
    assert il.p_agent == P
    kt = _dev(keys.view(np.int64))
    init = il.draw_agent_init(kt)
    rng = np.random.RandomState(3)
    theta = g["theta"]
    eps = (rng.randn(2, theta.size) * 0.05).astype(np.float32)
    worker, sign = np.array([0, 1, 1], np.int32), np.array([1.0, -1.0, 0.0], np.float32)
    il.run(_dev(theta), _dev(eps), _dev(worker), _dev(sign), None, rng_keys=kt)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0] * chains
    init_h = init.cpu().numpy()
    if ocfg.use_layer_norm and ocfg.layers >= 2:            # LayerNorm weight 1 / bias 0 inside every net
        H, S = ocfg.hidden, ocfg.state_dim
        o0 = S * H + H + H * H + H
        assert np.array_equal(init_h[:, o0:o0 + H], np.ones((chains, H), np.float32)) and not init_h[:, o0 + H:o0 + 2 * H].any()
    for c in range(chains):
        se = (np.float32(sign[c]) * eps[worker[c]] + theta).astype(np.float32)     # sign in {1, -1, 0}: the kernel's fma is exact
        o = orc.td3d_chain(ocfg, se, init_h[c], rng_key=int(keys[c]), trace_cap=512)
        assert o["rc"] == 0
        n = min(o["train_steps"], 512)
        _assert_chain_equals_oracle(il, c, o, n)
        assert np.array_equal(il.final_params[c].cpu().numpy(), o["final_params"])
    assert len({float(s) for s in il.score.cpu()}) >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "layer_norm_vary"])
def test_gtn_master_td3_discrete_generation(variant, tmp_path, monkeypatch):
    """`agent_name: TD3_discrete_vary` through GTN_Master: one launch per generation, every fitness equal to the oracle chain with
    that chain's key, fresh agent and (vary_hp) hyper-parameter draw; then a full run()."""
    import torch
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_td3_discrete, fixed_work
    from oracle import oracle as orc
    over = dict(hidden_size=40, batch_size=24, test_episodes=3)
    if variant != "plain":
        over.update(use_layer_norm=True, vary_hp=True, gumbel_softmax_hard=False, policy_delay=2)
    cfg = fixed_work(cartpole_syn_env_td3_discrete(num_workers=2, max_iterations=1, **over), 3)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 12
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    m = GTN_Master(cfg, bohb_id=0, seed=5)
    assert isinstance(m.task, tasks.Td3DiscreteTask) and m.inner.vary == (variant != "plain")
    if variant != "plain":
        assert m.cfg.batch_size == 72 and m.cfg.hidden == 120 and m.cfg.layers == 3 and m.cfg.use_layer_norm == 1
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    inits = m.inner.agent_init.cpu().numpy()
    hps = m.task.last_hp
    for p in range(2):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            c = 3 * p + kind
            key = orc.chain_key(m.seed, 0, p, kind)
            ocfg = orc.td3d_cfg_from_config(cfg, hp=hps[c] if hps else None)
            P = orc.td3d_num_params(ocfg)[0]
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            r = orc.td3d_chain(ocfg, w, inits[c, :P], rng_key=key)
            assert r["rc"] == 0 and r["learn_steps"] > 0
            sc.append(r["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and np.isfinite(mean_score)


@pytest.mark.gpu
def test_shipped_shape_runs_and_matches_oracle(orc):
    """The td3_discrete_vary section as the syn-env YAMLs ship it (510-wide tanh nets, batch 122, hard Gumbel softmax, temperature
    2.31) on an Acrobot SE, shortened to a few episodes: two chains, bit-equal to the oracle."""
    import torch
    from learning_environments_amd import engine
    from learning_environments_amd.config import td3d_cfg_from_config
    from learning_environments_amd.configs import acrobot_syn_env_td3_discrete, fixed_work
    engine.require_device()
    cfgd = fixed_work(acrobot_syn_env_td3_discrete(num_workers=2, train_episodes=3, test_episodes=2), 3)
    cfgd["envs"]["Acrobot-v1"]["max_steps"] = 20
    cfg = td3d_cfg_from_config(cfgd)
    ocfg = orc.td3d_cfg_from_config(cfgd)
    for f, _ in type(cfg)._fields_:
        assert getattr(cfg, f) == getattr(ocfg, f), f
    chains = 2
    il = engine.Td3DiscreteInnerLoop(cfg, chains, want_final_params=True)
    keys = np.array([901, 902], np.uint64)
    kt = _dev(keys.view(np.int64))
    init = il.draw_agent_init(kt).cpu().numpy()
    # nn.Linear default init: |w| <= 1 / sqrt(fan_in) per layer of the actor
    S, H = 6, 510
    assert np.abs(init[:, :S * H]).max() <= 1 / np.sqrt(S) and np.abs(init[:, S * H + H:S * H + H + H * H]).max() <= 1 / np.sqrt(H)
    rng = np.random.RandomState(1)
    theta = (rng.randn(il.p_theta) * 0.1).astype(np.float32)
    il.run(_dev(theta), None, None, None, None, rng_keys=kt)
    torch.cuda.synchronize()
    assert il.status.cpu().tolist() == [0, 0]
    for c in range(chains):
        o = orc.td3d_chain(ocfg, theta, init[c], rng_key=int(keys[c]))
        assert o["rc"] == 0 and o["learn_steps"] == 40
        assert float(il.score[c]) == o["score"]
        assert np.array_equal(il.final_params[c].cpu().numpy(), o["final_params"])
