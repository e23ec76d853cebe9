"""Pins the CPU oracle (oracle/lenv_oracle.c) to golden vectors produced by the reference itself
(oracle/gen_golden.py imports /root/reference; the .npz files under tests/golden are its outputs).

Tolerances: the oracle's batched dot-product order equals torch-CPU's (bitwise, verified), so the
only deviations are (a) the oracle's polynomial tanh vs torch's (few ulp), (b) torch's batch-1 gemv
order, (c) torch's batch-reduction order in backward.  All are O(1e-7) relative per op.
"""
import numpy as np
import pytest

from oracle import oracle as orc

ACTS = ["identity", "relu", "leakyrelu", "tanh", "prelu"]


def test_tanh_accuracy():
    x = np.concatenate([np.linspace(-12, 12, 20001), np.logspace(-30, 1, 4000), -np.logspace(-30, 1, 4000), [0.0]]).astype(np.float32)
    got = orc.tanhf(x)
    ref = np.tanh(x.astype(np.float64))
    err = np.abs(got - ref)
    assert np.max(err) < 7e-8                 # absolute (v4: the argument is not rounded on the way; v3 was 1.1e-7)
    assert np.array_equal(got, -orc.tanhf(-x))
    assert got[-1] == 0.0
    assert np.all(np.abs(got) <= 1.0)


def test_tanh_matches_the_documented_algorithm():
    """The canonical tanh (v4) restated in numpy from its description (tools/gen_tanh_table.py: one float addition of 2^19 rounds
    min(|x|, TMAX) to the 1/16 grid, the table entry is the low mantissa bits of the sum, d = t - (sum - 2^19), three fmaf) gives the
    oracle's bits, and the committed table is the one the generator fits."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gen_tanh_table", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_tanh_table.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    tab = G.fit()
    x = np.concatenate([np.linspace(-12, 12, 200001), np.logspace(-30, 1.2, 20000), -np.logspace(-30, 1.2, 20000), [0.0, 9.124999, 9.125, 1e30, -1e30]]).astype(np.float32)
    assert np.array_equal(G.eval32(tab, x), orc.tanhf(x))
    assert G.N == 147 and float(G.MAGIC) == 524288.0


def test_tanh_range_exhaustive():
    """Every float in [0, 16] (1.1e9 values): result in [0, 1]; across table-interval seams the result never steps down by
    more than 2 ulp(1) (the canonical tanh has no final clamp, so the table itself must guarantee the bound)."""
    bad, mx = orc.tanhf_scan(0.0, 16.0, slack=2.4e-7)
    assert bad == 0 and mx == 1.0


def test_sincos_accuracy():
    L = orc.lib()
    xs = np.concatenate([np.linspace(-12, 12, 5001), np.random.RandomState(0).uniform(-0.3, 0.3, 3000)])
    s = np.array([L.orc_sin(float(v)) for v in xs])
    c = np.array([L.orc_cos(float(v)) for v in xs])
    assert np.max(np.abs(s - np.sin(xs))) < 4e-16
    assert np.max(np.abs(c - np.cos(xs))) < 4e-16


def test_g1_virtual_env_step(golden):
    g = golden("g1_virtual_env_step")
    for ci in range(int(g["n_cases"])):
        pre = "c%02d_" % ci
        S, A, H, L, act = [int(v) for v in g[pre + "meta"]]
        descs = orc.se_descs(S, A, H, L, ACTS[act])
        n = g[pre + "state"].shape[0]
        ns, r, d = orc.se_step_population(descs, g[pre + "theta"], None, None, None, g[pre + "state"], g[pre + "action"])
        np.testing.assert_allclose(ns, g[pre + "next_state"], rtol=2e-6, atol=2e-6, err_msg=pre)
        np.testing.assert_allclose(r, g[pre + "reward"], rtol=2e-6, atol=2e-6, err_msg=pre)
        np.testing.assert_allclose(d, g[pre + "done"], rtol=2e-6, atol=2e-6, err_msg=pre)
        assert ns.shape == (n, S)


def test_g3_critic_dqn_forward(golden):
    g = golden("g3_critic_dqn_forward")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        S, A, H, L, act = [int(v) for v in g[pre + "meta"]]
        d = orc.mlp_desc(S, H, L, A, ACTS[act])
        y = orc.mlp_forward(d, g[pre + "params"], g[pre + "x"])
        if ACTS[act] != "tanh":
            # batched linear order == torch's: bit-exact
            assert np.array_equal(y, g[pre + "y"]), pre
        else:
            np.testing.assert_allclose(y, g[pre + "y"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(y[:5], g[pre + "y_single"], rtol=2e-6, atol=2e-6)


def _cfg_for(meta, hp, grad_chunk):
    S, A, H, L, act, B, _ = [int(v) for v in meta]
    return orc.DdqnCfg(env_id=0 if S == 4 else 1, state_dim=S, num_actions=A, max_steps=200, se_hidden=8, se_layers=1,
                       se_act=2, se_prelu=0.25, q_hidden=H, q_layers=L, q_act=act, q_prelu=0.25, batch_size=B,
                       rb_size=1000, train_episodes=1, test_episodes=1, init_episodes=0, early_out_num=1,
                       grad_chunk=grad_chunk, rng_mode=0, solved_reward=1e9, gamma=float(hp[0]), lr=float(hp[1]),
                       tau=float(hp[2]), eps_init=1.0, eps_min=0.1, eps_decay=0.9, adam_beta1=0.9, adam_beta2=0.999,
                       adam_eps=1e-8)


@pytest.mark.parametrize("grad_chunk", [0, 13])
def test_g4_ddqn_learn(golden, grad_chunk):
    g = golden("g4_ddqn_learn")
    for vi in range(int(g["n_variants"])):
        pre = "v%d_" % vi
        cfg = _cfg_for(g[pre + "meta"], g[pre + "hparams"], grad_chunk)
        nsteps = int(g[pre + "meta"][6])
        online, target = g[pre + "online0"].copy(), g[pre + "target0"].copy()
        m, v = np.zeros_like(online), np.zeros_like(online)
        b1p, b2p = 1.0, 1.0
        for step in range(nsteps):
            loss, online, target, m, v, b1p, b2p = orc.ddqn_learn(cfg, online, target, m, v, step + 1, b1p, b2p, g[pre + "rows"][step])
            assert abs(loss - g[pre + "loss"][step]) <= 2e-6 * max(1.0, abs(g[pre + "loss"][step])), (pre, step)
            np.testing.assert_allclose(m, g[pre + "adam_m"][step], rtol=2e-4, atol=2e-8, err_msg=pre + "m%d" % step)
            np.testing.assert_allclose(v, g[pre + "adam_v"][step], rtol=2e-4, atol=1e-12, err_msg=pre + "v%d" % step)
            # Adam's first steps move every weight by ~lr regardless of |g|; compare the parameters tightly
            np.testing.assert_allclose(online, g[pre + "online"][step], rtol=0, atol=3e-6, err_msg=pre + "online%d" % step)
            np.testing.assert_allclose(target, g[pre + "target"][step], rtol=0, atol=3e-6, err_msg=pre + "target%d" % step)


def test_g6_worker_noise(golden):
    g = golden("g6_worker_noise")
    theta, eps = g["theta"], g["eps"]
    P = theta.size
    descs = orc.se_descs(4, 2, 83, 1, "leakyrelu")
    assert sum(orc.mlp_num_params(d) for d in descs) == P
    # theta +/- eps (GTN_worker.py:165-175) is what the population step applies: fmaf(sign, eps, theta)
    plus = np.float32(1.0) * eps + theta
    assert np.array_equal((theta + eps).astype(np.float32), g["theta_plus"])
    assert np.array_equal((theta - eps).astype(np.float32), g["theta_minus"])
    assert np.array_equal(plus.astype(np.float32), g["theta_plus"])
    best, sign = orc.worker_best(g["score_add"], g["score_sub"], mirrored=True)
    assert np.array_equal(best, g["score_best"])
    for i in range(3):
        assert np.array_equal(sign[i] * eps, g["eps_after"][i])
        assert np.array_equal((theta + sign[i] * eps).astype(np.float32), g["env_after"][i])


def test_g7_master(golden):
    g = golden("g7_master")
    for t in range(8):
        got = orc.score_transform(t, g["scores"], g["scores_orig"])
        np.testing.assert_allclose(got, g["tf%d" % t], rtol=1e-15, atol=1e-15, err_msg="type %d" % t)
        tied = orc.score_transform(t, g["tied"], g["scores_orig"])
        ref = g["tf%d_tied" % t]
        # ties: np.argsort's order is implementation-defined; the multiset of weights and the weight of
        # every untied entry must agree
        np.testing.assert_allclose(np.sort(tied), np.sort(ref), rtol=1e-15, atol=1e-15, err_msg="tied type %d" % t)
        for val in np.unique(g["tied"]):
            sel = g["tied"] == val
            np.testing.assert_allclose(tied[sel].sum(), ref[sel].sum(), rtol=1e-14, atol=1e-15)
    with pytest.raises(ValueError):
        orc.score_transform(9, g["scores"], g["scores_orig"])
    pop = g["eps"].shape[0]
    sign = np.ones(pop, np.float32)
    th1 = orc.update_env(g["theta0"], g["eps"], sign, g["weights"], float(g["step_size"]))
    assert np.array_equal(th1, g["theta1"])
    th2 = orc.update_env(th1, g["eps"], sign, g["weights"], float(g["step_size"]), nes_step_size=True, weight_decay=0.01)
    assert np.array_equal(th2, g["theta2"])


@pytest.mark.parametrize("name,chunk", [("g8_calc_score_cartpole_a", 13), ("g8_calc_score_cartpole_b", 13),
                                        ("g8w_calc_score_cartpole_ringwrap", 13),
                                        ("g8l2_calc_score_acrobot_ddqn_2layer", 0),    # Critic_DQN 6-128-128-3: batch gradient
                                        ("g8ln_calc_score_acrobot_ddqn_layernorm", 0),  # use_layer_norm: Critic_DQN 6-40-40-3 with the LayerNorm behind its second Linear
                                        ("g8seln_calc_score_acrobot_ddqn_se_layernorm", 0),  # the ENV's use_layer_norm: SE nets 9-32-32-x normalise behind their second Linear
                                        ("g8m_calc_score_mountaincar_ddqn", 0)])       # default_config_mountaincar.yaml's pair
def test_g8_calc_score_trace(golden, name, chunk):
    import json
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=chunk, rng_mode=1, train_episodes=int(g["train_episodes"]),
                                   max_steps=int(g["max_steps"]))
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"],
                           g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10)
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["action"].size == n
    assert np.array_equal(tr["explored"], g["tr_explored"])
    assert np.array_equal(tr["action"], g["tr_action"])
    np.testing.assert_allclose(tr["state"], g["tr_state"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr["reward"], g["tr_reward"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr["done"], g["tr_done"], rtol=1e-5, atol=1e-5)
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3, atol=1e-6)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["final_test_returns"], g["reward_list_test"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("name", ["g8v_calc_score_cartpole_ddqn_vary", "g8v2_calc_score_cartpole_ddqn_vary_wide",
                                  "g8vd_calc_score_acrobot_dueling_vary"])
def test_g8v_vary_agents_replay_of_the_recorded_draw(golden, name):
    """DDQN_vary / DuelingDDQN_vary (agents/DDQN_vary.py:26-59): the reference run drew its hyper-parameters through the
    ConfigSpace stand-in; the oracle replays the run with the recorded draw (lr, batch 204/555/145, width 129/161/108,
    2/1/3 hidden layers) and the recorded tapes."""
    import json
    g = golden(name)
    cfgd, hp = json.loads(str(g["config_json"])), json.loads(str(g["hp_json"]))
    assert cfgd["agents"]["gtn"]["agent_name"].endswith("_vary")
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]),
                                   max_steps=int(g["max_steps"]), **orc.hp_overrides(hp))
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"],
                           g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10)
    assert out["rc"] == 0
    tr = out["trace"]
    assert np.array_equal(tr["action"], g["tr_action"]) and np.array_equal(tr["explored"], g["tr_explored"])
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr["reward"], g["tr_reward"], rtol=1e-5, atol=1e-5)
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3, atol=1e-6)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("name", ["g8i_calc_score_cartpole_ddqn_icm", "g8ia_calc_score_acrobot_dueling_icm"])
def test_g8i_agents_with_icm(golden, name):
    """DDQN / DuelingDDQN with the Intrinsic Curiosity Module inside learn() (agents/DDQN.py:74-76, models/icm_baseline.py):
    the reference's run (select_agent "ddqn_icm" / "duelingddqn_icm") replayed by the oracle -- same actions, DDQN losses
    (which see the intrinsic rewards), and the ICM parameters after every update of the run within 2e-7 (they move 2e-3).
    CartPole exercises the BCE inverse loss on one logit, Acrobot the cross-entropy over three."""
    import json
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.icm_enabled == 1 and orc.icm_num_params(cfg) == g["icm_init"].size
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10, icm_init=g["icm_init"])
    assert out["rc"] == 0
    tr = out["trace"]
    assert np.array_equal(tr["action"], g["tr_action"]) and np.array_equal(tr["explored"], g["tr_explored"])
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=1e-5, atol=1e-5)
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4, atol=1e-7)
    assert np.abs(g["icm_final"] - g["icm_init"]).max() > 1e-3                 # the module really trained
    np.testing.assert_allclose(out["icm_final"], g["icm_final"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4
    # without fresh ICM parameters an ICM config is refused
    assert orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes)["rc"] != 0


def test_mountaincar_step_physics():
    """gym==0.17.3 classic_control/mountain_car.py (third party, absent from the reference tree; restated): the oracle's step
    against a line-by-line float64 restatement over an episode that reaches the flag, the left wall, and the velocity clip."""
    import ctypes as C
    import math
    st = (C.c_double * 4)(-0.5, 0.0, 0.0, 0.0)
    rew, dn = C.c_double(), C.c_int()
    pos, vel = -0.5, 0.0
    hit_wall = reached = False
    for t in range(400):
        a = 0 if t < 60 else (2 if vel >= 0 else 0)          # first run into the left wall, then swing up to the flag
        orc.lib().orc_mountaincar_step(st, a, C.byref(rew), C.byref(dn))
        vel += (a - 1) * 0.001 + math.cos(3 * pos) * (-0.0025)
        vel = min(max(vel, -0.07), 0.07)
        pos += vel
        pos = min(max(pos, -1.2), 0.6)
        if pos == -1.2 and vel < 0:
            vel = 0.0
            hit_wall = True
        assert abs(st[0] - pos) <= 1e-13 and abs(st[1] - vel) <= 1e-14 and rew.value == -1.0
        # keep the two in lock-step so the comparison stays one step deep (orc_cos is a restated cos, not libm's)
        pos, vel = st[0], st[1]
        assert dn.value == int(pos >= 0.5 and vel >= 0)
        if dn.value:
            reached = True
            break
    assert hit_wall and reached and t < 399


def test_g1ln_mlp_with_layer_norm(golden):
    """build_nn_from_config with `use_layer_norm` (models/model_utils.py:22-37): ONE shared nn.LayerNorm after every hidden
    Linear but the first.  The oracle's forward against the reference module's (random LayerNorm affine), the package's
    builder against the reference's state-dict keys, and the flat Module.parameters() packing."""
    from learning_environments_amd.models.model_utils import build_nn_from_config, linear_params, mlp_desc, mlp_params
    g = golden("g1ln_mlp_layer_norm")
    acts = ["identity", "relu", "leakyrelu", "tanh", "prelu"]
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        din, dout, H, L, act = [int(v) for v in g[pre + "meta"]]
        d = orc.mlp_desc(din, H, L, dout, acts[act], use_layer_norm=True)
        assert orc.mlp_num_params(d) == g[pre + "params"].size == (din * H + H) + (L - 1) * (H * H + H) + (H * dout + dout) + (2 * H if L >= 2 else 0)
        y = orc.mlp_forward(d, g[pre + "params"], g[pre + "x"])
        np.testing.assert_allclose(y, g[pre + "y"], rtol=2e-5, atol=2e-6)
        net = build_nn_from_config(din, dout, {"hidden_size": H, "hidden_layer": L, "activation_fn": acts[act], "use_layer_norm": True})
        assert list(net.state_dict().keys()) == [str(k) for k in g[pre + "keys"]]
        assert sum(p.numel() for p in mlp_params(net)) == g[pre + "params"].size
        # the NES layout stays Linear-only: the shared LayerNorm's affine is never perturbed (GTN_worker.py:158)
        assert sum(p.numel() for p in linear_params(net)) == g[pre + "params"].size - (2 * H if L >= 2 else 0)
        md = mlp_desc(net, acts[act])
        assert md.use_layer_norm == (1 if L >= 2 else 0) and md.layers == L
    # a plain MLP still has none
    assert mlp_desc(build_nn_from_config(4, 2, {"hidden_size": 8, "hidden_layer": 2, "activation_fn": "relu"}), "relu").use_layer_norm == 0


@pytest.mark.parametrize("name", ["g8r_calc_score_cartpole_ddqn_reward_env", "g8r6_calc_score_cartpole_ddqn_reward_env_t6",
                                  "g8mr_calc_score_mountaincar_ddqn_reward_env",
                                  "g8rl_calc_score_cartpole_ddqn_reward_env_2layer",        # reward net 4-24-24-1
                                  "g8rln_calc_score_cartpole_ddqn_reward_env_layernorm"])   # the same with use_layer_norm in the env's section (type 1)
def test_g8r_ddqn_on_a_reward_env(golden, name):
    """default_config_cartpole_reward_env.yaml's experiment (synthetic_env_type 1): DDQN trains on a RewardEnv over the real
    CartPole -- real transitions, reward through the reward network (type 2 with a PReLU net / type 6 with tanh)."""
    import json
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.synthetic_env_type == 1 and cfg.reward_env_type in (1, 2, 6)
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10)
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["action"].size == n and np.array_equal(tr["action"], g["tr_action"]) and np.array_equal(tr["done"], g["tr_done"])
    assert np.array_equal(tr["next_state"], g["tr_next_state"])                      # the real env's fp64 step, cast once
    np.testing.assert_allclose(tr["reward"], g["tr_reward"], rtol=0, atol=1e-6)      # shaped rewards
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4, atol=1e-7)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("name,k", [("g8k_calc_score_cartpole_ddqn_same_action_2", 2), ("g8kd_calc_score_acrobot_duelingddqn_same_action_3", 3),
                                    ("g8kr_calc_score_cartpole_ddqn_reward_env_same_action_2", 2)])
def test_g8k_same_action_num(golden, name, k):
    """same_action_num > 1 in the DDQN family (agents/base_agent.py:104,122,194; envs/env_wrapper.py:24-29 virtual: every repeat runs,
    fp32 reward sum; :56-61 real: the repeats stop at done, python-float sum): reference runs on a CartPole / Acrobot VirtualEnv and
    on the CartPole RewardEnv replayed from their recorded draws."""
    import json
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.same_action_num == k
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10)
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["action"].size == n and np.array_equal(tr["action"], g["tr_action"])
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr["done"], g["tr_done"], rtol=0, atol=2e-6)
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4, atol=1e-7)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])             # episode_length += same_action_num per agent step
    assert out["episode_len"].max() > g["tr_action"].size / len(out["episode_len"])  # ... so it exceeds the number of agent steps
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["final_test_returns"], g["reward_list_test"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


def test_vary_hyperparameter_draw():
    """The package's sampler (agents/vary.py) against the oracle's numpy restatement of ConfigSpace 0.4.13 on the same
    uniforms, the reference's bounds, and the log-uniform shape of the draw."""
    from learning_environments_amd.agents import vary
    sec = dict(lr=1e-3, batch_size=199, hidden_size=57, hidden_layer=1)
    rs = np.random.RandomState(4)
    draws = []
    for _ in range(4000):
        u = rs.rand(4)
        d = vary.vary_hyperparameters(sec, u)
        o = orc.vary_hyperparameters(sec, u)
        assert d["batch_size"] == o["batch_size"] and d["hidden_size"] == o["hidden_size"] and d["hidden_layer"] == o["hidden_layer"]
        assert abs(d["lr"] - o["lr"]) <= 1e-15
        draws.append(d)
    b = np.array([d["batch_size"] for d in draws]); h = np.array([d["hidden_size"] for d in draws])
    l = np.array([d["hidden_layer"] for d in draws]); lr = np.array([d["lr"] for d in draws])
    assert b.min() >= 66 and b.max() <= 597 and h.min() >= 19 and h.max() <= 171 and set(l.tolist()) == {0, 1, 2}
    assert lr.min() >= 1e-3 / 3 and lr.max() <= 3e-3
    # log-uniform: the median sits at the geometric centre, a third of the integer draws per layer count
    assert abs(np.median(np.log(lr)) - np.log(1e-3)) < 0.08 and abs(np.median(np.log(b)) - 0.5 * (np.log(65.5) + np.log(597.5))) < 0.08
    assert np.all(np.abs(np.bincount(l) / l.size - 1 / 3) < 0.03)
    # edge of the range: u -> 0 / 1 hit the bounds exactly
    lo = vary.vary_hyperparameters(sec, [0.0, 0.0, 0.0, 0.0]); hi = vary.vary_hyperparameters(sec, [1.0 - 2 ** -53] * 4)
    assert (lo["batch_size"], lo["hidden_size"], lo["hidden_layer"]) == (66, 19, 0)
    assert (hi["batch_size"], hi["hidden_size"], hi["hidden_layer"]) == (597, 171, 2)


def test_g10_gridworld_tables(golden):
    """The package's table compiler (envs/gridworld.py) against tables produced by stepping the reference's classes."""
    from learning_environments_amd.envs.gridworld import transition_tables
    g = golden("g10_gridworld_tables")
    for name in [str(n) for n in g["names"]]:
        t = transition_tables(name)
        ok = ~g[name + "_walls"]
        assert t["start_state"] == int(g[name + "_start"])
        assert np.array_equal(t["next_state"][ok], g[name + "_next"][ok]), name
        assert np.array_equal(t["reward"][ok], g[name + "_reward"][ok]), name
        assert np.array_equal(t["done"][ok], g[name + "_done"][ok]), name


def _ql_cfg(cfgd, rng_mode=1, **over):
    from learning_environments_amd.envs.gridworld import transition_tables
    tables = transition_tables(cfgd["env_name"])
    return orc.ql_cfg_from_config(cfgd, tables, rng_mode=rng_mode, **over), tables


def test_g2_reward_env_shaping(golden):
    import json
    g = golden("g2_reward_env_cliff")
    cfgd = json.loads(str(golden("g9_calc_score_cliff_a")["config_json"]))
    for t in [int(v) for v in g["types"]]:
        for act, layers in (("prelu", 1), ("tanh", 2)):
            cfg, tables = _ql_cfg(cfgd, reward_env_type=t, rn_act=orc.ACT[act], rn_layers=layers)
            key = "t%d_%s%d_" % (t, act, layers)
            _, shaped = orc.rn_shaped_rewards(cfg, g[key + "theta"], tables)
            if t == 0:
                assert np.array_equal(shaped.astype(np.float64), g[key + "shaped"])
            else:
                np.testing.assert_allclose(shaped, g[key + "shaped"], rtol=2e-6, atol=2e-6, err_msg=key)


@pytest.mark.parametrize("name", ["g9_calc_score_cliff_a", "g9_calc_score_cliff_b", "g9s_calc_score_cliff_sarsa", "g9c_calc_score_cliff_ql_cb",
                                  "g9sc_calc_score_cliff_sarsa_cb", "g9i_calc_score_cliff_ql_init2",
                                  "g9k_calc_score_cliff_ql_same_action_2", "g9ks_calc_score_cliff_sarsa_same_action_3",     # same_action_num 2 / 3
                                  "g9ln_calc_score_cliff_ql_reward_net_layernorm"])     # the ENV section's use_layer_norm: two-hidden-layer reward net
def test_g9_calc_score_cliff(golden, name):
    """cfg 4: integer-state path.  Trajectories, Q-table argmax decisions, episode lengths and returns are EXACT -- for QL and
    for the other tabular agents of select_agent (SARSA, count-based QL / SARSA) and with init_episodes > 0."""
    import json
    g = golden(name)
    cfg, tables = _ql_cfg(json.loads(str(g["config_json"])))
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], np.zeros(0, np.int32), np.zeros((0, 4)), np.zeros((0, 4)))
    n = g["tr_action"].size
    # phi's 32-term output dot product runs through torch's batch-1 gemv, whose summation order is MKL-defined; a 1-ulp
    # difference in a shaped reward can flip an argmax between two near-tied Q entries.  The integer path is therefore
    # pinned with the reference's own shaped-reward table as input (phi itself is pinned by G2 at 2e-6).
    _, shaped = orc.rn_shaped_rewards(cfg, g["theta"], tables)
    np.testing.assert_allclose(shaped, g["shaped_ref"], rtol=2e-6, atol=2e-6)
    out = orc.ql_rn_chain(cfg, g["theta"], tables, tapes=tapes, trace_cap=n + 4, shaped_override=g["shaped_ref"])
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["action"].size == n
    assert np.array_equal(tr["action"] & 0xFFFF, g["tr_action"])
    if cfg.agent_kind == 0:      # (for SARSA the fixture's explored flag also counts the draws of learn's next_action)
        assert np.array_equal(tr["action"] >> 16, g["tr_explored"])
    assert np.array_equal(tr["state"], g["tr_state"])
    assert np.array_equal(tr["next_state"], g["tr_next_state"])
    assert np.array_equal(tr["done"], g["tr_done"])
    assert np.array_equal(tr["reward"], g["tr_reward"])
    assert np.array_equal(out["q_table"], g["q_table"])          # fp64 Q-table: bit-exact
    assert np.array_equal(out["episode_len"][:g["episode_length_train"].size], g["episode_length_train"])
    assert np.array_equal(out["episode_test_mean"][:g["reward_list_train"].size], g["reward_list_train"])
    assert np.array_equal(out["final_test_returns"], g["reward_list_test"])
    assert out["score"] == float(g["score"])


def _dueling_cfg(meta, hp, grad_chunk):
    S, A, H, L, F, act, B, _ = [int(v) for v in meta]
    return orc.DdqnCfg(env_id=0 if S == 4 else 1, state_dim=S, num_actions=A, max_steps=200, se_hidden=8, se_layers=1, se_act=2,
                       se_prelu=0.25, q_hidden=H, q_layers=L, q_act=act, q_prelu=0.25, batch_size=B, rb_size=1000,
                       train_episodes=1, test_episodes=1, init_episodes=0, early_out_num=1, grad_chunk=grad_chunk, rng_mode=0,
                       agent_kind=1, feature_dim=F, solved_reward=1e9, gamma=float(hp[0]), lr=float(hp[1]), tau=float(hp[2]),
                       eps_init=1.0, eps_min=0.1, eps_decay=0.9, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)


@pytest.mark.parametrize("grad_chunk", [0, 7])
def test_g4d_dueling_forward_and_learn(golden, grad_chunk):
    g = golden("g4d_dueling_learn")
    for vi in range(int(g["n_variants"])):
        pre = "v%d_" % vi
        cfg = _dueling_cfg(g[pre + "meta"], g[pre + "hparams"], grad_chunk)
        assert orc.dueling_num_params(cfg) == g[pre + "online0"].size
        # forward: global advantage mean over batch x actions (actor_critic.py:121) and the per-state mean for single inputs
        np.testing.assert_allclose(orc.dueling_forward(cfg, g[pre + "online0"], g[pre + "fwd_x"]), g[pre + "fwd_q"], rtol=2e-6, atol=2e-6)
        for i in range(4):
            np.testing.assert_allclose(orc.dueling_forward(cfg, g[pre + "online0"], g[pre + "fwd_x"][i:i + 1])[0],
                                       g[pre + "fwd_q_single"][i], rtol=2e-6, atol=2e-6)
        online, target = g[pre + "online0"].copy(), g[pre + "target0"].copy()
        m, v = np.zeros_like(online), np.zeros_like(online)
        b1p, b2p = 1.0, 1.0
        for step in range(int(g[pre + "meta"][7])):
            loss, online, target, m, v, b1p, b2p = orc.dueling_learn(cfg, online, target, m, v, b1p, b2p, g[pre + "rows"][step])
            assert abs(loss - g[pre + "loss"][step]) <= 3e-6 * max(1.0, abs(g[pre + "loss"][step])), (pre, step)
            np.testing.assert_allclose(online, g[pre + "online"][step], rtol=0, atol=2e-5, err_msg=pre + "online%d" % step)
            np.testing.assert_allclose(target, g[pre + "target"][step], rtol=0, atol=2e-5, err_msg=pre + "target%d" % step)


@pytest.mark.parametrize("name", ["g8d_calc_score_acrobot_dueling", "g8df_calc_score_acrobot_dueling_fullshape",
                                  "g8dln_calc_score_acrobot_dueling_layernorm"])     # use_layer_norm: feature stream 6-24-24-24-16, ONE LayerNorm at two positions
def test_g8d_calc_score_acrobot_dueling(golden, name):
    import json
    g = golden(name)
    cfgd = json.loads(str(g["config_json"]))
    cfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=0, rng_mode=1, train_episodes=int(g["train_episodes"]), max_steps=int(g["max_steps"]))
    assert cfg.agent_kind == 1 and cfg.feature_dim == (128 if name.endswith("fullshape") else 16)
    assert cfg.q_layer_norm == (1 if name.endswith("layernorm") else 0)
    assert g["agent_init"].size == orc.dueling_num_params(cfg)
    tapes = orc.make_tapes(g["tape_eps_uniform"], g["tape_rand_action"], g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_action"].size
    out = orc.ddqn_se_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 10)
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["action"].size == n
    assert np.array_equal(tr["action"], g["tr_action"])
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=1e-5, atol=1e-5)
    losses = tr["loss"][~np.isnan(tr["loss"])]
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-3, atol=1e-6)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


def _td3_cfg(meta, hp):
    H, L, act, B, delay, _ = [int(v) for v in meta]
    return orc.Td3Cfg(env_id=2, state_dim=17, action_dim=6, max_steps=10, rn_hidden=8, rn_layers=1, rn_act=4, rn_prelu=0.25,
                      reward_env_type=2, hidden=H, layers=L, act=act, prelu=0.25, batch_size=B, rb_size=1000, train_episodes=1,
                      test_episodes=1, init_episodes=0, early_out_num=1, policy_delay=delay, rng_mode=0, solved_reward=1e9,
                      gamma=float(hp[0]), lr=float(hp[1]), tau=float(hp[2]), action_std=0.05, policy_std=float(hp[3]),
                      policy_std_clip=float(hp[4]), max_action=1.0, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)


def test_g4t_td3_forward_and_learn(golden):
    g = golden("g4t_td3_learn")
    for vi in range(int(g["n_variants"])):
        pre = "v%d_" % vi
        cfg = _td3_cfg(g[pre + "meta"], g[pre + "hparams"])
        Pa, Pc = orc.td3_param_counts(cfg)
        assert Pa + 2 * Pc == g[pre + "params0"].size
        p0 = g[pre + "params0"]
        np.testing.assert_allclose(orc.td3_actor_forward(cfg, p0[:Pa], g[pre + "fwd_s"]), g[pre + "fwd_actor"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(orc.td3_critic_forward(cfg, p0[Pa:Pa + Pc], g[pre + "fwd_s"], g[pre + "fwd_a"]), g[pre + "fwd_critic1"],
                                   rtol=2e-6, atol=2e-6)
        params, targets = p0.copy(), g[pre + "targets0"].copy()
        m, v = np.zeros_like(params), np.zeros_like(params)
        pows = [1.0, 1.0, 1.0, 1.0]
        for step in range(int(g[pre + "meta"][5])):
            params, targets, m, v, pows, _ = orc.td3_learn(cfg, params, targets, m, v, pows, step + 1, g[pre + "rows"][step],
                                                           g[pre + "policy_noise"][step])
            # lr = 3e-3: Adam moves a weight by lr*g/(|g|+1e-8) on its first steps, so a weight whose gradient is ~1e-8 is
            # sensitive to rounding-level differences in g; a wrong sign or a missing term would show as >= 3e-3
            np.testing.assert_allclose(params, g[pre + "params"][step], rtol=0, atol=1.5e-4, err_msg=pre + "params%d" % step)
            np.testing.assert_allclose(targets, g[pre + "targets"][step], rtol=0, atol=1.5e-4, err_msg=pre + "targets%d" % step)
            assert np.mean(np.abs(params - g[pre + "params"][step]) > 3e-6) < 0.01


def test_cheetah_standin_matches_shim_run(golden):
    """The stand-in dynamics inside the oracle reproduce, bit for bit, the next states / raw rewards of the shim env the
    reference ran on (plain float64 arithmetic, same order)."""
    import ctypes as C
    g = golden("g8t_calc_score_cheetah_td3")
    L = orc.lib()
    x = (C.c_double * 17)(*g["tape_train_reset"][0])
    rew = C.c_double()
    for k in range(7):                       # first training episode
        a = (C.c_float * 6)(*g["tr_action"][k])
        L.orc_cheetah_step(x, a, C.byref(rew))
        assert np.array_equal(np.array(list(x)).astype(np.float32), g["tr_next_state"][k])


@pytest.mark.parametrize("name", ["g8t_calc_score_cheetah_td3", "g8tf_calc_score_cheetah_td3_fullshape",
                                  "g8tln_calc_score_cheetah_td3_layernorm", "g8tln3_calc_score_cheetah_td3_layernorm_3layer"])
def test_g8t_calc_score_cheetah_td3(golden, name):
    """The *_fullshape fixture is BASELINE configs[4] at its real network shapes (128x2 actor/critics, B 192, RN hidden 128); the
    *_layernorm ones have `use_layer_norm: True` in the td3 section (one shared nn.LayerNorm per net, at one / two positions)."""
    import json
    g = golden(name)
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert cfg.use_layer_norm == (1 if "layernorm" in name else 0)
    assert g["agent_init"].size == sum(orc.td3_param_counts(cfg)) + orc.td3_param_counts(cfg)[1]
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4)
    assert out["rc"] == 0
    tr = out["trace"]
    assert tr["reward"].size == n
    np.testing.assert_allclose(tr["action"], g["tr_action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(tr["next_state"], g["tr_next_state"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(tr["reward"], g["tr_reward"], rtol=0, atol=5e-5)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    # north_star: returns within 1e-4 of the reference (measured here: 2.4e-7 on the score)
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["final_test_returns"], g["reward_list_test"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


def test_g8tv_td3_vary_replay_of_the_recorded_draw(golden):
    """TD3_vary (agents/TD3_vary.py:24-58): the reference run drew batch 145 / width 108 / 3 hidden layers / its own lr
    through the ConfigSpace stand-in; the oracle replays it with the recorded draw."""
    import json
    g = golden("g8tv_calc_score_cheetah_td3_vary")
    hp = json.loads(str(g["hp_json"]))
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1, lr=float(hp["lr"]), batch_size=int(hp["batch_size"]),
                                  hidden=int(hp["hidden_size"]), layers=max(1, int(hp["hidden_layer"])))
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4)
    assert out["rc"] == 0 and out["trace"]["reward"].size == n
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["trace"]["reward"], g["tr_reward"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4


def test_g8ti_td3_with_icm(golden):
    """select_agent "td3_icm" = TD3(icm=True) (agents/TD3.py:44-60,68-70): continuous actions, so the ICM's inverse loss is an
    MSE on the action vector.  Reference run replayed by the oracle; ICM parameters after the run within 2e-7."""
    import json
    g = golden("g8ti_calc_score_cheetah_td3_icm")
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert cfg.icm_enabled == 1 and orc.td3_icm_num_params(cfg) == g["icm_init"].size
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4, icm_init=g["icm_init"])
    assert out["rc"] == 0
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    assert np.abs(g["icm_final"] - g["icm_init"]).max() > 1e-3
    np.testing.assert_allclose(out["icm_final"], g["icm_final"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4
    assert orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes)["rc"] != 0


@pytest.mark.parametrize("name", ["g8ts_calc_score_cheetah_td3_virtual_env", "g8tseln_calc_score_cheetah_td3_virtual_env_layernorm"])
def test_g8ts_td3_on_a_virtual_env(golden, name):
    """default_config_halfcheetah.yaml's combination (synthetic_env_type 0): TD3 trains on a VirtualEnv -- three SE nets with two
    hidden layers on cat(action, state) -- and is tested on the real (stand-in) env.  *_layernorm: `use_layer_norm` in the ENV's section
    (the SE nets' LayerNorm is never perturbed by NES: theta holds the nn.Linear parameters only)."""
    import json
    g = golden(name)
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert cfg.virtual_env == 1 and cfg.rn_layers == 2 and cfg.rn_layer_norm == int(name.endswith("layernorm"))
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4)
    assert out["rc"] == 0 and out["trace"]["reward"].size == n
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["trace"]["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    assert np.array_equal(out["episode_len"], g["episode_length_train"])
    assert abs(out["score"] - float(g["score"])) <= 1e-4


@pytest.mark.parametrize("name", ["g8pf_calc_score_pendulum_td3_virtual_env_fullshape", "g8hf_calc_score_cheetah_td3_virtual_env_fullshape"])
def test_g8_full_shape_virtual_env_td3(golden, name):
    """The td3 sections of default_config_pendulum.yaml (SE nets 4-32-32-x) and default_config_halfcheetah.yaml (SE nets 23-128-128-128-x)
    at their REAL shapes -- actor / critics 128 x 2, batch 256, policy_delay 2, ten test episodes per test phase: the shapes the TD3
    wave-chain kernel's instantiations 5 and 6 run.  The reference's runs (agent td3 = td3_vary with vary_hp off) replayed by the oracle:
    traces, per-episode test means, the ten final returns and ALL final parameters."""
    import json
    g = golden(name)
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.policy_delay, cfg.virtual_env, cfg.test_episodes) == (128, 2, 256, 2, 1, 10)
    assert (cfg.rn_hidden, cfg.rn_layers) == ((32, 2) if cfg.env_id == 4 else (128, 3))
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=cfg.action_dim, S=orc.TD3_STATE_WORDS[cfg.env_id])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4, want_final_params=True)
    assert out["rc"] == 0 and out["trace"]["reward"].size == n and out["learn_steps"] >= 14
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["trace"]["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    m = g["episode_length_train"].size
    assert np.array_equal(out["episode_len"][:m], g["episode_length_train"]) and out["episodes_run"] == m      # early-out like the reference
    np.testing.assert_allclose(out["episode_test_mean"][:m], g["reward_list_train"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["final_test_returns"], g["reward_list_test"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4
    np.testing.assert_allclose(out["final_params"], g["final_params"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["g8p_calc_score_pendulum_td3_virtual_env", "g8pr_calc_score_pendulum_td3_reward_env",
                                  "g8trnln_calc_score_pendulum_td3_reward_net_layernorm"])     # the ENV section's use_layer_norm (reward net 3-20-20-1)
def test_g8p_td3_on_pendulum(golden, name):
    """default_config_pendulum.yaml / default_config_pendulum_reward_env.yaml's env: TD3 (max_action 2) on a VirtualEnv of
    Pendulum-v0 and on a RewardEnv (type 2) over the real Pendulum; the reference's runs replayed by the oracle.  The real
    env's transitions (RewardEnv mode) agree to an ulp or two, the rest within the TD3 tolerances."""
    import json
    g = golden(name)
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert (cfg.env_id, cfg.state_dim, cfg.action_dim, cfg.max_action) == (4, 3, 1, 2.0)
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=1, S=orc.TD3_STATE_WORDS[4])
    n = g["tr_reward"].size
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4)
    assert out["rc"] == 0 and out["trace"]["reward"].size == n
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    if cfg.virtual_env:
        np.testing.assert_allclose(out["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=2e-6)
    else:
        # the real env's fp64 step, cast once -- driven by the oracle's OWN actions, which sit an ulp off torch's where the actor's
        # tanh does (5 of them in this run): the states agree to the last bit or two
        np.testing.assert_allclose(out["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=5e-7)
        assert np.abs(g["tr_action"]).max() > 1.0                                    # random actions span [-2, 2]
    np.testing.assert_allclose(out["trace"]["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    m = g["episode_length_train"].size
    assert np.array_equal(out["episode_len"][:m], g["episode_length_train"]) and out["episodes_run"] == m   # early-out like the reference
    np.testing.assert_allclose(out["episode_test_mean"][:m], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4
    # reward types that read the step's info dict do not exist on Pendulum (its info dict is empty)
    cfg.reward_env_type, cfg.info_dim = 3, 4
    if not cfg.virtual_env:
        assert orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes)["rc"] != 0


@pytest.mark.parametrize("name", ["g8c_calc_score_cmc_td3_virtual_env", "g8cr_calc_score_cmc_td3_reward_env",
                                  "g8cf_calc_score_cmc_td3_virtual_env_fullshape", "g8co_calc_score_cmc_td3_syn_env_opt_fullshape"])
def test_g8c_td3_on_mountaincar_continuous(golden, name):
    """default_config_cmc.yaml / default_config_cmc_reward_env.yaml's env with their same_action_num = 2: TD3 on a VirtualEnv of
    MountainCarContinuous-v0 and on a RewardEnv (type 2, tanh) over the real env.  Every chosen action is applied twice
    (EnvWrapper.step, env_wrapper.py:24-29,56-61), episode lengths count env steps (base_agent.py:122).  g8cf: default_config_cmc.yaml
    itself at its REAL shapes (actor 2-128-128-1, critics 3-128-128-1, batch 256, policy_delay 2, SE nets 3-96-96-x): 30 learn steps,
    15 delayed policy updates -- the shape the TD3 wave-chain kernel's fourth instantiation runs."""
    import json
    g = golden(name)
    cfg = orc.td3_cfg_from_config(json.loads(str(g["config_json"])), rng_mode=1)
    assert (cfg.env_id, cfg.state_dim, cfg.action_dim, cfg.max_action, cfg.same_action_num) == (5, 2, 1, 1.0, 2)
    tapes = orc.make_td3_tapes(g["tape_rand_action"], g["tape_act_noise"], g["tape_test_noise"], g["tape_policy_noise"],
                               g["tape_replay_idx"], g["tape_train_reset"], g["tape_test_reset"], A=1, S=2)
    n = g["tr_reward"].size
    full = name.endswith("fullshape")
    out = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], tapes=tapes, trace_cap=n + 4, want_final_params=full)
    assert out["rc"] == 0 and out["trace"]["reward"].size == n
    np.testing.assert_allclose(out["trace"]["action"], g["tr_action"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["trace"]["next_state"], g["tr_next_state"], rtol=0, atol=2e-6 if cfg.virtual_env else 1e-9)
    np.testing.assert_allclose(out["trace"]["reward"], g["tr_reward"], rtol=0, atol=2e-6)
    if full and "syn_env_opt" in name:
        # g8co: default_config_cmc_syn_env_opt.yaml at its REAL shapes (actor 2-64-1, critics 3-64-1 leakyrelu: ONE hidden layer; SE nets
        # 3-128-128-128-x relu) -- the shape the DIRECT instantiations of the TD3 GEMM-queue kernel run: 40 learn steps, 20 policy updates
        assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.policy_delay, cfg.rn_hidden, cfg.rn_layers, cfg.virtual_env) == (64, 1, 256, 2, 128, 3, 1)
        assert out["learn_steps"] == 40 and g["agent_init"].size == 899 and g["theta"].size == 101124
        np.testing.assert_allclose(out["final_params"], g["final_params"], rtol=0, atol=2e-6)
    elif full:
        assert (cfg.hidden, cfg.layers, cfg.batch_size, cfg.policy_delay, cfg.rn_hidden, cfg.rn_layers, cfg.virtual_env) == (128, 2, 256, 2, 96, 2, 1)
        assert out["learn_steps"] == 30 and g["agent_init"].size == 51331
        # all 51 331 parameters after 30 critic steps and 15 actor steps + soft updates (Adam's first steps move every weight by ~lr)
        np.testing.assert_allclose(out["final_params"], g["final_params"], rtol=0, atol=2e-6)
    assert np.array_equal(out["episode_len"], g["episode_length_train"]) and int(g["episode_length_train"][0]) == 2 * ((cfg.max_steps + 1) // 2)
    np.testing.assert_allclose(out["episode_test_mean"], g["reward_list_train"], rtol=0, atol=1e-4)
    assert abs(out["score"] - float(g["score"])) <= 1e-4
    # one env step per action again: a different run (twice as many actions per episode)
    cfg.same_action_num, cfg.rng_mode = 1, 0
    out1 = orc.td3_rn_chain(cfg, g["theta"], g["agent_init"], rng_key=5)
    assert out1["rc"] == 0 and out1["train_steps"] > out["train_steps"]


def test_mountaincar_continuous_step_physics():
    """gym==0.17.3 classic_control/continuous_mountain_car.py (third party, restated): the oracle's step against a line-by-line
    float64 restatement -- left wall, speed clip, the flag with its +100, the action cost on the UNclipped action."""
    import ctypes as C
    import math
    st = (C.c_double * 2)(-0.5, 0.0)
    rew, dn = C.c_double(), C.c_int()
    pos, vel = -0.5, 0.0
    hit_wall = reached = False
    for t in range(600):
        a = np.float32(-1.5 if t < 80 else (1.25 if vel >= 0 else -1.25))      # into the left wall first, then swing up; |a| > 1 is clipped
        orc.lib().orc_cmc_step(st, (C.c_float * 1)(a), C.byref(rew), C.byref(dn))
        force = min(max(float(a), -1.0), 1.0)
        vel += force * 0.0015 - 0.0025 * math.cos(3 * pos)
        vel = min(max(vel, -0.07), 0.07)
        pos += vel
        pos = min(max(pos, -1.2), 0.6)
        if pos == -1.2 and vel < 0:
            vel = 0.0
            hit_wall = True
        done = pos >= 0.45 and vel >= 0
        r = (100.0 if done else 0) - math.pow(float(a), 2) * 0.1
        assert abs(st[0] - pos) <= 1e-13 and abs(st[1] - vel) <= 1e-14 and abs(rew.value - r) <= 1e-12 and dn.value == int(done)
        pos, vel = st[0], st[1]
        if done:
            reached = True
            assert rew.value > 99.0
            break
    assert hit_wall and reached


def test_pendulum_step_physics():
    """gym==0.17.3 classic_control/pendulum.py (third party, restated): the oracle's step against a line-by-line float64
    restatement -- swing with saturated torques (clipped to +-2), the speed clip at 8, angle_normalize across several turns."""
    import ctypes as C
    import math
    st = (C.c_double * 2)(3.0, 0.5)
    rew = C.c_double()
    th, thdot = 3.0, 0.5
    clipped_speed = wrapped = False
    for t in range(300):
        a = np.float32((3.0 if thdot >= 0 else -3.0) if t < 200 else -0.7 * (1 + t % 3))     # pump energy in, then small torques
        orc.lib().orc_pendulum_step(st, (C.c_float * 1)(a), C.byref(rew))
        u = np.float32(min(max(a, np.float32(-2.0)), np.float32(2.0)))
        an = ((th + math.pi) % (2 * math.pi)) - math.pi
        costs = an ** 2 + .1 * thdot ** 2 + .001 * float(np.float32(u * u))
        nthdot = thdot + (-3 * 10.0 / 2 * math.sin(th + math.pi) + 3. * float(u)) * .05
        nth = th + nthdot * .05
        if abs(nthdot) > 8:
            clipped_speed = True
        nthdot = min(max(nthdot, -8.0), 8.0)
        wrapped = wrapped or abs(nth) > 2 * math.pi
        assert abs(st[0] - nth) <= 1e-12 and abs(st[1] - nthdot) <= 1e-12 and abs(rew.value + costs) <= 1e-12
        th, thdot = st[0], st[1]                                   # lock-step (orc_sin is a restated sine, not libm's)
    assert clipped_speed and wrapped


def _standin_rollout(g, t):
    """Replay the fixture's episode on the oracle's stand-in env: fp32 states, info vectors and raw fp32 rewards."""
    import ctypes as C
    L = orc.lib()
    pre = "t%d_" % t
    x = (C.c_double * 17)(*g[pre + "reset_state"])
    rew = C.c_double()
    s, s2, info, r = [], [], [], []
    for k in range(g[pre + "actions"].shape[0]):
        a = g[pre + "actions"][k]
        s.append(np.array(list(x)).astype(np.float32))
        L.orc_cheetah_step(x, (C.c_float * 6)(*a), C.byref(rew))
        xa = np.array(list(x))
        ctrl = 0.0
        for v in a:
            ctrl = ctrl + float(v) * float(v)
        s2.append(xa.astype(np.float32))
        info.append(np.array([xa[0], xa[8], xa[8], -0.1 * ctrl]).astype(np.float32))
        r.append(np.float32(rew.value))
    return np.array(s), np.array(s2), np.array(info), np.array(r)


def test_g2f_reward_env_vector_state_all_types(golden):
    """RewardEnv._calc_reward on a vector-state env for all 11 reward types (reward_env.py:29-133), incl. the types that
    feed the real env's info vector to the network (3,4,7,8) and the linear info baselines (101,102)."""
    g = golden("g2f_reward_env_cheetah_info")
    H = int(g["hidden"])
    for t in g["types"]:
        t = int(t)
        pre = "t%d_" % t
        s, s2, info, r = _standin_rollout(g, t)
        assert np.array_equal(s2, g[pre + "next_states"])
        n_par = orc.rn_num_params(t, 17, 4, H, 1)
        if t != 0:
            assert n_par == g[pre + "theta"].size
        shaped = orc.rn_shape_rows(t, 17, 4, H, 1, "prelu", 0.25, float(g["gamma"]), g[pre + "theta"], s, s2, info, r)
        np.testing.assert_allclose(shaped, g[pre + "shaped"], rtol=0, atol=3e-6, err_msg="type %d" % t)
    with pytest.raises(ValueError):
        orc.rn_shape_rows(9, 17, 4, H, 1, "prelu", 0.25, 0.98, np.zeros(4), s, s2, info, r)


def test_g6m_worker_best_multi(golden):
    """calc_best_score with num_grad_evals = 3 (GTN_worker.py:234-254): statistics.mean (exactly rounded) / min of both lists,
    mirrored or not -- best scores and eps signs equal to the reference's."""
    g = golden("g6m_worker_best_multi")
    for gt in ("mean", "minmax"):
        for m in (1, 0):
            best, sign = orc.worker_best_multi(g["score_add"], g["score_sub"], bool(m), gt)
            assert np.array_equal(best, g["best_%s_%d" % (gt, m)]), (gt, m)
            assert np.array_equal(sign, g["sign_%s_%d" % (gt, m)]), (gt, m)
    # G = 1 degenerates to the single-evaluation rule
    b1, s1 = orc.worker_best_multi(g["score_add"][:, :1], g["score_sub"][:, :1], True, "mean")
    b0, s0 = orc.worker_best(g["score_add"][:, 0], g["score_sub"][:, 0], True)
    assert np.array_equal(b1, b0) and np.array_equal(s1, s0)


def test_step_budget_semantics():
    """lenv_ddqn_cfg::step_budget, the env-step stand-in for BaseAgent's wall-clock time-out (base_agent.py:30-47,90-97,
    177-184): no budget == huge budget; a budget of one step stops after the first episode, pads the reward list with that
    episode's value and makes the final test return -1e9 (the reference pads an empty list with -1e9); more budget never
    runs fewer episodes."""
    from learning_environments_amd import configs
    cfgd = configs.fixed_work(configs.cartpole_syn_env_ddqn(2), 5)
    cfgd["envs"]["CartPole-v0"]["max_steps"] = 14
    cfgd["agents"]["ddqn"].update(test_episodes=4, batch_size=24, hidden_size=20)
    rng = np.random.RandomState(3)
    P_se = sum(orc.mlp_num_params(d) for d in orc.se_descs(4, 2, 83, 1, "leakyrelu"))
    theta = (rng.randn(P_se) * 0.15).astype(np.float32)
    theta[-1] = -10.0
    init = rng.uniform(-0.4, 0.4, orc.mlp_num_params(orc.mlp_desc(4, 20, 1, 2, "tanh"))).astype(np.float32)

    def run(budget):
        cfgd["agents"]["ddqn"]["step_budget"] = budget
        return orc.ddqn_se_chain(orc.ddqn_cfg_from_config(cfgd, grad_chunk=2, rng_mode=0), theta, init, rng_key=99)

    free, huge, one = run(0), run(10 ** 9), run(1)
    for k in ("score", "episodes_run", "train_steps", "learn_steps", "test_steps"):
        assert free[k] == huge[k]
    assert np.array_equal(free["episode_test_mean"], huge["episode_test_mean"])
    assert one["episodes_run"] == 1 and one["score"] == -1e9 and np.all(one["final_test_returns"] == -1e9)
    assert np.all(one["episode_test_mean"] == one["episode_test_mean"][0])
    assert one["episode_test_mean"][0] == free["episode_test_mean"][0]
    assert np.all(one["episode_len"] == one["episode_len"][0])
    runs = [run(b)["episodes_run"] for b in (1, 30, 60, 90, 120, 200, 400)]
    assert runs == sorted(runs) and runs[-1] == 5
    # a budget that ends inside the final test: returns padded with the minimum of the episodes that did run
    total = free["train_steps"] + free["test_steps"]
    cut = run(total - 20)
    assert cut["episodes_run"] == 5 and cut["test_steps"] < free["test_steps"]
    r = cut["final_test_returns"]
    k = int(np.argmax(r != free["final_test_returns"])) if np.any(r != free["final_test_returns"]) else len(r)
    assert 0 < k < len(r) and np.all(r[k:] == r[:k].min()) and np.array_equal(r[:k], free["final_test_returns"][:k])


def test_nes_draw_statistics():
    """The counter-RNG noise of a generation (CPU twin of lenv_nes_draw): N(0,1)*noise_std moments, independent rows, agent
    inits inside the nn.Linear default bounds."""
    pop, P = 64, 2247
    bounds = np.full(401, 0.5, np.float32)
    eps, init, keys = orc.nes_draw(1234, 0, pop, P, 0.0124, 3 * pop, 3, 0, bounds)
    z = eps.astype(np.float64) / 0.0124
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.1
    c = np.corrcoef(z[:8])
    assert np.max(np.abs(c - np.eye(8))) < 0.1
    assert init.shape == (192, 401) and np.all(np.abs(init) <= 0.5) and abs(init.mean()) < 0.01
    assert abs(init.std() - 0.5 / np.sqrt(3.0)) < 0.01 and len(set(keys.tolist())) == 192
    e2, _, _ = orc.nes_draw(1234, 1, pop, P, 0.0124, 0, 3, 0, None)
    assert not np.array_equal(e2, eps)
