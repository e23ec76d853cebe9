"""GPU tests of the drop-in Python surface (EnvFactory / EnvWrapper / VirtualEnv / RewardEnv / GTN_Master / GTN_Worker):
the same calls a user of the reference makes, checked against the reference's golden vectors and the oracle."""
import json
import math
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _load_theta(envw, flat):
    sd = envw.state_dict()
    off, new = 0, {}
    for k, v in sd.items():
        prelu = k.endswith("weight") and (k[:-6] + "bias") not in sd
        if prelu:
            new[k] = v
        else:
            new[k] = torch.from_numpy(flat[off:off + v.numel()].reshape(tuple(v.shape)).copy())
            off += v.numel()
    assert off == flat.size
    envw.load_state_dict(new)


def test_envwrapper_step_virtual_env_matches_reference(golden):
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    from learning_environments_amd.envs.env_factory import EnvFactory
    g = golden("g1_virtual_env_step")
    venv = EnvFactory(cartpole_syn_env_ddqn()).generate_virtual_env()
    _load_theta(venv, g["c00_theta"])
    for i in range(g["c00_state"].shape[0]):
        venv.env.state = torch.from_numpy(g["c00_state"][i].copy())
        ns, r, d = venv.step(torch.tensor([float(g["c00_action"][i])]))
        assert ns.device.type == "cpu" and ns.dtype == torch.float32 and tuple(ns.shape) == (4,) and tuple(r.shape) == (1,)
        np.testing.assert_allclose(ns.numpy(), g["c00_next_state"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(r.numpy(), g["c00_reward"][i:i + 1], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(d.numpy(), g["c00_done"][i:i + 1], rtol=2e-6, atol=2e-6)
    # internal state advanced like the reference's VirtualEnv (virtual_env.py:52)
    ns2, _, _ = venv.step(torch.tensor([1.0]))
    assert tuple(ns2.shape) == (4,)
    # batched states (histogram experiment, env_wrapper.py:33-40)
    ns, r, d = venv.step(torch.from_numpy(g["c00_action"].astype(np.float32)), state=torch.from_numpy(g["c00_state"].copy()))
    np.testing.assert_allclose(ns.numpy(), g["c00_next_state"], rtol=2e-6, atol=2e-6)
    assert tuple(r.shape) == (12, 1)
    # reset draws a real-env reset state
    s0 = venv.reset()
    assert tuple(s0.shape) == (4,) and float(s0.abs().max()) <= 0.05


def _load_linear_theta(envw, flat):
    """theta (nn.Linear parameters only, state-dict order) into the mirror's modules; PReLU slopes and LayerNorm affines stay."""
    sd = envw.state_dict()
    off, new = 0, {}
    for k, v in sd.items():
        if k.endswith("weight") and v.dim() == 2:
            new[k] = torch.from_numpy(flat[off:off + v.numel()].reshape(tuple(v.shape)).copy()); off += v.numel()
            b = k[:-6] + "bias"
            new[b] = torch.from_numpy(flat[off:off + sd[b].numel()].copy()); off += sd[b].numel()
    for k, v in sd.items():
        new.setdefault(k, v)
    assert off == flat.size
    envw.load_state_dict(new)


@pytest.mark.parametrize("name", ["g8ts_calc_score_cheetah_td3_virtual_env", "g8p_calc_score_pendulum_td3_virtual_env",
                                  "g8tseln_calc_score_cheetah_td3_virtual_env_layernorm"])
def test_envwrapper_step_continuous_action_virtual_env_matches_reference(golden, name):
    """EnvWrapper.step -> VirtualEnv.step of an SE over a CONTINUOUS action space (virtual_env.py:43-54: input = cat(action, state), the
    action vector as it comes): every SE transition of the reference's TD3 runs (HalfCheetah stand-in 23-20-20-x, Pendulum 4-20-20-x, and
    the HalfCheetah SE with `use_layer_norm`) replayed through the host mirror."""
    from learning_environments_amd.envs.env_factory import EnvFactory
    g = golden(name)
    cfg = json.loads(str(g["config_json"]))
    cfg["device"] = "cuda"
    venv = EnvFactory(cfg).generate_virtual_env()
    assert not venv.has_discrete_action_space()
    _load_linear_theta(venv, g["theta"])
    n = min(40, g["tr_reward"].size)
    for k in range(n):
        ns, r, d = venv.step(torch.from_numpy(g["tr_action"][k].copy()), state=torch.from_numpy(g["tr_state"][k].copy()))
        assert tuple(ns.shape) == g["tr_next_state"][k].shape and ns.device.type == "cpu"
        np.testing.assert_allclose(ns.numpy(), g["tr_next_state"][k], rtol=0, atol=3e-6)
        assert abs(float(r) - float(g["tr_reward"][k])) <= 3e-6
    # rows at once
    ns, r, d = venv.env.step(torch.from_numpy(g["tr_action"][:n].copy()), state=torch.from_numpy(g["tr_state"][:n].copy()))
    np.testing.assert_allclose(ns.cpu().numpy(), g["tr_next_state"][:n], rtol=0, atol=3e-6)
    assert tuple(r.shape) == (n, 1) and tuple(d.shape) == (n, 1)


def test_host_mirrors_with_layer_norm_env_nets():
    """`use_layer_norm: True` in the env's section with two hidden layers: the one-step entries behind VirtualEnv.step,
    EnvWrapper.step_population and RewardEnv.step read the LayerNorm's weight | bias behind each net's second Linear (lenv_mlp_desc layout)
    while theta / eps stay the nn.Linear parameters.  Checked against the mirror's own torch modules (built by build_nn_from_config)."""
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, pendulum_reward_env_td3
    from learning_environments_amd.envs.env_factory import EnvFactory
    from learning_environments_amd.models.model_utils import linear_params
    torch.manual_seed(3)
    cfg = cartpole_syn_env_ddqn()
    cfg["envs"]["CartPole-v0"].update(hidden_layer=2, hidden_size=24, use_layer_norm=True, activation_fn="tanh")
    venv = EnvFactory(cfg).generate_virtual_env()
    nets = (venv.env.state_net, venv.env.reward_net, venv.env.done_net)
    assert all(sum(isinstance(m, torch.nn.LayerNorm) for m in net) == 1 for net in nets)
    theta = venv.env.flat_params()
    assert theta.numel() == sum(p.numel() for p in linear_params(venv.env)) == 3 * (6 * 24 + 24 + 24 * 24 + 24) + 6 * 25
    with torch.no_grad():                                   # an affine that is not the constructor's: both vectors must be read
        for net in nets:
            ln = [m for m in net if isinstance(m, torch.nn.LayerNorm)][0]
            ln.weight.add_(0.2 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    states = torch.randn(9, 4) * 0.3
    actions = torch.randint(0, 2, (9,))
    onehot = torch.nn.functional.one_hot(actions, 2).float()
    ns, r, d = venv.step(actions.float(), state=states)
    with torch.no_grad():
        x = torch.cat([onehot, states], dim=1).to(theta.device)
        want = [net(x).cpu() for net in nets]
    np.testing.assert_allclose(ns.numpy(), want[0].numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r.numpy(), want[1].numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(d.numpy(), want[2].numpy(), rtol=2e-5, atol=2e-6)
    # a population of perturbed SEs: eps in theta's (Linear-only) layout
    eps = (0.05 * torch.randn(2, theta.numel())).to(theta.device)
    worker = torch.tensor([0, 1, 1], dtype=torch.int32, device=theta.device)
    sign = torch.tensor([1.0, -1.0, 0.0], device=theta.device)
    pns, pr, pd = venv.step_population(actions[:3].to(torch.int32).to(theta.device), states[:3].to(theta.device).contiguous(), eps, worker, sign)
    saved = theta.clone()
    for c in range(3):
        with torch.no_grad():
            theta.copy_(saved + sign[c] * eps[worker[c]])
            w = [net(x[c:c + 1]).cpu() for net in nets]
        np.testing.assert_allclose(pns[c].cpu().numpy().reshape(-1), w[0].numpy().reshape(-1), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(float(pr[c].reshape(-1)[0]), float(w[1]), rtol=2e-5, atol=2e-6)
    with torch.no_grad():
        theta.copy_(saved)
    # RewardEnv over Pendulum: the shipped two-hidden-layer reward net (default_config_pendulum_reward_env.yaml), here with the LayerNorm
    rcfg = pendulum_reward_env_td3()
    rcfg["device"] = "cuda"
    rcfg["envs"]["Pendulum-v0"].update(hidden_size=20, use_layer_norm=True, reward_env_type=2)
    assert int(rcfg["envs"]["Pendulum-v0"]["hidden_layer"]) == 2
    renv = EnvFactory(rcfg).generate_reward_env()
    renv.set_agent_params(same_action_num=1, gamma=0.97)
    rn = renv.env.reward_net
    with torch.no_grad():
        ln = [m for m in rn if isinstance(m, torch.nn.LayerNorm)][0]
        ln.weight.add_(0.2 * torch.randn_like(ln.weight)); ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    renv.reset()
    for k in range(5):
        s_before = np.asarray(renv.env.state, np.float32).copy()
        ns, rr, dd = renv.step(torch.tensor([0.7 * (-1) ** k]))
        with torch.no_grad():
            dev_ = next(rn.parameters()).device
            phi_s = float(rn(torch.from_numpy(s_before).to(dev_)))
            phi_s2 = float(rn(ns.to(dev_).float().reshape(-1)))
        # type 2: r + gamma * phi(s') - phi(s); the real reward is recovered from the shaped one
        r_real = float(rr) - (0.97 * phi_s2 - phi_s)
        th, thdot = np.arctan2(s_before[1], s_before[0]), s_before[2]
        u = float(np.clip(0.7 * (-1) ** k, -2, 2))
        assert abs(r_real - (-(th ** 2 + 0.1 * thdot ** 2 + 0.001 * u ** 2))) <= 2e-4, (k, r_real)


def test_real_env_step_matches_oracle():
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    from learning_environments_amd.envs.env_factory import EnvFactory
    from oracle import oracle as orc
    import ctypes as C
    real = EnvFactory(cartpole_syn_env_ddqn()).generate_real_env()
    s = real.reset().numpy().astype(np.float64)
    st = (C.c_double * 4)(*real.env._alloc()["state"].cpu().tolist())
    rew, dn = C.c_double(), C.c_int()
    total = 0
    for t in range(200):
        a = t % 2
        ns, r, d = real.step(torch.tensor([float(a)]))
        orc.lib().orc_cartpole_step(st, a, C.byref(rew), C.byref(dn))
        assert np.array_equal(ns.numpy(), np.array(list(st), np.float64).astype(np.float32))
        assert float(r) == rew.value
        total += 1
        if float(d) > 0.5:
            assert dn.value == 1 or total == 200
            break
    assert total < 200


def test_real_env_mountaincar_step_matches_oracle():
    """MountainCar-v0 through EnvWrapper.reset/step on the device against the oracle's step: swing to the flag (bang-bang on the
    velocity sign), bit-exact observations, -1 rewards, done at the flag."""
    from learning_environments_amd.configs import mountaincar_syn_env_ddqn
    from learning_environments_amd.envs.env_factory import EnvFactory
    from oracle import oracle as orc
    import ctypes as C
    real = EnvFactory(mountaincar_syn_env_ddqn()).generate_real_env()
    assert real.get_state_dim() == 2 and real.get_action_dim() == 3
    s = real.reset()
    assert tuple(s.shape) == (2,) and -0.6 <= float(s[0]) <= -0.4 and float(s[1]) == 0.0
    st = (C.c_double * 4)(*real.env._alloc()["state"].cpu().tolist())
    rew, dn = C.c_double(), C.c_int()
    for t in range(200):
        a = 2 if st[1] >= 0 else 0
        ns, r, d = real.step(torch.tensor([float(a)]))
        orc.lib().orc_mountaincar_step(st, a, C.byref(rew), C.byref(dn))
        assert np.array_equal(ns.numpy(), np.array([st[0], st[1]], np.float64).astype(np.float32))
        assert float(r) == rew.value == -1.0
        if float(d) > 0.5:
            assert dn.value == 1 and st[0] >= 0.5
            break
    assert 60 < t < 199


def test_real_env_pendulum_step_matches_oracle():
    """Pendulum-v0 through EnvWrapper.reset/step on the device against the oracle's step: pumped up to the speed clip, bit-exact
    observations and rewards, TimeLimit after 200 steps, max_action 2 and a Box(-2, 2) action space."""
    from learning_environments_amd.configs import pendulum_syn_env_td3
    from learning_environments_amd.envs.env_factory import EnvFactory
    from oracle import oracle as orc
    import ctypes as C
    real = EnvFactory(pendulum_syn_env_td3()).generate_real_env()
    assert (real.get_state_dim(), real.get_action_dim(), real.get_max_action()) == (3, 1, 2)
    assert float(real.env.action_space.high[0]) == 2.0 and not real.has_discrete_action_space()
    s = real.reset()
    st = (C.c_double * 2)(*real.env._alloc()["state"].cpu().tolist())
    assert -math.pi <= st[0] <= math.pi and -1 <= st[1] <= 1
    assert abs(float(s[0]) - math.cos(st[0])) <= 1e-6 and float(s[2]) == np.float32(st[1])
    rew = C.c_double()
    for t in range(200):
        a = np.float32(2.5 if st[1] >= 0 else -2.5)
        ns, r, d = real.step(torch.tensor([float(a)]))
        orc.lib().orc_pendulum_step(st, (C.c_float * 1)(a), C.byref(rew))
        dev_state = real.env._alloc()["state"].cpu().tolist()
        assert dev_state == [st[0], st[1]]
        assert abs(float(ns[0]) - math.cos(st[0])) <= 1e-6 and abs(float(ns[1]) - math.sin(st[0])) <= 1e-6 and float(ns[2]) == np.float32(st[1])
        assert float(r) == np.float32(rew.value)
        assert float(d) == (1.0 if t == 199 else 0.0)
    assert abs(st[1]) == 8.0 or abs(st[0]) > 2 * math.pi


def test_reward_env_step_matches_reference(golden):
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.env_factory import EnvFactory
    g = golden("g9_calc_score_cliff_a")
    renv = EnvFactory(cliff_reward_env_ql()).generate_reward_env()
    _load_theta(renv, g["theta"])
    renv.set_agent_params(same_action_num=1, gamma=0.8)
    table = renv.env.shaped_table().numpy()
    np.testing.assert_allclose(table, g["shaped_ref"], rtol=2e-6, atol=2e-6)
    # replay the reference's first training episode through EnvWrapper.step
    renv.reset()
    for k in range(int(g["episode_length_train"][0])):
        ns, r, d = renv.step(torch.tensor([float(g["tr_action"][k])]))
        assert int(ns.item()) == int(g["tr_next_state"][k]) and float(d) == float(g["tr_done"][k])
        assert abs(float(r) - float(g["tr_reward"][k])) <= 2e-6


def _master_pair(cfg, tmp_path, monkeypatch, hip_only=False):
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    hip = GTN_Master(cfg, bohb_id=0, seed=5)
    return hip


def test_gtn_master_run_ddqn_se_matches_oracle_engine(tmp_path, monkeypatch):
    """Full NES generations through GTN_Master.run() on the GPU; theta after the update and every fitness equal a
    CPU-oracle evaluation of the same population (same noise, same keys)."""
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    from oracle import oracle as orc
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=3, max_iterations=2), 2)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 15
    cfg["agents"]["ddqn"]["test_episodes"] = 3
    m = _master_pair(cfg, tmp_path, monkeypatch)
    theta0 = m.theta.cpu().numpy().copy()
    mean_score, mean_list, model_name = m.run()
    assert len(mean_list) == 2 and isinstance(model_name, str) and len(m.score_list) == 3
    # re-evaluate the LAST generation with the oracle from the theta it started from
    it = 1
    m2 = _master_pair(cfg, tmp_path, monkeypatch)
    m2.step(0)
    theta1 = m2.theta.cpu().numpy().copy()
    gathered = m2.evaluate_population(it).cpu().numpy()
    eps = m2.eps.cpu().numpy()
    # the generation's inputs come from lenv_nes_draw; its CPU twin reproduces them bit for bit
    oeps, init, okeys = orc.nes_draw(m2.seed, it, 3, m2.p_theta, cfg["agents"]["gtn"]["noise_std"], 9, 3, 0, m2.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=m2.cfg.grad_chunk)
    scores = orc.ddqn_se_population(ocfg, theta1, eps, init, seed=m2.seed, generation=it, threads=4)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    assert np.array_equal(gathered[:, 2], sign.astype(np.float64))
    m2._transform_and_update(torch.from_numpy(gathered).cuda())
    w = orc.score_transform(3, gathered[:, 0], gathered[:, 1])
    assert np.array_equal(m2.theta.cpu().numpy(), orc.update_env(theta1, eps, sign, w, cfg["agents"]["gtn"]["step_size"]))
    assert np.array_equal(m2.theta.cpu().numpy(), m.theta.cpu().numpy())      # run() == step(0); step(1)
    assert not np.array_equal(theta0, theta1)
    # the module parameters alias the flat theta: state_dict() shows the updated weights (checkpoint contract)
    sd = m.synthetic_env_orig.state_dict()
    assert np.array_equal(sd["env.state_net.0.weight"].cpu().numpy().reshape(-1), m.theta.cpu().numpy()[:83 * 6])


def test_gtn_master_run_ql_cliff(tmp_path, monkeypatch):
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.gridworld import transition_tables
    from oracle import oracle as orc
    cfg = cliff_reward_env_ql(num_workers=6, max_iterations=2)
    cfg["agents"]["gtn"]["quit_when_solved"] = False
    m = _master_pair(cfg, tmp_path, monkeypatch)
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    tables = transition_tables("Cliff")
    ocfg = orc.ql_cfg_from_config(cfg, tables)
    for p in range(6):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.ql_rn_chain(ocfg, w, tables, rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])
        assert gathered[p, 2] == (-1.0 if sc[2] > sc[1] else 1.0)
    # the shaped-reward table of the wrapper must follow theta although the kernels update it through a raw
    # pointer (no torch version bump): one env step before and after an update sees different shaped rewards
    renv = m.synthetic_env_orig
    renv.set_agent_params(same_action_num=1, gamma=cfg["agents"]["ql"]["gamma"])
    table_before = renv.env.shaped_table().clone()
    m._gathered = torch.from_numpy(gathered).cuda()
    m.score_list, m.score_orig_list = gathered[:, 0].tolist(), gathered[:, 1].tolist()
    m.update_env()
    table_after = renv.env.shaped_table()
    assert not torch.equal(table_before, table_after)
    _, want = __import__("learning_environments_amd.engine", fromlist=["x"]).rn_shape_population(
        renv.env.ql_cfg(), m.theta, None, None, None,
        torch.from_numpy(tables["next_state"]).contiguous().cuda(), torch.from_numpy(tables["reward"]).contiguous().cuda(), 1)
    assert torch.equal(table_after, want[0].cpu())
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 2 and -102.0 <= mean_score <= -12.0
    # RN models are saved whenever the mean improves (GTN_master.py:127-129)
    saved = torch.load(os.path.join(m.model_dir, os.path.basename(m.model_name)), weights_only=False)
    assert set(saved.keys()) == {"model", "config"} and "env.reward_net.1.weight" in saved["model"]


def test_gtn_worker_file_protocol(tmp_path, monkeypatch):
    """A GTN_Worker of this package served through the reference's sync-file protocol (GTN_base.py:19-29)."""
    from learning_environments_amd.agents.GTN import GTN_Worker
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    from learning_environments_amd.envs.env_factory import EnvFactory
    monkeypatch.chdir(tmp_path)
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=1, max_iterations=1), 2)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 10
    cfg["agents"]["gtn"]["time_sleep_worker"] = 0.01
    w = GTN_Worker(id=0, bohb_id=7, seed=3)
    venv = EnvFactory(cfg).generate_virtual_env()
    data = {"timeout": 600.0, "quit_flag": True, "config": cfg,
            "synthetic_env_orig": {k: v.cpu() for k, v in venv.state_dict().items()}}
    torch.save(data, w.get_input_file_name(0))
    torch.save({}, w.get_input_check_file_name(0))
    w.run()
    res = torch.load(w.get_result_file_name(0), weights_only=False)
    assert os.path.isfile(w.get_result_check_file_name(0)) and not os.path.isfile(w.get_input_file_name(0))
    assert set(res.keys()) == {"eps", "synthetic_env", "time_elapsed", "score", "score_orig"}
    assert set(res["eps"].keys()) == set(data["synthetic_env_orig"].keys())
    assert 1.0 <= res["score"] <= 10.0 and 1.0 <= res["score_orig"] <= 10.0
    # eps has the configured scale and the returned env is theta +/- eps (mirrored sampling)
    e = res["eps"]["env.state_net.0.weight"]
    assert 0.5 * cfg["agents"]["gtn"]["noise_std"] < float(e.std()) < 2 * cfg["agents"]["gtn"]["noise_std"]
    d = res["synthetic_env"]["env.state_net.0.weight"] - data["synthetic_env_orig"]["env.state_net.0.weight"]
    assert torch.allclose(d, e, atol=1e-6)


def test_gtn_master_acrobot_dueling_generation(tmp_path, monkeypatch):
    """BASELINE config 3 shapes (Acrobot SE 9-128-{6,1,1}, DuelingDDQN 6-128-128-128 / 128-128-{1,3}, B=128) through
    GTN_Master on a tiny step budget; the fitness triples equal an oracle evaluation of the same population."""
    from learning_environments_amd.agents.nes_common import fresh_agent_init
    from learning_environments_amd.configs import acrobot_syn_env_duelingddqn, fixed_work
    from oracle import oracle as orc
    cfg = fixed_work(acrobot_syn_env_duelingddqn(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["Acrobot-v1"]["max_steps"] = 10
    cfg["agents"]["duelingddqn"].update(init_episodes=1, test_episodes=2)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.cfg.agent_kind == 1 and m.inner.p_agent == 67460 and m.p_theta == 4872
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0)
    scores = orc.ddqn_se_population(ocfg, theta0, eps, init, seed=m.seed, generation=0, threads=6)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    assert np.array_equal(gathered[:, 2], sign.astype(np.float64))
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and -10.0 <= mean_score <= 0.0


def test_gtn_master_acrobot_ddqn_two_layer_generation(tmp_path, monkeypatch):
    """default_config_acrobot.yaml's DDQN (Critic_DQN 6-128-128-3, B 128) through GTN_Master: the config builder keeps
    grad_chunk 0, InnerLoop routes to the GEMM-tiled kernel, and the fitness triples equal an oracle evaluation."""
    from learning_environments_amd.configs import acrobot_syn_env_ddqn, fixed_work
    from oracle import oracle as orc
    cfg = fixed_work(acrobot_syn_env_ddqn(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["Acrobot-v1"]["max_steps"] = 10
    cfg["agents"]["ddqn"].update(test_episodes=2)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.cfg.agent_kind == 0 and m.cfg.grad_chunk == 0 and m.inner.dueling and m.inner.p_agent == 17795
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0)
    scores = orc.ddqn_se_population(ocfg, theta0, eps, init, seed=m.seed, generation=0, threads=6)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    assert np.array_equal(gathered[:, 2], sign.astype(np.float64))


def test_gtn_master_ddqn_layer_norm_generation(tmp_path, monkeypatch):
    """`use_layer_norm: True` in the ddqn section through GTN_Master: the config builder sets cfg.q_layer_norm, the fresh agents carry the
    shared LayerNorm's block (weight 1, bias 0) behind the second Linear, the launch trains through it, and the fitness triples equal an
    oracle evaluation of the same agents."""
    from learning_environments_amd.configs import acrobot_syn_env_ddqn, fixed_work
    from oracle import oracle as orc
    cfg = fixed_work(acrobot_syn_env_ddqn(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["Acrobot-v1"]["max_steps"] = 10
    cfg["agents"]["ddqn"].update(test_episodes=2, hidden_size=48, batch_size=8, use_layer_norm=True)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    H = 48
    P = (6 * H + H) + (H * H + H) + 2 * H + (3 * H + 3)
    assert m.cfg.q_layer_norm == 1 and m.inner.dueling and m.inner.p_agent == P and m.agent_bounds.numel() == P
    off = (6 * H + H) + (H * H + H)
    assert m.task.ln_slice == (off, H) and float(m.agent_bounds[off:off + 2 * H].abs().max()) == 0.0
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps) and np.all(init[:, off:off + 2 * H] == 0.0)
    init[:, off:off + H] = 1.0                               # nn.LayerNorm: weight 1, bias 0 (tasks.set_layer_norm_init)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0)
    assert ocfg.q_layer_norm == 1
    scores = orc.ddqn_se_population(ocfg, theta0, eps, init, seed=m.seed, generation=0, threads=6)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    assert np.array_equal(gathered[:, 2], sign.astype(np.float64))
    # the captured generation (the LayerNorm block's fill is one more node of the graph) against the eager one: same theta
    from learning_environments_amd.agents.GTN import GTN_Master
    thetas = []
    for graph in (True, False):
        torch.manual_seed(0)
        mm = GTN_Master(cfg, bohb_id=0, seed=5, graph=graph)
        mm.step(0); mm.step(1)
        torch.cuda.synchronize()
        thetas.append(mm.theta.cpu().numpy().copy())
    assert np.array_equal(thetas[0], thetas[1])


def test_gtn_master_ddqn_vary_generation(tmp_path, monkeypatch):
    """`agent_name: DDQN_vary` (what default_config_acrobot.yaml:26 ships) through GTN_Master: every chain draws its own
    hyper-parameters, the population runs as one launch, the fitness records equal an oracle evaluation chain by chain;
    vary_hp False is the base agent (DDQN_vary.py:16-21)."""
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.config import agent_layer_dims
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work, with_vary
    from learning_environments_amd import _lib
    from oracle import oracle as orc
    base = fixed_work(cartpole_syn_env_ddqn(num_workers=2, max_iterations=1), 2)
    base["envs"]["CartPole-v0"]["max_steps"] = 10
    base["agents"]["ddqn"].update(test_episodes=2, init_episodes=1)
    cfg = with_vary(base)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert isinstance(m.task, tasks.DdqnVaryTask) and m.inner.vary and m.cfg.batch_size == 597 and m.cfg.q_hidden == 171
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    hps = m.task.last_hp
    assert len(hps) == 6 and len({h["batch_size"] for h in hps}) > 1
    scores = []
    for c in range(6):
        key = orc.chain_key(m.seed, 0, c // 3, c % 3)
        oh = orc.vary_chain_hp(cfg["agents"]["ddqn"], key)       # numpy's exp/log vs libm's: lr may differ in the last bit
        assert all(hps[c][n] == oh[n] for n in ("batch_size", "hidden_size", "hidden_layer")) and abs(hps[c]["lr"] / oh["lr"] - 1) < 1e-14
        ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0, **orc.hp_overrides(hps[c]))
        pc = _lib.DdqnCfg()
        for f, _ in _lib.DdqnCfg._fields_:
            setattr(pc, f, getattr(ocfg, f, 0))
        init = orc.agent_init_from_key(key, agent_layer_dims(pc))
        sg = np.float32([0.0, 1.0, -1.0][c % 3])
        w = (sg * eps[c // 3] + theta0).astype(np.float32)
        scores.append(orc.ddqn_se_chain(ocfg, w, init, rng_key=key)["score"])
    scores = np.array(scores)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    assert np.array_equal(gathered[:, 2], sign.astype(np.float64))
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1
    # vary_hp False: the plain DDQN task, same kernel and numbers as agent_name DDQN
    m0 = _master_pair(with_vary(base, vary_hp=False), tmp_path, monkeypatch)
    m1 = _master_pair(base, tmp_path, monkeypatch)
    assert isinstance(m0.task, tasks.DdqnSeTask) and not m0.inner.vary
    assert torch.equal(m0.evaluate_population(0), m1.evaluate_population(0))


def test_reward_env_cheetah_standin_step_matches_reference(golden):
    """EnvWrapper.step on the continuous-state RewardEnv: next states bit-equal to the shim run of the reference, shaped
    rewards within the fixture tolerance."""
    from learning_environments_amd.configs import halfcheetah_reward_env_td3
    from learning_environments_amd.envs.env_factory import EnvFactory
    g = golden("g8t_calc_score_cheetah_td3")
    cfg = json.loads(str(g["config_json"]))
    cfg["device"] = "cuda"
    renv = EnvFactory(cfg).generate_reward_env()
    assert list(renv.state_dict().keys()) == ['env.reward_net.0.weight', 'env.reward_net.0.bias', 'env.reward_net.1.weight',
                                              'env.reward_net.2.weight', 'env.reward_net.2.bias']
    assert renv.get_state_dim() == 17 and renv.get_action_dim() == 6 and not renv.has_discrete_action_space()
    assert renv.get_max_action() == 1 and renv.get_min_action() == -1
    a = renv.get_random_action()
    assert tuple(a.shape) == (6,) and a.dtype == torch.float32 and float(a.abs().max()) <= 1.0
    _load_theta(renv, g["theta"])
    renv.set_agent_params(same_action_num=1, gamma=0.98)
    renv.reset()
    # put the device env into the reference's first reset state and replay its first training episode
    st = renv.env.real_env._alloc()["state"]
    st.copy_(torch.from_numpy(g["tape_train_reset"][0]).to(st.device))
    renv.env.state = g["tape_train_reset"][0].copy()
    for k in range(7):
        ns, r, d = renv.step(torch.from_numpy(g["tr_action"][k].copy()))
        assert np.array_equal(ns.numpy(), g["tr_next_state"][k])
        assert abs(float(r) - float(g["tr_reward"][k])) <= 5e-5
        assert float(d) == (1.0 if k == 6 else 0.0)


def test_mlp_forward_entry_matches_oracle(golden):
    from learning_environments_amd import engine
    from oracle import oracle as orc
    g = golden("g3_critic_dqn_forward")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        S, A, H, L, act = [int(v) for v in g[pre + "meta"]]
        y = engine.mlp_forward(engine.mlp_desc(S, H, L, A, act), torch.from_numpy(g[pre + "params"]).cuda(), torch.from_numpy(g[pre + "x"]).cuda())
        assert np.array_equal(y.cpu().numpy(), orc.mlp_forward(orc.mlp_desc(S, H, L, A, act), g[pre + "params"], g[pre + "x"]))
        np.testing.assert_allclose(y.cpu().numpy(), g[pre + "y"], rtol=1e-6, atol=1e-6)


def test_mlp_forward_with_layer_norm_matches_oracle_and_reference(golden):
    """`use_layer_norm` nets (models/model_utils.py:22-37) through lenv_mlp_forward: bit-exact against the oracle, the reference
    module's outputs within 2e-5 (the one-step SE entry with such nets: test_se_step_population_layer_norm)."""
    from learning_environments_amd import engine
    from oracle import oracle as orc
    g = golden("g1ln_mlp_layer_norm")
    acts = ["identity", "relu", "leakyrelu", "tanh", "prelu"]
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        din, dout, H, L, act = [int(v) for v in g[pre + "meta"]]
        d = engine.mlp_desc(din, H, L, dout, acts[act], use_layer_norm=True)
        assert engine.mlp_num_params(d) == g[pre + "params"].size
        y = engine.mlp_forward(d, torch.from_numpy(g[pre + "params"]).cuda(), torch.from_numpy(g[pre + "x"]).cuda()).cpu().numpy()
        assert np.array_equal(y, orc.mlp_forward(orc.mlp_desc(din, H, L, dout, acts[act], use_layer_norm=True), g[pre + "params"], g[pre + "x"]))
        np.testing.assert_allclose(y, g[pre + "y"], rtol=2e-5, atol=2e-6)


def test_gtn_master_td3_cheetah_generation(tmp_path, monkeypatch):
    from learning_environments_amd.agents.nes_common import fresh_agent_init
    from learning_environments_amd.configs import fixed_work, halfcheetah_reward_env_td3
    from oracle import oracle as orc
    cfg = fixed_work(halfcheetah_reward_env_td3(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["HalfCheetah-v3"]["max_steps"] = 5
    cfg["agents"]["td3"].update(init_episodes=1, batch_size=32)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.task.name == "td3_rn" and m.p_theta == 17 * 128 + 128 + 128 + 1 and m.inner.p_agent == 59016
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.td3_cfg_from_config(cfg)
    for p in range(2):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.td3_rn_chain(ocfg, w, init[3 * p + kind], rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and np.isfinite(mean_score)


def test_gtn_master_td3_layer_norm_generation(tmp_path, monkeypatch):
    """`use_layer_norm: True` in the td3 section through GTN_Master: cfg.use_layer_norm, the three nets' LayerNorm blocks (weight 1, bias
    0) in the fresh agents, and fitness values equal to an oracle evaluation of the same agents."""
    from learning_environments_amd.configs import fixed_work, halfcheetah_reward_env_td3
    from oracle import oracle as orc
    cfg = fixed_work(halfcheetah_reward_env_td3(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["HalfCheetah-v3"]["max_steps"] = 5
    cfg["agents"]["td3"].update(init_episodes=1, batch_size=32, hidden_size=40, use_layer_norm=True)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    H = 40
    pa, pc = (17 * H + H) + (H * H + H) + 2 * H + (6 * H + 6), (23 * H + H) + (H * H + H) + 2 * H + (H + 1)
    assert m.task.name == "td3_rn" and m.cfg.use_layer_norm == 1 and m.inner.p_agent == pa + 2 * pc and m.agent_bounds.numel() == pa + 2 * pc
    assert [o for o, _ in m.task.ln_slice] == [17 * H + H + H * H + H, pa + 23 * H + H + H * H + H, pa + pc + 23 * H + H + H * H + H]
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    for off, h in m.task.ln_slice:
        assert np.all(init[:, off:off + 2 * h] == 0.0)
        init[:, off:off + h] = 1.0                           # nn.LayerNorm: weight 1, bias 0 (tasks.set_layer_norm_init)
    ocfg = orc.td3_cfg_from_config(cfg)
    assert ocfg.use_layer_norm == 1
    for p in range(2):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.td3_rn_chain(ocfg, w, init[3 * p + kind], rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])


def test_gtn_master_acrobot_ddqn_vary_as_shipped(tmp_path, monkeypatch):
    """default_config_acrobot.yaml as the reference ships it (`agent_name: DDQN_vary`, ddqn 128 / 2 layers / batch 128): the
    launch is sized for batch 384, width 384, 3 hidden layers; one short generation runs and every chain's draw is inside
    the reference's ranges."""
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.configs import acrobot_syn_env_ddqn, fixed_work, with_vary
    cfg = with_vary(fixed_work(acrobot_syn_env_ddqn(num_workers=3, max_iterations=1), 2))
    cfg["envs"]["Acrobot-v1"]["max_steps"] = 8
    cfg["agents"]["ddqn"].update(test_episodes=2)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert isinstance(m.task, tasks.DdqnVaryTask)
    assert (m.cfg.batch_size, m.cfg.q_hidden, m.cfg.q_layers, m.cfg.agent_kind) == (384, 384, 3, 0) and m.inner.p_agent == 299523
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and -8.0 <= mean_score <= 0.0
    assert m.inner.status.cpu().tolist() == [0] * 9
    for h in m.task.last_hp:
        assert 42 <= h["batch_size"] <= 384 and 42 <= h["hidden_size"] <= 384 and 1 <= h["hidden_layer"] <= 3
        assert 1e-3 / 3 <= h["lr"] <= 3e-3
    st = m.inner.stats.cpu().numpy()
    assert (st[:, 2] > 0).all()                              # every chain learned


def test_gtn_master_ddqn_icm_generation(tmp_path, monkeypatch):
    """`agent_name: DDQN_icm` through GTN_Master: every chain gets a fresh ICM (counter-RNG stream 12), trains it inside
    learn() and sees the intrinsic rewards; fitness records equal an oracle evaluation."""
    from learning_environments_amd import _lib
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work, with_icm
    from oracle import oracle as orc
    cfg = fixed_work(with_icm(cartpole_syn_env_ddqn(num_workers=2, max_iterations=1), feature_dim=16, hidden_size=24), 2)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 10
    cfg["agents"]["ddqn"].update(test_episodes=2, init_episodes=1, batch_size=24)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.cfg.icm_enabled == 1 and m.cfg.grad_chunk == 0 and m.inner.icm and m.inner.p_icm == 7081
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0)
    scores = []
    for c in range(6):
        key = orc.chain_key(m.seed, 0, c // 3, c % 3)
        icm_init = orc.agent_init_from_key(key, orc.icm_layer_dims(ocfg), stream=orc.STREAM_ICM_INIT)
        w = (np.float32([0.0, 1.0, -1.0][c % 3]) * eps[c // 3] + theta0).astype(np.float32)
        scores.append(orc.ddqn_se_chain(ocfg, w, init[c], rng_key=key, icm_init=icm_init)["score"])
    scores = np.array(scores)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1


def test_gtn_master_cartpole_reward_env_ddqn(tmp_path, monkeypatch):
    """default_config_cartpole_reward_env.yaml's experiment through GTN_Master: theta is the reward network (RewardEnv wrapper,
    state-dict keys of the reference), the agents train on the real CartPole with shaped rewards inside the fused kernel; the
    fitness records equal an oracle evaluation, and a short NES run updates theta."""
    from learning_environments_amd.configs import cartpole_reward_env_ddqn, fixed_work
    from oracle import oracle as orc
    cfg = fixed_work(cartpole_reward_env_ddqn(num_workers=3, max_iterations=2), 3)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 25
    cfg["agents"]["ddqn"].update(batch_size=32, test_episodes=2)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    # (round 6: the register-resident kernel's RENV instantiation runs this configuration -- the cfg carries config.pick_grad_chunk's micro-chunk)
    assert m.cfg.synthetic_env_type == 1 and m.cfg.reward_env_type == 2 and not m.inner.dueling and m.cfg.grad_chunk > 0
    assert m.p_theta == 4 * 64 + 64 + 64 + 1
    assert "env.reward_net.0.weight" in m.synthetic_env_orig.state_dict() and not m.synthetic_env_orig.is_virtual_env()
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 3, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 9, 3, 0, m.agent_bounds.cpu().numpy())
    assert np.array_equal(eps, oeps)
    ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=m.cfg.grad_chunk)
    scores = []
    for c in range(9):
        w = (np.float32([0.0, 1.0, -1.0][c % 3]) * eps[c // 3] + theta0).astype(np.float32)
        scores.append(orc.ddqn_se_chain(ocfg, w, init[c], rng_key=orc.chain_key(m.seed, 0, c // 3, c % 3))["score"])
    scores = np.array(scores)
    best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
    assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 2 and not np.array_equal(m.theta.cpu().numpy(), theta0)


def test_gtn_master_halfcheetah_virtual_env_td3_vary_as_shipped(tmp_path, monkeypatch):
    """default_config_halfcheetah.yaml as shipped (agent td3_vary, synthetic_env_type 0, SE 23-128-128-128-{17,1,1}, TD3 batch 256 /
    128 x 2) on the HalfCheetah stand-in: theta is the VirtualEnv (state-dict keys of the reference), the launch is sized for
    batch 768 / width 384 / 3 layers, a short generation runs, and the plain-td3 form of the same config equals the oracle."""
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.configs import fixed_work, halfcheetah_syn_env_td3, with_vary
    from oracle import oracle as orc
    base = fixed_work(halfcheetah_syn_env_td3(num_workers=2, max_iterations=1), 2)
    base["envs"]["HalfCheetah-v3"]["max_steps"] = 6
    base["agents"]["td3"].update(init_episodes=1, test_episodes=1)
    m = _master_pair(with_vary(base), tmp_path, monkeypatch)
    assert isinstance(m.task, tasks.Td3VaryTask) and m.cfg.virtual_env == 1
    assert (m.cfg.batch_size, m.cfg.hidden, m.cfg.layers, m.cfg.rn_layers) == (768, 384, 3, 3)
    assert m.synthetic_env_orig.is_virtual_env() and "env.state_net.0.weight" in m.synthetic_env_orig.state_dict()
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and np.isfinite(mean_score) and m.inner.status.cpu().tolist() == [0] * 6
    # plain td3 on the same VirtualEnv: oracle-equal fitness
    base["agents"]["td3"].update(batch_size=32)
    m = _master_pair(base, tmp_path, monkeypatch)
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, base["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
    ocfg = orc.td3_cfg_from_config(base)
    for p in range(2):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.td3_rn_chain(ocfg, w, init[3 * p + kind], rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])


def test_gtn_master_pendulum_configs_as_shipped(tmp_path, monkeypatch):
    """default_config_pendulum.yaml (td3_vary on a VirtualEnv 4-32-32-{3,1,1}) and default_config_pendulum_reward_env.yaml (td3 on
    a RewardEnv with a two-hidden-layer PReLU reward net) through GTN_Master: the synthetic envs carry the reference's state-dict
    keys, a short generation runs, and the plain-td3 fitness values equal the oracle's."""
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.configs import fixed_work, pendulum_reward_env_td3, pendulum_syn_env_td3, with_vary
    from oracle import oracle as orc
    base = fixed_work(pendulum_syn_env_td3(num_workers=2, max_iterations=1), 2)
    base["envs"]["Pendulum-v0"]["max_steps"] = 8
    base["agents"]["td3"].update(init_episodes=1, test_episodes=2)
    m = _master_pair(with_vary(base), tmp_path, monkeypatch)
    assert isinstance(m.task, tasks.Td3VaryTask) and m.cfg.virtual_env == 1 and (m.cfg.state_dim, m.cfg.action_dim, m.cfg.max_action) == (3, 1, 2.0)
    assert m.synthetic_env_orig.is_virtual_env() and tuple(m.synthetic_env_orig.state_dict()["env.state_net.0.weight"].shape) == (32, 4)
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and np.isfinite(mean_score) and m.inner.status.cpu().tolist() == [0] * 6
    for cfg in (base, fixed_work(pendulum_reward_env_td3(num_workers=2, max_iterations=1), 2)):
        cfg["envs"]["Pendulum-v0"]["max_steps"] = 8
        cfg["agents"]["td3"].update(init_episodes=1, test_episodes=2, batch_size=32)
        m = _master_pair(cfg, tmp_path, monkeypatch)
        if cfg["agents"]["gtn"]["synthetic_env_type"] == 1:
            assert m.p_theta == (3 * 128 + 128) + (128 * 128 + 128) + (128 + 1) and m.cfg.rn_layers == 2
        theta0 = m.theta.cpu().numpy().copy()
        gathered = m.evaluate_population(0).cpu().numpy()
        eps = m.eps.cpu().numpy()
        oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
        assert np.array_equal(eps, oeps)
        ocfg = orc.td3_cfg_from_config(cfg)
        for p in range(2):
            sc = []
            for kind, sg in enumerate((0.0, 1.0, -1.0)):
                w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
                sc.append(orc.td3_rn_chain(ocfg, w, init[3 * p + kind], rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
            assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])


def test_real_env_cmc_step_matches_oracle():
    """MountainCarContinuous-v0 through EnvWrapper.reset/step on the device against the oracle's step, with same_action_num 2
    (EnvWrapper.step repeats the action and sums the rewards, env_wrapper.py:56-61): swing up to the flag, done there, +100."""
    from learning_environments_amd.configs import cmc_syn_env_td3
    from learning_environments_amd.envs.env_factory import EnvFactory
    from oracle import oracle as orc
    import ctypes as C
    real = EnvFactory(cmc_syn_env_td3()).generate_real_env()
    assert (real.get_state_dim(), real.get_action_dim(), real.get_max_action()) == (2, 1, 1) and not real.has_discrete_action_space()
    real.set_agent_params(same_action_num=2, gamma=0.99)
    s = real.reset()
    st = (C.c_double * 2)(*real.env._alloc()["state"].cpu().tolist())
    assert -0.6 <= st[0] <= -0.4 and st[1] == 0.0 and float(s[0]) == np.float32(st[0])
    rew, dn = C.c_double(), C.c_int()
    reached = False
    for t in range(400):
        a = np.float32(1.0 if st[1] >= 0 else -1.0)
        ns, r, d = real.step(torch.tensor([float(a)]))
        rsum = 0.0
        for _ in range(2):
            orc.lib().orc_cmc_step(st, (C.c_float * 1)(a), C.byref(rew), C.byref(dn))
            rsum = rsum + float(np.float32(rew.value))        # the one-step device API hands rewards back as fp32
            if dn.value:
                break
        assert np.array_equal(ns.numpy(), np.array([st[0], st[1]]).astype(np.float32)) and float(r) == np.float32(rsum)
        assert float(d) == float(dn.value)
        if dn.value:
            reached = float(r) > 99.0
            break
    assert reached and t < 300


def test_gtn_master_cmc_configs_as_shipped(tmp_path, monkeypatch):
    """default_config_cmc.yaml (td3, VirtualEnv 3-96-96-{2,1,1}) and default_config_cmc_reward_env.yaml (td3, tanh reward net), both
    with same_action_num 2, through GTN_Master: fitness values equal to the oracle's."""
    from learning_environments_amd.configs import cmc_reward_env_td3, cmc_syn_env_td3, fixed_work
    from oracle import oracle as orc
    for make in (cmc_syn_env_td3, cmc_reward_env_td3):
        cfg = fixed_work(make(num_workers=2, max_iterations=1), 2)
        cfg["envs"]["MountainCarContinuous-v0"]["max_steps"] = 9
        cfg["agents"]["td3"].update(init_episodes=1, test_episodes=2, batch_size=32)
        m = _master_pair(cfg, tmp_path, monkeypatch)
        assert m.cfg.same_action_num == 2 and (m.cfg.state_dim, m.cfg.action_dim) == (2, 1)
        theta0 = m.theta.cpu().numpy().copy()
        gathered = m.evaluate_population(0).cpu().numpy()
        eps = m.eps.cpu().numpy()
        oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.cpu().numpy())
        assert np.array_equal(eps, oeps)
        ocfg = orc.td3_cfg_from_config(cfg)
        for p in range(2):
            sc = []
            for kind, sg in enumerate((0.0, 1.0, -1.0)):
                w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
                sc.append(orc.td3_rn_chain(ocfg, w, init[3 * p + kind], rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
            assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])
        mean_score, mean_list, _ = m.run()
        assert len(mean_list) == 1 and np.isfinite(mean_score)


def test_gtn_master_td3_vary_generation(tmp_path, monkeypatch):
    """`agent_name: TD3_vary` through GTN_Master (agents/TD3_vary.py): per-chain draws, one launch, oracle-equal fitness."""
    from learning_environments_amd import _lib
    from learning_environments_amd.agents import tasks
    from learning_environments_amd.config import td3_layer_dims
    from learning_environments_amd.configs import fixed_work, halfcheetah_reward_env_td3, with_vary
    from oracle import oracle as orc
    cfg = fixed_work(halfcheetah_reward_env_td3(num_workers=2, max_iterations=1), 2)
    cfg["envs"]["HalfCheetah-v3"]["max_steps"] = 5
    cfg["agents"]["td3"].update(init_episodes=1, batch_size=32, hidden_size=40)
    cfg = with_vary(cfg)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert isinstance(m.task, tasks.Td3VaryTask) and m.inner.vary and m.cfg.batch_size == 96 and m.cfg.hidden == 120 and m.cfg.layers == 3
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    hps = m.task.last_hp
    for p in range(2):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            h = hps[3 * p + kind]
            key = orc.chain_key(m.seed, 0, p, kind)
            ocfg = orc.td3_cfg_from_config(cfg, lr=float(h["lr"]), batch_size=int(h["batch_size"]), hidden=int(h["hidden_size"]),
                                           layers=max(1, int(h["hidden_layer"])))
            pc = _lib.Td3Cfg()
            for f, _ in _lib.Td3Cfg._fields_:
                setattr(pc, f, getattr(ocfg, f, 0))
            init = orc.agent_init_from_key(key, td3_layer_dims(pc))
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.td3_rn_chain(ocfg, w, init, rng_key=key)["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1 and np.isfinite(mean_score)


def test_reward_env_all_types_on_vector_state_env(golden):
    """EnvWrapper.step on the RewardEnv over the stand-in for all 11 reward types (reward_env.py:29-133), incl. the
    info-vector types: next states bit-equal to the reference run, shaped rewards within 5e-6, state-dict keys equal."""
    from learning_environments_amd.envs.env_factory import EnvFactory
    g = golden("g2f_reward_env_cheetah_info")
    for t in g["types"]:
        t = int(t)
        pre = "t%d_" % t
        cfg = json.loads(str(g["config_json"]))
        cfg["device"] = "cuda"
        cfg["envs"]["HalfCheetah-v3"]["reward_env_type"] = t
        renv = EnvFactory(cfg).generate_reward_env()
        assert list(renv.state_dict().keys()) == [str(k) for k in g[pre + "sd_keys"]]
        if t in (101, 102):
            with torch.no_grad():
                renv.env.reward_net.weight.copy_(torch.from_numpy(g[pre + "theta"]).reshape(1, -1))
            renv.env._flat = None
        else:
            _load_theta(renv, g[pre + "theta"])
        renv.set_agent_params(same_action_num=1, gamma=float(g["gamma"]))
        renv.reset()
        st = renv.env.real_env._alloc()["state"]
        st.copy_(torch.from_numpy(g[pre + "reset_state"]).to(st.device))
        renv.env.state = g[pre + "reset_state"].astype(np.float32)
        for k in range(g[pre + "actions"].shape[0]):
            ns, r, d = renv.step(torch.from_numpy(g[pre + "actions"][k].copy()))
            assert np.array_equal(ns.numpy(), g[pre + "next_states"][k]), (t, k)
            assert abs(float(r) - float(g[pre + "shaped"][k])) <= 5e-6, (t, k, float(r), float(g[pre + "shaped"][k]))
            assert float(d) == float(g[pre + "done"][k])


def test_grid_reward_env_info_types_raise_like_reference():
    """Gridworlds return an empty info dict, so the info-vector reward types raise ValueError (reward_env.py:95-96)."""
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.env_factory import EnvFactory
    cfg = cliff_reward_env_ql(4)
    cfg["device"] = "cuda"
    cfg["envs"]["Cliff"]["reward_env_type"] = 4
    cfg["envs"]["Cliff"]["info_dim"] = 2
    renv = EnvFactory(cfg).generate_reward_env()
    renv.set_agent_params(same_action_num=1, gamma=0.8)
    renv.reset()
    with pytest.raises(ValueError):
        renv.step(torch.tensor([0]))


def test_reference_checkpoint_round_trip(golden, tmp_path):
    """a checkpoint WRITTEN BY THE REFERENCE ({'model','config'}, GTN_master.py:133-139) loads through
    load_envs_and_config; EnvWrapper.step on it reproduces the reference's outputs; re-saving through our state_dict gives the same file
    payload.  (train_test_agents on such a checkpoint: tests/test_train_test_agents.py, against the reference's own function.)"""
    import shutil
    from learning_environments_amd.experiments.syn_env_evaluate import load_envs_and_config
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    shutil.copy(os.path.join(here, "ckpt_cartpole_se_reference.pt"), tmp_path / "model.pt")
    venv, real_env, config = load_envs_and_config("model.pt", str(tmp_path), "cuda")
    g = golden("ckpt_cartpole_se_reference_steps")
    assert venv.is_virtual_env() and not real_env.is_virtual_env()
    assert np.array_equal(venv.env.flat_params().cpu().numpy(), g["theta"])
    for k in range(g["states"].shape[0]):
        ns, r, d = venv.step(action=torch.tensor([float(g["actions"][k])]), state=torch.from_numpy(g["states"][k].copy()))
        np.testing.assert_allclose(ns.numpy().reshape(-1), g["next_states"][k], rtol=0, atol=2e-6)
        assert abs(float(r) - float(g["rewards"][k])) <= 2e-6 and float(d) == float(g["dones"][k])
    # our state_dict() re-saves to the same payload (same keys, shapes, dtypes and values)
    ref = torch.load(os.path.join(here, "ckpt_cartpole_se_reference.pt"), map_location="cpu")
    ours = {k: v.detach().cpu() for k, v in venv.state_dict().items()}
    assert list(ours.keys()) == list(ref["model"].keys())
    for k in ours:
        assert ours[k].dtype == ref["model"][k].dtype and torch.equal(ours[k], ref["model"][k]), k


def test_experiment_wrapper_compute(tmp_path, monkeypatch):
    """the BOHB-facing compute() runs GTN_Master and returns the reference's payload; failures map to
    loss = +Inf (experiments/GTNC_evaluate_cartpole.py:36-73)."""
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.experiments.GTNC_evaluate import ExperimentWrapper
    monkeypatch.chdir(tmp_path)
    cfg = cliff_reward_env_ql(num_workers=4, max_iterations=2)
    cfg["agents"]["gtn"]["quit_when_solved"] = False
    ew = ExperimentWrapper(cfg)
    assert ew.get_bohb_parameters()["eta"] == 2
    res = ew.compute(working_dir=str(tmp_path), bohb_id=7, config_id=(0, 0, 0), cso={}, budget=1)
    assert res["loss"] == 2 and res["info"]["error"] == ""
    bad = dict(cfg, agents=dict(cfg["agents"], gtn=dict(cfg["agents"]["gtn"], agent_name="ppo")))
    res = ExperimentWrapper(bad).compute(working_dir=str(tmp_path), bohb_id=8, config_id=(0, 0, 1), cso={}, budget=1)
    assert res["loss"] == float("inf") and "NotImplementedError" in res["info"]["error"]


def test_gtn_master_num_grad_evals(tmp_path, monkeypatch):
    """num_grad_evals = 2, grad_eval_type 'minmax' (GTN_worker.py:84-104,234-242): 1 + 2*2 chains per worker in one launch,
    per-worker (score_best, score_orig, sign) equal to the oracle's chains + calc_best_score."""
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.gridworld import transition_tables
    from oracle import oracle as orc
    cfg = cliff_reward_env_ql(num_workers=5, max_iterations=1)
    cfg["agents"]["gtn"].update(quit_when_solved=False, num_grad_evals=2, grad_eval_type="minmax")
    cfg["agents"]["ql"]["eps_init"] = cfg["agents"]["ql"]["eps_min"] = 0.3      # make the evaluations of one direction differ
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.cpw == 5
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    tables = transition_tables("Cliff")
    ocfg = orc.ql_cfg_from_config(cfg, tables)
    for p in range(5):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, 1.0, -1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.ql_rn_chain(ocfg, w, tables, rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        best, sign = orc.worker_best_multi(np.array([sc[1:3]]), np.array([sc[3:5]]), True, "minmax")
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == best[0] and gathered[p, 2] == sign[0]
    mean_score, mean_list, _ = m.run()
    assert len(mean_list) == 1


def test_gtn_master_sarsa_cb_generation(tmp_path, monkeypatch):
    """GTN with a count-based SARSA inner agent (agent_utils.py:63-64) on the Cliff RewardEnv: per-worker results equal to the
    oracle's chains."""
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.gridworld import transition_tables
    from oracle import oracle as orc
    cfg = cliff_reward_env_ql(num_workers=4, max_iterations=1)
    cfg["agents"]["gtn"].update(quit_when_solved=False, agent_name="SARSA_cb")
    cfg["agents"]["sarsa"] = dict(cfg["agents"]["ql"], eps_init=0.1, eps_min=0.1, eps_decay=0.0, beta=0.05)
    m = _master_pair(cfg, tmp_path, monkeypatch)
    assert m.cfg.agent_kind == 1 and m.cfg.count_based == 1
    theta0 = m.theta.cpu().numpy().copy()
    gathered = m.evaluate_population(0).cpu().numpy()
    eps = m.eps.cpu().numpy()
    tables = transition_tables("Cliff")
    ocfg = orc.ql_cfg_from_config(cfg, tables)
    for p in range(4):
        sc = []
        for kind, sg in enumerate((0.0, 1.0, -1.0)):
            w = (np.float32(sg) * eps[p] + theta0).astype(np.float32)
            sc.append(orc.ql_rn_chain(ocfg, w, tables, rng_key=orc.chain_key(m.seed, 0, p, kind))["score"])
        assert gathered[p, 1] == sc[0] and gathered[p, 0] == max(sc[1], sc[2])


def test_ddqn_inner_loop_captured_on_first_call():
    """include/lenv_hip.h's contract (never allocate, never synchronise, graph-capturable): the FIRST call of
    lenv_ddqn_se_inner_loop for a configuration is captured into a HIP graph on a non-default stream under the strictest capture
    mode, and two replays reproduce an eager launch bit for bit (the Adam bias-correction table is rebuilt by a prologue kernel
    into the caller's workspace on every launch)."""
    from learning_environments_amd import engine
    from learning_environments_amd.config import ddqn_cfg_from_config
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    cfgd = fixed_work(cartpole_syn_env_ddqn(num_workers=2), 3)
    cfgd["envs"]["CartPole-v0"]["max_steps"] = 25
    cfgd["agents"]["ddqn"]["lr"] = 1.2345e-3                 # a learning rate no other test used: nothing can be cached
    cfg = ddqn_cfg_from_config(cfgd)
    chains = 6
    bounds = torch.full((401,), 0.4, device="cuda")
    eps, init, keys = engine.nes_draw(77, 0, 2, 2247, 0.0124, chains, 3, 0, bounds)
    theta = (torch.randn(2247, generator=torch.Generator().manual_seed(3)) * 0.1).cuda()
    worker = torch.arange(2, dtype=torch.int32).repeat_interleave(3).cuda()
    sign = torch.tensor([0.0, 1.0, -1.0] * 2, device="cuda")
    torch.cuda.synchronize()
    il = engine.InnerLoop(cfg, chains, want_final_online=True)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
        il.run(theta, eps, worker, sign, init, rng_keys=keys)
    outs = []
    for _ in range(2):
        il.score.fill_(-1.0); il.final_online.zero_(); il.status.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        outs.append((il.score.cpu().clone(), il.final_online.cpu().clone(), il.stats.cpu().clone()))
        assert il.status.cpu().tolist() == [0] * chains
    il2 = engine.InnerLoop(cfg, chains, want_final_online=True)
    il2.run(theta, eps, worker, sign, init, rng_keys=keys)
    torch.cuda.synchronize()
    for sc, fo, st in outs:
        assert torch.equal(sc, il2.score.cpu()) and torch.equal(fo, il2.final_online.cpu()) and torch.equal(st, il2.stats.cpu())
    assert int(il2.stats[:, 2].min()) > 0                     # learn steps happened (the table was read)


@pytest.mark.parametrize("which", ["ddqn", "ql"])
def test_gtn_master_graph_generation_equals_eager(which, tmp_path, monkeypatch):
    """A generation replayed as ONE captured graph (draw -> fused -> worker_best -> status_fold -> score_transform + update_env,
    generation counter on the device) gives the same fitness lists and the same theta as the eager launches, generation after
    generation, including the save-before-update semantics of reference GTN_master.py:95-101."""
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, cliff_reward_env_ql, fixed_work
    monkeypatch.chdir(tmp_path)
    if which == "ddqn":
        cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=4, max_iterations=3), 2)
        cfg["envs"]["CartPole-v0"]["max_steps"] = 12
        cfg["agents"]["ddqn"]["test_episodes"] = 2
    else:
        cfg = cliff_reward_env_ql(num_workers=6, max_iterations=3)
        cfg["agents"]["gtn"]["quit_when_solved"] = False
    torch.manual_seed(0)
    a = GTN_Master(cfg, bohb_id=0, seed=9, graph=True)
    torch.manual_seed(0)
    b = GTN_Master(cfg, bohb_id=1, seed=9, graph=False)
    assert a.use_graph and not b.use_graph and torch.equal(a.theta, b.theta)
    for it in range(3):
        ra, rb = a.step(it), b.step(it)
        assert ra == rb
        assert a.score_list == b.score_list and a.score_orig_list == b.score_orig_list
        assert torch.equal(a.theta, b.theta), it
        assert a.get_score_transform_list() == b.get_score_transform_list()
    # a generation out of sequence (the device counter is reset) and the explicit reference-style calls still work
    ra, rb = a.step(7), b.step(7)
    assert ra == rb and torch.equal(a.theta, b.theta)
    if which == "ql":             # RN runs save whenever the mean improves: the file holds the PRE-update theta on both paths
        sa = torch.load(os.path.join(a.model_dir, os.path.basename(a.model_name)), weights_only=False)["model"]
        sb = torch.load(os.path.join(b.model_dir, os.path.basename(b.model_name)), weights_only=False)["model"]
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
