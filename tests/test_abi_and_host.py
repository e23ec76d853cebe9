"""CPU suite: the C-ABI library loads and exports every symbol include/lenv_hip.h declares (no compute calls without a
GPU), argument validation that needs no device, and the host-side mirror of the reference API."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from learning_environments_amd import _lib
    header = open(os.path.join(ROOT, "include", "lenv_hip.h")).read()
    declared = set(re.findall(r"\b(lenv_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "liblenv_hip.so does not export " + name
    assert set(_lib.EXPORTS) <= declared
    assert L.lenv_abi_version() == 7
    assert L.lenv_error_string(-2) == b"unsupported shape or option"


def test_struct_mirrors_documented_in_integration_md_have_the_library_sizes():
    """Every `class X(C.Structure)` a maintainer can paste from INTEGRATION.md is built here and its ctypes.sizeof compared with
    lenv_struct_size (include/lenv_hip.h) -- a stub one field short hands the library a struct whose tail is whatever follows it."""
    from learning_environments_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    L = _lib.lib()
    L.lenv_struct_size.restype = C.c_int64
    L.lenv_struct_size.argtypes = [C.c_int32]
    index = {cls.__name__: i for i, cls in enumerate(_lib.ABI_STRUCTS)}
    found = []
    # a class statement and its (possibly multi-line) _fields_ list, up to the closing bracket
    for m in re.finditer(r"^class (\w+)\(C\.Structure\):[^\n]*\n((?:[ \t]+[^\n]*\n)+)", text, re.M):
        name, body = m.group(1), m.group(2)
        ns = {"C": C}
        exec("class %s(C.Structure):\n%s" % (name, body), ns)
        assert name in index, "INTEGRATION.md documents a struct the ABI table does not know: " + name
        assert C.sizeof(ns[name]) == L.lenv_struct_size(index[name]), \
            "INTEGRATION.md's %s is %d bytes, the library's is %d" % (name, C.sizeof(ns[name]), L.lenv_struct_size(index[name]))
        assert [f[0] for f in ns[name]._fields_] == [f[0] for f in _lib.ABI_STRUCTS[index[name]]._fields_]
        found.append(name)
    assert "MlpDesc" in found
    # and the package's own mirrors, the same way (what _lib.lib() checks at load)
    for i, cls in enumerate(_lib.ABI_STRUCTS):
        assert C.sizeof(cls) == L.lenv_struct_size(i), cls.__name__


def test_host_only_entry_points():
    from learning_environments_amd import _lib, config, configs
    from oracle import oracle as orc
    L = _lib.lib()
    d = _lib.MlpDesc(6, 83, 1, 4, 2, 0.25)
    assert L.lenv_mlp_num_params(C.byref(d)) == 6 * 83 + 83 + 83 * 4 + 4
    for args in ((1, 2, 3, 1), (0, 0, 0, 0), (2 ** 40, 7, 63, 2)):
        assert L.lenv_chain_key(*args) == orc.chain_key(*args)
    cfgd = configs.fixed_work(configs.cartpole_syn_env_ddqn(), 20)
    cfg = config.ddqn_cfg_from_config(cfgd)
    assert cfg.grad_chunk >= (cfg.batch_size + 15) // 16
    lds = L.lenv_ddqn_se_lds_bytes(C.byref(cfg))
    assert 0 < lds <= 160 * 1024
    assert L.lenv_ddqn_se_workspace_bytes(C.byref(cfg), 192) >= 192 * 4000 * 12 * 4
    bad = _lib.DdqnCfg.from_buffer_copy(cfg)
    bad.q_layers = 2
    assert L.lenv_ddqn_se_lds_bytes(C.byref(bad)) == -2          # NotImplementedError path
    # argument validation happens before any device work
    assert L.lenv_se_step_population(C.byref(d), C.byref(d), C.byref(d), None, None, None, None, 1, 1, None, None, None, None, None, None) == -1
    assert L.lenv_nes_rank_update(9, None, None, 4, None, None, 0, 0.1, 0, 0.0, None, None) == -1


def test_forward_layout_decisions_of_the_ddqn_kernel():
    """`lenv_ddqn_se_forward_split` (host only): minibatches of more than 170 samples spill forward items beyond the workgroup's
    first eight waves; the kernel cuts the spilled items into 4 / 3 / 2 parts (<= 64 / <= 85 / <= 128 items) unless the net has
    fewer than eight hidden-unit pairs or the shared activation rows do not fit the 160 KiB of LDS next to everything else."""
    from learning_environments_amd import _lib, configs
    from learning_environments_amd.config import ddqn_cfg_from_config
    L = _lib.lib()
    cfg = ddqn_cfg_from_config(configs.cartpole_syn_env_ddqn(4))            # BASELINE configs[1]: B 199, 4-57-2, SE hidden 83
    items, parts = C.c_int32(), C.c_int32()
    assert L.lenv_ddqn_se_forward_split(C.byref(cfg), C.byref(items), C.byref(parts)) == 0 and (items.value, parts.value) == (85, 3)
    assert L.lenv_ddqn_se_lds_bytes(C.byref(cfg)) <= 160 * 1024
    expect = {(170, 57): (0, 0), (171, 57): (1, 4), (180, 16): (28, 4), (200, 24): (88, 2), (213, 33): (127, 2),
              (213, 57): (0, 0),      # the rows of 127 items would not fit: plain layout, still with the 16 tanh-table copies
              (199, 10): (0, 0),      # five pairs: too narrow to cut
              (199, 64): (0, 0)}      # does not fit either
    for (B, H), want in expect.items():
        c = _lib.DdqnCfg.from_buffer_copy(cfg)
        c.batch_size, c.q_hidden, c.grad_chunk = B, H, (B + 11) // 12
        assert L.lenv_ddqn_se_forward_split(C.byref(c), C.byref(items), C.byref(parts)) == 0, (B, H)
        assert (items.value, parts.value) == want, (B, H, items.value, parts.value)
        assert 0 < L.lenv_ddqn_se_lds_bytes(C.byref(c)) <= 160 * 1024
    assert L.lenv_ddqn_se_forward_split(None, C.byref(items), C.byref(parts)) != 0


def test_chain_keys_vectorised_matches_abi():
    from learning_environments_amd import _lib
    from learning_environments_amd.agents.nes_common import chain_keys, shard_bounds
    from oracle import oracle as orc
    L = _lib.lib()
    w = np.repeat(np.arange(5, 9), 3)
    k = np.tile(np.arange(3), 4)
    got = chain_keys(1234, 17, w, k)
    for i in range(12):
        assert int(got[i]) == L.lenv_chain_key(1234, 17, int(w[i]), int(k[i]))
    # every (worker, kind) of a generation has its own key, also with num_grad_evals 2 and 3 (5 / 7 chains per worker):
    # the old key mixed worker*4 + kind, so (w, 4 + j) collided with (w + 1, j)
    for g in (1, 2, 3):
        cpw = 1 + 2 * g
        ws, ks = np.repeat(np.arange(64), cpw), np.tile(np.arange(cpw), 64)
        keys = chain_keys(7, 3, ws, ks)
        assert len(set(keys.tolist())) == 64 * cpw
        assert int(keys[cpw + 1]) == L.lenv_chain_key(7, 3, 1, 1) == orc.chain_key(7, 3, 1, 1)
    # sharding: contiguous blocks cover the population exactly once
    for pop, world in ((64, 8), (64, 1), (10, 4), (3, 8)):
        seen = []
        for r in range(world):
            lo, hi, per = shard_bounds(pop, r, world)
            assert hi - lo <= per
            seen += list(range(lo, hi))
        assert seen == list(range(pop))


def test_rank_table_matches_reference_vectors(golden):
    from learning_environments_amd.agents.nes_common import rank_table
    from oracle import oracle as orc
    g = golden("g7_master")
    n = g["scores"].size
    # type 1: table by ascending rank == reference output sorted
    assert np.allclose(np.sort(g["tf1"]), rank_table(1, n))
    # types 2/3: raw utilities, normalised in worker order by the kernel; here via the oracle
    for t in (2, 3):
        assert np.allclose(orc.score_transform(t, g["scores"], g["scores_orig"]), g["tf%d" % t], rtol=1e-15, atol=1e-15)
        raw = rank_table(t, n)
        assert raw[0] == pytest.approx(np.log(n / 2 + 1)) and np.all(np.diff(raw) <= 0) and raw.min() == 0.0


def test_model_layout_matches_reference_state_dict(golden):
    """State-dict keys/shapes of the SE are the drop-in surface; the flat packing order is the
    reference's state-dict order."""
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    from learning_environments_amd.envs.env_factory import EnvFactory
    from learning_environments_amd.models.model_utils import linear_params, mlp_desc
    fac = EnvFactory(cartpole_syn_env_ddqn())
    venv = fac.generate_virtual_env()
    sd = venv.state_dict()
    assert list(sd.keys()) == ['env.%s.%d.%s' % (n, i, p) for n in ('state_net', 'reward_net', 'done_net')
                               for i in (0, 2) for p in ('weight', 'bias')]
    assert tuple(sd['env.state_net.0.weight'].shape) == (83, 6) and tuple(sd['env.done_net.2.weight'].shape) == (1, 83)
    assert sum(p.numel() for p in linear_params(venv)) == 2247
    d = mlp_desc(venv.env.state_net, "leakyrelu")
    assert (d.in_dim, d.hidden, d.layers, d.out_dim, d.act) == (6, 83, 1, 4, 2)
    # a reference-produced theta round-trips through load_state_dict
    g = golden("g6_worker_noise")
    off, new = 0, {}
    for k, v in sd.items():
        new[k] = torch.from_numpy(g["theta"][off:off + v.numel()].reshape(tuple(v.shape)).copy())
        off += v.numel()
    venv.load_state_dict(new)
    flat = torch.cat([p.detach().reshape(-1) for p in linear_params(venv)]).numpy()
    assert np.array_equal(flat, g["theta"])
    # API surface of the wrapper
    assert venv.is_virtual_env() and venv.has_discrete_action_space() and not venv.has_discrete_state_space()
    assert venv.get_state_dim() == 4 and venv.get_action_dim() == 2 and venv.max_episode_steps() == 200
    assert venv.get_solved_reward() == 195.0 and venv.can_be_solved()
    real = fac.generate_real_env()
    assert not real.is_virtual_env() and real.max_episode_steps() == 200 and real.get_min_action() == 0
    if torch.cuda.is_available():
        renv = fac.generate_reward_env()        # constructible like the reference's; continuous-state shaping is "next"
        assert not renv.is_virtual_env()
        with pytest.raises(NotImplementedError):
            renv.env.ql_cfg()
    else:
        from learning_environments_amd import _lib
        with pytest.raises(_lib.LenvError):     # RewardEnv.__init__ resets the (device-resident) real env
            fac.generate_reward_env()


def test_reward_env_layout_cliff():
    from learning_environments_amd.configs import cliff_reward_env_ql
    from learning_environments_amd.envs.env_factory import EnvFactory
    from learning_environments_amd.models.model_utils import linear_params
    fac = EnvFactory(cliff_reward_env_ql())
    renv = fac.generate_reward_env()
    assert list(renv.state_dict().keys()) == ['env.reward_net.0.weight', 'env.reward_net.0.bias', 'env.reward_net.1.weight',
                                              'env.reward_net.2.weight', 'env.reward_net.2.bias']
    assert sum(p.numel() for p in linear_params(renv)) == 1601          # the PReLU slope is not perturbed (GTN_worker.py:158)
    assert renv.has_discrete_state_space() and renv.has_discrete_action_space()
    assert renv.get_state_dim() == 48 and renv.get_action_dim() == 4 and renv.max_episode_steps() == 50
    assert renv.reset().tolist() == [36.0]
    real = fac.generate_real_env()
    assert real.reset().tolist() == [36.0]
    ns, r, d = real.step(torch.tensor([3.0]))          # up
    assert (ns.tolist(), float(r), float(d)) == ([24.0], -1.0, 0.0)
    ns, r, d = real.step(torch.tensor([2.0]))          # down, back to start
    ns, r, d = real.step(torch.tensor([0.0]))          # right: into the cliff
    assert (ns.tolist(), float(r), float(d)) == ([37.0], -100.0, 1.0)
    cfg = renv.env.ql_cfg()
    assert (cfg.n_states, cfg.n_actions, cfg.start_state, cfg.rn_hidden, cfg.rn_act, cfg.reward_env_type) == (48, 4, 36, 32, 4, 2)


def test_prelu_and_layers_in_model_builder():
    from learning_environments_amd.models.model_utils import build_nn_from_config, linear_params, mlp_desc
    net = build_nn_from_config(9, 6, {"hidden_size": 16, "hidden_layer": 2, "activation_fn": "prelu"})
    keys = list(net.state_dict().keys())
    assert keys == ['0.weight', '0.bias', '1.weight', '2.weight', '2.bias', '4.weight', '5.weight', '5.bias']
    assert sum(p.numel() for p in linear_params(net)) == 9 * 16 + 16 + 16 * 16 + 16 + 16 * 6 + 6
    d = mlp_desc(net, "prelu")
    assert (d.layers, d.act) == (2, 4) and d.prelu == pytest.approx(0.25)
    with pytest.raises(NotImplementedError):
        build_nn_from_config(4, 2, {"hidden_size": 8, "hidden_layer": 1, "activation_fn": "gelu"})


def test_one_hot_helpers():
    from learning_environments_amd.utils import AverageMeter, from_one_hot_encoding, to_one_hot_encoding
    assert to_one_hot_encoding(torch.tensor([1.0]), 3).tolist() == [0, 1, 0]
    assert to_one_hot_encoding(2, 4).tolist() == [0, 0, 1, 0]
    assert to_one_hot_encoding(torch.tensor([0.0, 2.9]), 3).tolist() == [[1, 0, 0], [0, 0, 1]]     # int() truncation
    assert from_one_hot_encoding(torch.tensor([0.0, 0.0, 1.0])).tolist() == [2]
    m = AverageMeter("x")
    for v in (1.0, 2.0, 3.0, 4.0):
        m.update(v, print_rate=10 ** 9)
    assert m.get_mean(num=2) == pytest.approx(3.5, abs=1e-6) and m.get_mean_last(num=2) == pytest.approx(1.5, abs=1e-6)


def test_product_path_fails_loudly_without_device():
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from learning_environments_amd import _lib, engine
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    with pytest.raises(_lib.LenvError):
        engine.require_device()
    with pytest.raises(_lib.LenvError):
        GTN_Master(cartpole_syn_env_ddqn(num_workers=2))


def test_master_rejects_unknown_options(tmp_path, monkeypatch):
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    monkeypatch.chdir(tmp_path)
    cfg = cartpole_syn_env_ddqn(num_workers=2)
    cfg["agents"]["gtn"]["score_transform_type"] = 9
    with pytest.raises(ValueError):
        GTN_Master(cfg, engine=OracleNesEngine())
    cfg = cartpole_syn_env_ddqn(num_workers=2)
    cfg["agents"]["gtn"]["synthetic_env_type"] = 5
    with pytest.raises(NotImplementedError):
        GTN_Master(cfg, engine=OracleNesEngine())
    cfg = cartpole_syn_env_ddqn(num_workers=2)
    cfg["agents"]["gtn"]["agent_name"] = "PPO"
    with pytest.raises(NotImplementedError):
        GTN_Master(cfg, engine=OracleNesEngine())


def test_reference_checkpoint_fixture_is_plain_payload():
    """The committed reference-format checkpoint ({'model','config'}, GTN_master.py:133-139) loads under torch's safe
    loader and carries the state-dict layout SURVEY.md §8(b) lists."""
    import os
    import torch
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_cartpole_se_reference.pt")
    d = torch.load(path, map_location="cpu", weights_only=True)
    assert sorted(d.keys()) == ["config", "model"]
    keys = list(d["model"].keys())
    assert keys[:4] == ["env.state_net.0.weight", "env.state_net.0.bias", "env.state_net.2.weight", "env.state_net.2.bias"]
    assert tuple(d["model"]["env.state_net.0.weight"].shape) == (83, 6) and tuple(d["model"]["env.done_net.2.bias"].shape) == (1,)
    assert d["config"]["env_name"] == "CartPole-v0"


def test_hidden_layer_zero_builds_the_one_hidden_layer_net():
    """build_nn_from_config adds `hidden_layer - 1` extra blocks (reference models/model_utils.py:33-37), so hidden_layer 0 --
    which the *_vary agents sample (DDQN_vary.py:44-46: hidden_layer in {L-1, L, L+1}) -- is the same network as 1."""
    from learning_environments_amd.config import ddqn_cfg_from_config
    from learning_environments_amd.configs import cartpole_syn_env_ddqn
    from learning_environments_amd.models.model_utils import build_nn_from_config
    n0 = build_nn_from_config(4, 2, {"hidden_size": 8, "hidden_layer": 0, "activation_fn": "tanh"})
    n1 = build_nn_from_config(4, 2, {"hidden_size": 8, "hidden_layer": 1, "activation_fn": "tanh"})
    assert [type(m) for m in n0] == [type(m) for m in n1] and list(n0.state_dict().keys()) == list(n1.state_dict().keys())
    cfgd = cartpole_syn_env_ddqn(2)
    cfgd["agents"]["ddqn"]["hidden_layer"] = 0
    assert ddqn_cfg_from_config(cfgd).q_layers == 1


def test_master_host_logic_for_icm_and_multilayer_agents_on_the_oracle_engine(tmp_path, monkeypatch):
    """GTN_Master + select_task on CPU tensors (oracle-backed engine, no GPU): `DDQN_icm` and a two-hidden-layer DDQN go through
    the same host path as on the GPU -- config parsing, task choice, per-chain keys, fresh agents / ICMs from the counter RNG --
    and the fitness records equal a direct oracle evaluation."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work, with_icm
    from oracle import oracle as orc
    monkeypatch.chdir(tmp_path)
    base = fixed_work(cartpole_syn_env_ddqn(num_workers=2, max_iterations=1), 2)
    base["envs"]["CartPole-v0"]["max_steps"] = 8
    base["agents"]["ddqn"].update(test_episodes=2, init_episodes=1, batch_size=12, hidden_size=10)
    deep = fixed_work(cartpole_syn_env_ddqn(num_workers=2, max_iterations=1), 2)
    deep["envs"]["CartPole-v0"]["max_steps"] = 8
    deep["agents"]["ddqn"].update(test_episodes=2, init_episodes=1, batch_size=12, hidden_size=10, hidden_layer=2)
    for cfg in (with_icm(base, feature_dim=6, hidden_size=8), deep):
        torch.manual_seed(0)
        m = GTN_Master(cfg, bohb_id=0, engine=OracleNesEngine(), seed=4)
        assert m.cfg.grad_chunk == 0
        theta0 = m.theta.numpy().copy()
        gathered = m.evaluate_population(0).numpy()
        eps = m.eps.numpy()
        oeps, init, okeys = orc.nes_draw(m.seed, 0, 2, m.p_theta, cfg["agents"]["gtn"]["noise_std"], 6, 3, 0, m.agent_bounds.numpy())
        assert np.array_equal(eps, oeps)
        ocfg = orc.ddqn_cfg_from_config(cfg, grad_chunk=0)
        scores = []
        for c in range(6):
            key = orc.chain_key(m.seed, 0, c // 3, c % 3)
            icm_init = orc.agent_init_from_key(key, orc.icm_layer_dims(ocfg), stream=orc.STREAM_ICM_INIT) if ocfg.icm_enabled else None
            w = (np.float32([0.0, 1.0, -1.0][c % 3]) * eps[c // 3] + theta0).astype(np.float32)
            scores.append(orc.ddqn_se_chain(ocfg, w, init[c], rng_key=key, icm_init=icm_init)["score"])
        scores = np.array(scores)
        best, sign = orc.worker_best(scores[1::3], scores[2::3], True)
        assert np.array_equal(gathered[:, 0], best) and np.array_equal(gathered[:, 1], scores[0::3])


def test_config_builders_take_layer_norm_sections():
    """theta / eps are the nn.Linear parameters only; a `use_layer_norm` section must not slip into a fused loop
    that would ignore the normalisation.  Round 4: no builder refuses the flag any more -- the agent's LayerNorm is a parameter block of the
    agent vector, the env nets' LayerNorm a cfg word (it is never perturbed, theta keeps its layout)."""
    from learning_environments_amd import config, configs
    c = configs.cartpole_syn_env_ddqn(2)
    c["envs"]["CartPole-v0"]["use_layer_norm"] = True
    config.ddqn_cfg_from_config(c)                           # one hidden layer: the module is never appended (model_utils.py:33-36) -- the same net
    c["envs"]["CartPole-v0"]["hidden_layer"] = 2
    # the DDQN / DuelingDDQN loop over a synthetic env normalises inside its SE step (cfg.se_layer_norm; theta stays the Linear parameters) ...
    scfg = config.ddqn_cfg_from_config(c)
    assert scfg.se_layer_norm == 1 and scfg.se_layers == 2 and scfg.q_layer_norm == 0
    # ... and so does its reward-env mode (the reward net's LayerNorm)
    c = configs.cartpole_reward_env_ddqn(2)
    c["envs"]["CartPole-v0"].update(use_layer_norm=True, hidden_layer=2)
    rcfg = config.ddqn_cfg_from_config(c)
    assert rcfg.synthetic_env_type == 1 and rcfg.se_layer_norm == 1 and rcfg.se_layers == 2
    # the TD3 family and the tabular agents take the env nets' LayerNorm as a cfg word as well (ABI 6); theta keeps its Linear-only size
    c = configs.halfcheetah_reward_env_td3(2)
    c["envs"]["HalfCheetah-v3"].update(use_layer_norm=True, hidden_layer=2)       # the reward net
    tcfg = config.td3_cfg_from_config(c)
    assert tcfg.rn_layer_norm == 1 and tcfg.rn_layers == 2 and tcfg.use_layer_norm == 0
    c = configs.cartpole_syn_env_td3_discrete(2)
    c["envs"]["CartPole-v0"].update(use_layer_norm=True, hidden_layer=2)
    assert config.td3d_cfg_from_config(c).se_layer_norm == 1
    # TD3 takes the AGENT's LayerNorm (cfg.use_layer_norm): one block per net (actor, critic_1, critic_2) behind its second Linear
    c = configs.halfcheetah_reward_env_td3(2)
    c["agents"]["td3"]["use_layer_norm"] = True
    tcfg = config.td3_cfg_from_config(c)
    sl = config.td3_layer_norm_slices(tcfg)
    pa, pc = (17 * 128 + 128) + (128 * 128 + 128) + 256 + (6 * 128 + 6), (23 * 128 + 128) + (128 * 128 + 128) + 256 + 129
    assert tcfg.use_layer_norm == 1 and sl == [(17 * 128 + 128 + 128 * 128 + 128, 128), (pa + 23 * 128 + 128 + 128 * 128 + 128, 128),
                                               (pa + pc + 23 * 128 + 128 + 128 * 128 + 128, 128)]
    from learning_environments_amd.agents import nes_common as nc
    tb = nc.with_layer_norm_block(nc.linear_init_bounds(config.td3_layer_dims(tcfg)), sl)
    assert tb.size == pa + 2 * pc and all(not tb[o:o + 256].any() and tb[o - 1] > 0 and tb[o + 256] > 0 for o, _ in sl)
    # the DDQN / DuelingDDQN loop takes the AGENT's LayerNorm (cfg.q_layer_norm); the block of the shared module sits behind the second Linear
    from learning_environments_amd.agents import nes_common
    c = configs.acrobot_syn_env_ddqn(2)
    c["agents"]["ddqn"]["use_layer_norm"] = True
    cfg = config.ddqn_cfg_from_config(c)
    assert cfg.q_layer_norm == 1 and config.agent_layer_norm_slice(cfg) == ((6 * 128 + 128) + (128 * 128 + 128), 128)
    b = nes_common.with_layer_norm_block(nes_common.linear_init_bounds(config.agent_layer_dims(cfg)), config.agent_layer_norm_slice(cfg))
    assert b.size == 17795 + 256 and not b[17408:17664].any() and b[17407] > 0 and b[17664] > 0
    c["agents"]["ddqn"]["hidden_layer"] = 1                  # one hidden layer: no position, no parameters
    assert config.agent_layer_norm_slice(config.ddqn_cfg_from_config(c, grad_chunk=4)) is None


def test_team_exchange_area_is_sized_for_every_launch_that_may_be_teamed():
    """the team picker and the exchange area of the DDQN kernel share ONE bound (DDQN_TEAM_MAX_CHAINS = 128 chains): the
    workspace carries chains * team_stride floats for every counter-mode launch of at most 128 chains -- whatever the occupancy API would say --
    and none above; the picker (GPU half: tests/test_gpu_parity.py team tests) returns 1 above the bound before it asks the occupancy API.
    Host-only: the size query does no device work."""
    from learning_environments_amd import _lib, config, configs
    L = _lib.lib()
    cfg = config.ddqn_cfg_from_config(configs.fixed_work(configs.cartpole_syn_env_ddqn(), 20))
    tape = _lib.DdqnCfg.from_buffer_copy(cfg)
    tape.rng_mode = 1                                          # tape mode: never teamed, never an exchange area
    per_chain = None
    for chains in (1, 24, 48, 96, 128):
        extra = L.lenv_ddqn_se_workspace_bytes(C.byref(cfg), chains) - L.lenv_ddqn_se_workspace_bytes(C.byref(tape), chains)
        assert extra > 0 and extra % chains == 0
        per_chain = per_chain or extra // chains
        assert extra == chains * per_chain
    for chains in (129, 192, 400):
        assert L.lenv_ddqn_se_workspace_bytes(C.byref(cfg), chains) == L.lenv_ddqn_se_workspace_bytes(C.byref(tape), chains)


@pytest.mark.gpu
def test_team_picker_respects_the_exchange_area_bound():
    from learning_environments_amd import _lib, config, configs, engine
    engine.require_device()
    L = _lib.lib()
    cfg = config.ddqn_cfg_from_config(configs.fixed_work(configs.cartpole_syn_env_ddqn(), 20))
    assert L.lenv_ddqn_se_team_size(C.byref(cfg), 24) > 1 and L.lenv_ddqn_se_team_size(C.byref(cfg), 96) > 1     # the 8- / 2-GPU shards are teamed
    for chains in (129, 192, 400):
        assert L.lenv_ddqn_se_team_size(C.byref(cfg), chains) == 1
