"""SURVEY.md §8(f).1, one row up: the caller of the evaluation harness -- reference experiments/syn_env_run_vary_hp.py
(`get_all_files` :8-29, `run_vary_hp` :32-139) and its result file (utils.py:144-160 `save_lists`).

CPU half: the file selection rules and the list / file shapes with a stand-in harness function (no GPU work).
GPU half: the product's train_test_agents behind it -- all models of a mode in ONE fused launch (model_num * agents_num chains, chain (m, i)
reading model m's weights) equal to the model-by-model calls bit for bit, and to the oracle chain of that (model, agent)."""
import json
import os
import sys
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from learning_environments_amd.experiments import syn_env_run_vary_hp as rv      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_loader(vary_of):
    def load(file_name, model_dir, device):
        real = types.SimpleNamespace(env=types.SimpleNamespace(env_name="CartPole-v0"))
        return "venv:" + file_name, real, {"agents": {"ddqn_vary": {"vary_hp": vary_of[file_name]}}, "file": file_name}
    return load


def test_get_all_files_rules(tmp_path):
    """:10-29: files of the env only, vary_hp flag of the checkpoint's config, sorted by the LAST NINE characters (the random tag + '.pt'),
    the first model_num of them; too few -> ValueError unless a filter list is given; the filter keeps the sorted order."""
    vary_of = {"CartPole-v0_3_ZZZZZZ.pt": True, "CartPole-v0_25_AAAAAB.pt": True, "CartPole-v0_9_MMMMMM.pt": True, "CartPole-v0_1_BBBBBB.pt": False,
               "Acrobot-v1_1_AAAAAA.pt": True}
    for f in vary_of:
        (tmp_path / f).write_bytes(b"")
    load = _fake_loader(vary_of)
    got = rv.get_all_files(True, 2, str(tmp_path), load, "CartPole", "cpu")
    assert got == ["CartPole-v0_25_AAAAAB.pt", "CartPole-v0_9_MMMMMM.pt"]
    assert rv.get_all_files(False, 1, str(tmp_path), load, "CartPole", "cpu") == ["CartPole-v0_1_BBBBBB.pt"]
    assert rv.get_all_files(True, 1, str(tmp_path), load, "Acrobot", "cpu") == ["Acrobot-v1_1_AAAAAA.pt"]
    with pytest.raises(ValueError):
        rv.get_all_files(True, 4, str(tmp_path), load, "CartPole", "cpu")
    flt = ["CartPole-v0_3_ZZZZZZ.pt", "CartPole-v0_25_AAAAAB.pt", "CartPole-v0_1_BBBBBB.pt"]
    assert rv.get_all_files(True, 40, str(tmp_path), load, "CartPole", "cpu", filter_models_list=flt) == ["CartPole-v0_25_AAAAAB.pt", "CartPole-v0_3_ZZZZZZ.pt"]


@pytest.mark.parametrize("mode,correlation", [(0, False), (1, False), (2, False), (2, True)])
def test_run_vary_hp_lists_and_result_file_with_a_stand_in_harness(tmp_path, mode, correlation):
    """A callable without `.fused` is called model by model like the reference's pool-less branch (:47-63,84-98); the three lists are the
    concatenation over the models (one entry per model with correlation_exp on a syn. env, :112-117); the file holds what save_lists writes."""
    vary_of = {"CartPole-v0_3_ZZZZZZ.pt": True, "CartPole-v0_25_AAAAAB.pt": True, "CartPole-v0_1_BBBBBB.pt": False}
    model_dir = tmp_path / "models"
    model_dir.mkdir()
    for f in vary_of:
        (model_dir / f).write_bytes(b"")
    calls = []

    def harness(train_env, test_env, config, agents_num):
        calls.append(train_env)
        k = len(calls)
        return [[float(10 * k + i)] * 3 for i in range(agents_num)], [[100 * k + i] for i in range(agents_num)], [[k] for _ in range(agents_num)]
    model_num = {0: 2, 1: 1, 2: 2}[mode]
    out = rv.run_vary_hp(mode, "exp", model_num, 2, str(model_dir), _fake_loader(vary_of), harness, "CartPole", device="cpu",
                         correlation_exp=correlation, out_dir=str(tmp_path))
    assert len(calls) == model_num
    if mode == 0:
        assert all(not isinstance(c, str) for c in calls)                   # the real env itself is the training env
        rows = ["CartPole-v0_0", "CartPole-v0_1"]
    elif mode == 1:
        assert calls == ["venv:CartPole-v0_1_BBBBBB.pt"]
        rows = ["CartPole-v0_1_BBBBBB.pt"]
    else:
        assert calls == ["venv:CartPole-v0_25_AAAAAB.pt", "venv:CartPole-v0_3_ZZZZZZ.pt"]
        rows = ["CartPole-v0_25_AAAAAB.pt", "CartPole-v0_3_ZZZZZZ.pt"]
    saved = torch.load(str(tmp_path / ("%d_exp.pt" % mode)), weights_only=False)
    assert set(saved) == {"config", "reward_list", "train_steps_needed", "episode_length_needed", "env_reward_overview"}
    assert saved["reward_list"] == out[0] and saved["train_steps_needed"] == out[1] and saved["episode_length_needed"] == out[2]
    if correlation:
        assert len(out[0]) == model_num and len(out[0][0]) == 2 and saved["env_reward_overview"].shape[1] == 0      # (:114: empty dicts)
    else:
        assert list(saved["env_reward_overview"].index) == rows
        assert len(out[0]) == 2 * model_num and out[1][0] == [100] and out[2][-1] == [model_num]
        assert saved["env_reward_overview"].shape == (model_num, 6)          # np.hstack of the model's agents' return lists
    with pytest.raises(ValueError):
        rv.run_vary_hp(3, "exp", 1, 1, str(model_dir), _fake_loader(vary_of), harness, "CartPole")


def test_result_file_has_the_layout_of_the_reference_written_one(tmp_path):
    """Fixture G13 = the file the REFERENCE's run_vary_hp wrote (mode 2, two checkpoints named as below, two agents each; oracle/gen_golden.py
    g13).  The product's run_vary_hp on the same directory layout with a stand-in harness returning lists of the same shapes writes a file
    with the same keys, the same nesting of the three lists, the same DataFrame (rows = checkpoints in get_all_files order, one column per
    test return of the checkpoint's agents) and the comparability settings in the saved config."""
    ref = torch.load(os.path.join(HERE, "golden", "g13_ref_run_vary_hp_mode2.pt"), weights_only=False)
    names = list(ref["env_reward_overview"].index)
    assert names == ["CartPole-v0_7_CCCCCC.pt", "CartPole-v0_4_QQQQQQ.pt"]                      # sorted by the last nine characters
    model_dir = tmp_path / "models"
    model_dir.mkdir()
    vary_of = {n: True for n in names}
    for f in names:
        (model_dir / f).write_bytes(b"")
    it = iter(range(100))

    def harness(train_env, test_env, config, agents_num):
        from learning_environments_amd.experiments.syn_env_evaluate import apply_comparability_settings
        config["agents"].setdefault("ddqn", {})
        apply_comparability_settings(config)
        k = next(it)
        return ([[float(v) for v in ref["reward_list"][2 * k + i]] for i in range(agents_num)], [list(ref["train_steps_needed"][2 * k + i]) for i in range(agents_num)],
                [list(ref["episode_length_needed"][2 * k + i]) for i in range(agents_num)])
    rv.run_vary_hp(2, "g13", 2, 2, str(model_dir), _fake_loader(vary_of), harness, "CartPole", device="cpu", out_dir=str(tmp_path))
    mine = torch.load(str(tmp_path / "2_g13.pt"), weights_only=False)
    assert list(mine) == list(ref)                                                               # same keys, same order
    for k in ("reward_list", "train_steps_needed", "episode_length_needed"):
        assert mine[k] == ref[k] and type(mine[k]) is type(ref[k]) and type(mine[k][0]) is type(ref[k][0])
    assert mine["env_reward_overview"].equals(ref["env_reward_overview"])
    a, b = mine["config"]["agents"]["ddqn"], ref["config"]["agents"]["ddqn"]
    for k in ("print_rate", "early_out_num", "train_episodes", "init_episodes", "test_episodes", "early_out_virtual_diff"):
        assert a[k] == b[k]
    assert mine["config"]["agents"]["ddqn_vary"]["vary_hp"] is True and ref["config"]["agents"]["ddqn_vary"]["vary_hp"] is True


def _stand_in_harness(train_env, test_env, config, agents_num):
    """deterministic in the model it is given (its file name), so that any dealing of the models over ranks must reproduce the same lists"""
    k = sum(ord(c) for c in str(train_env)) % 97
    return [[float(k + i)] * 3 for i in range(agents_num)], [[100 * k + i] for i in range(agents_num)], [[k] for _ in range(agents_num)]


def _rv_rank(rank, world, port, model_dir, out_dir, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from learning_environments_amd.experiments import syn_env_run_vary_hp as rv_
    vary_of = {f: True for f in os.listdir(model_dir)}
    out = rv_.run_vary_hp(2, "mr", 5, 2, model_dir, _fake_loader(vary_of), _stand_in_harness, "CartPole", device="cpu", out_dir=out_dir)
    q.put((rank, out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_deal_the_models_and_gather_the_same_lists(tmp_path):
    """SURVEY.md §8(e) for this row: the unit (a checkpoint's agents) is independent, so N ranks take the models round-robin, exchange the per-model
    lists once (all_gather_object) and rank 0 writes the file -- the lists and the file equal the single-process run's."""
    import torch.multiprocessing as mp
    model_dir = tmp_path / "models"
    model_dir.mkdir()
    for i, tag in enumerate(("QQQQQQ", "CCCCCC", "HHHHHH", "AAAAAA", "ZZZZZZ")):
        (model_dir / ("CartPole-v0_%d_%s.pt" % (i, tag))).write_bytes(b"")
    results = {}
    for world, port in ((1, 29731), (2, 29732)):
        out_dir = tmp_path / ("out%d" % world)
        out_dir.mkdir()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rv_rank, args=(r, world, port, str(model_dir), str(out_dir), q)) for r in range(world)]
        for p_ in procs:
            p_.start()
        got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
        results[world] = (got, torch.load(str(out_dir / "2_mr.pt"), weights_only=False))
        assert sorted(os.listdir(str(out_dir))) == ["2_mr.pt"]                                   # written once, by rank 0
    single, double = results[1][0][0][1], results[2][0]
    assert len(single[0]) == 10
    for rank, out in double:
        assert out == single                                                                     # every rank holds the full lists
    assert results[2][1]["reward_list"] == results[1][1]["reward_list"]
    assert results[2][1]["env_reward_overview"].equals(results[1][1]["env_reward_overview"])


# ------------------------------------------------------------------------------------------------------------------------------
# GPU half
# ------------------------------------------------------------------------------------------------------------------------------
def _write_models(tmp_path):
    """Three CartPole SE checkpoints in the reference's format: the reference-written one of G12 and two perturbed copies; the vary_hp
    flag of `ddqn_vary` (what get_all_files selects on) on for two, off for one."""
    from learning_environments_amd.experiments.syn_env_evaluate import load_envs_and_config
    src = os.path.join(HERE, "golden", "ckpt_cartpole_se_reference_b.pt")
    d = tmp_path / "models"
    d.mkdir()
    base = torch.load(src, map_location="cpu", weights_only=False)
    gen = torch.Generator().manual_seed(9)
    for name, vary, amp in (("CartPole-v0_4_QQQQQQ.pt", True, 0.0), ("CartPole-v0_7_CCCCCC.pt", True, 0.02), ("CartPole-v0_2_HHHHHH.pt", False, 0.03)):
        sd = {k: (v + amp * torch.randn(v.shape, generator=gen)) if v.dtype.is_floating_point else v for k, v in base["model"].items()}
        cfg = json.loads(json.dumps(base["config"]))
        cfg["agents"]["ddqn_vary"]["vary_hp"] = vary
        torch.save({"model": sd, "config": cfg}, str(d / name))
    return str(d), load_envs_and_config


@pytest.mark.gpu
@pytest.mark.parametrize("mode,lpt", [(2, False), (1, False), (0, False), (2, True), (0, True)])
def test_run_vary_hp_one_launch_equals_model_by_model_and_the_oracle(tmp_path, monkeypatch, mode, lpt):
    """lpt: the launch order of big launches (more chains than compute units: the expensive draws first) forced on for this small one --
    the returned lists stay in (model, agent) order."""
    from oracle import oracle as orc
    from learning_environments_amd.agents.nes_common import chain_keys
    from learning_environments_amd.experiments import syn_env_evaluate as se
    from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents
    if lpt:
        monkeypatch.setattr(se, "LPT_MIN_CHAINS", 4)
    model_dir, load = _write_models(tmp_path)
    model_num, agents_num = {2: 2, 1: 1, 0: 2}[mode], 3
    rewards, steps, episodes = rv.run_vary_hp(mode, "t", model_num, agents_num, model_dir, load, train_test_agents, "CartPole", out_dir=str(tmp_path))
    last = train_test_agents.last
    assert last["inner"].chains == model_num * agents_num and last["inner"].cfg.test_mode == 1          # ONE launch for all models
    assert len(rewards) == len(steps) == len(episodes) == model_num * agents_num
    fused_train = last["reward_train"]
    hps = last["hp"]
    assert (last["order"] is not None) == lpt
    if lpt:
        assert sorted(last["order"].tolist()) == list(range(model_num * agents_num))
        monkeypatch.setattr(se, "LPT_MIN_CHAINS", 256)
    slot = (lambda c_: c_) if not lpt else (lambda c_, inv=np.argsort(last["order"]): int(inv[c_]))       # chain of `inner` that ran pair c_
    fused_inner = last["inner"]
    # model by model through the same function (what a harness callable without `.fused` gets)
    if mode == 0:
        _, real_env, config = load(os.listdir(model_dir)[0], model_dir, "cuda")
        seq = [train_test_agents(real_env, real_env, config, agents_num, model_index=m) for m in range(model_num)]
        files = [None] * model_num
    else:
        files = rv.get_all_files(mode == 2, model_num, model_dir, load, "CartPole", "cuda")
        assert files == (["CartPole-v0_7_CCCCCC.pt", "CartPole-v0_4_QQQQQQ.pt"] if mode == 2 else ["CartPole-v0_2_HHHHHH.pt"])
        seq = []
        for m, f in enumerate(files):
            venv, real_env, config = load(f, model_dir, "cuda")
            # (mode 1 selects the checkpoints whose `vary_hp` is off; the harness function switches it on for its agents all the same,
            # syn_env_evaluate_cartpole_vary_hp_2.py:30 -- the reference's behaviour)
            seq.append(train_test_agents(venv, real_env, config, agents_num, model_index=m))
    assert [r for s_ in seq for r in s_[0]] == rewards and [r for s_ in seq for r in s_[1]] == steps and [r for s_ in seq for r in s_[2]] == episodes
    # one (model, agent) pair against the oracle chain: its model's weights, its key (seed 0, model index, agent index), its draw
    m, i = model_num - 1, agents_num - 1
    c = m * agents_num + i
    cfgd = json.loads(json.dumps(load(os.listdir(model_dir)[0] if mode == 0 else files[m], model_dir, "cuda")[2]))
    from learning_environments_amd.experiments.syn_env_evaluate import apply_comparability_settings
    apply_comparability_settings(cfgd)
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    over = dict(synthetic_env_type=1, reward_env_type=0) if mode == 0 else {}
    over.update(orc.hp_overrides(hps[c]))
    inner = fused_inner
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=inner.cfg.grad_chunk, rng_mode=0, test_mode=1, **over)
    key = int(chain_keys(0, m, np.array([i]), np.zeros(1, np.int64))[0])
    assert key == int(last["keys"][c])
    theta = np.zeros(1, np.float32) if mode == 0 else load(files[m], model_dir, "cuda")[0].env.flat_params().cpu().numpy()
    p_c = orc.mlp_num_params(orc.mlp_desc(4, ocfg.q_hidden, ocfg.q_layers, 2, ocfg.q_act))
    o = orc.ddqn_se_chain(ocfg, theta, inner.agent_init[slot(c)].cpu().numpy()[:p_c], rng_key=key)
    assert o["rc"] == 0
    assert rewards[c] == o["final_test_returns"].tolist() and steps[c] == [o["train_steps"]] and episodes[c] == [o["episodes_run"]]
    assert fused_train[c] == o["episode_test_mean"][:o["episodes_run"]].tolist()
    saved = torch.load(os.path.join(str(tmp_path), "%d_t.pt" % mode), weights_only=False)
    assert saved["reward_list"] == rewards and saved["env_reward_overview"].shape == (model_num, 10 * agents_num)


def test_cli_flags_are_the_scripts_flags():
    """`python -m learning_environments_amd.experiments.syn_env_run_vary_hp`: the flags of the scripts' __main__ (--mode, --pool, --agents_num,
    --model_num, --device) plus where the models are; an unknown agent is refused before any GPU work."""
    with pytest.raises(SystemExit):
        rv.main(["--model_dir", "/nonexistent", "--agent", "PPO"])
    with pytest.raises(SystemExit):
        rv.main([])                                            # --model_dir is required


@pytest.mark.gpu
def test_cli_runs_a_mode_end_to_end(tmp_path):
    model_dir, _ = _write_models(tmp_path)
    out = rv.main(["--model_dir", model_dir, "--mode", "2", "--agents_num", "2", "--model_num", "2", "--out_dir", str(tmp_path)])
    rewards, steps, episodes = out[2]
    assert len(rewards) == 4 and all(len(r) == 10 for r in rewards) and all(e[0] >= 21 for e in episodes)
    saved = torch.load(os.path.join(str(tmp_path), "2_ddqn_vary_transfer_reward_overview_2_agents_num_2_model_num.pt"), weights_only=False)
    assert saved["reward_list"] == rewards and saved["env_reward_overview"].shape == (2, 20)


def _rv_rank_gpu(rank, world, port, model_dir, out_dir, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from learning_environments_amd.experiments import syn_env_run_vary_hp as rv_
    from learning_environments_amd.experiments.syn_env_evaluate import load_envs_and_config, train_test_agents
    out = rv_.run_vary_hp(2, "mg", 2, 2, model_dir, load_envs_and_config, train_test_agents, "CartPole", out_dir=out_dir)
    q.put((rank, out, train_test_agents.last["inner"].chains))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_equal_the_single_process_launch(tmp_path):
    """The product's harness behind run_vary_hp with two ranks (both on this box's one GPU, gloo for the list exchange): each rank launches ITS
    model's agents (2 chains instead of 4), the gathered lists equal the one-process launch of all four bit for bit -- the chains are keyed by
    (seed, model index, agent index), not by where they run."""
    import torch.multiprocessing as mp
    model_dir, _ = _write_models(tmp_path)
    res = {}
    for world, port in ((1, 29741), (2, 29742)):
        out_dir = tmp_path / ("o%d" % world)
        out_dir.mkdir()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rv_rank_gpu, args=(r, world, port, model_dir, str(out_dir), q)) for r in range(world)]
        for p_ in procs:
            p_.start()
        got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
        for p_ in procs:
            p_.join(timeout=120)
            assert p_.exitcode == 0
        res[world] = got
    single = res[1][0]
    assert single[2] == 4 and [g_[2] for g_ in res[2]] == [2, 2]
    for g_ in res[2]:
        assert g_[1] == single[1]


@pytest.mark.gpu
def test_generalization_gap_script_runs_on_the_register_resident_kernel(tmp_path):
    """experiments/syn_env_evaluate_cartpole_vary_hp_2_eval_generalization_gap.py: vary_hp off + the fixed optimised DDQN hyper-parameters = the
    headline kernel's shape (4-57-2 tanh, batch 199) with test_mode 1: all models in one launch of `ddqn_se_inner_kernel`, one (model, agent)
    pair against the oracle chain."""
    from oracle import oracle as orc
    from learning_environments_amd.agents.nes_common import chain_keys, fresh_agent_init
    from learning_environments_amd.experiments.syn_env_evaluate import GENERALIZATION_GAP_HP, train_test_agents, train_test_agents_generalization_gap
    model_dir, load = _write_models(tmp_path)
    rewards, steps, episodes = rv.run_vary_hp(2, "gg", 2, 3, model_dir, load, train_test_agents_generalization_gap, "CartPole", out_dir=str(tmp_path))
    last = train_test_agents.last
    inner = last["inner"]
    assert inner.chains == 6 and not inner.dueling and inner.cfg.grad_chunk > 0 and inner.cfg.test_mode == 1
    assert (inner.cfg.batch_size, inner.cfg.q_hidden, inner.cfg.q_layers) == (199, 57, 1)
    files = rv.get_all_files(True, 2, model_dir, load, "CartPole", "cuda")
    m, i = 1, 2
    c = m * 3 + i
    venv, _, cfgd = load(files[m], model_dir, "cuda")
    from learning_environments_amd.experiments.syn_env_evaluate import apply_comparability_settings
    apply_comparability_settings(cfgd)
    cfgd["agents"]["ddqn"].update(GENERALIZATION_GAP_HP)
    cfgd["agents"]["gtn"]["agent_name"] = "DDQN"
    ocfg = orc.ddqn_cfg_from_config(json.loads(json.dumps(cfgd)), grad_chunk=inner.cfg.grad_chunk, rng_mode=0, test_mode=1)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0 + 1000003 * m)
    init = fresh_agent_init(last["task"].agent_bounds, 3, gen, torch.device("cuda")).cpu().numpy()[i]
    key = int(chain_keys(0, m, np.array([i]), np.zeros(1, np.int64))[0])
    o = orc.ddqn_se_chain(ocfg, venv.env.flat_params().cpu().numpy(), init, rng_key=key)
    assert o["rc"] == 0
    assert rewards[c] == o["final_test_returns"].tolist() and steps[c] == [o["train_steps"]] and episodes[c] == [o["episodes_run"]]
