"""The master side of the sync-file transport (reference agents/GTN_master.py:147-195) against file-based workers.

CPU: GTN_Master(transport="file") with the oracle stand-in engine drives oracle/file_worker.py processes through
results/GTN_sync; the theta it ends with must equal the update recomputed from the workers' own result payloads.
GPU (-m gpu): the same master drives this package's HIP GTN_Worker for BASELINE configs 2, 4 and 5."""
import copy
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_cartpole(num_workers, iterations):
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=num_workers, max_iterations=iterations), 2)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 12
    cfg["agents"]["ddqn"].update(test_episodes=3, batch_size=32)
    cfg["agents"]["gtn"].update(mode="single", time_sleep_master=0.05, time_sleep_worker=0.1)
    return cfg


@pytest.mark.timeout(300)
def test_file_master_drives_oracle_workers(tmp_path, monkeypatch):
    from oracle import oracle as orc
    from oracle.engine_standin import OracleNesEngine
    from oracle.file_worker import flat_linear
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    W, iters = 3, 2
    cfg = _tiny_cartpole(W, iters)
    cfg["device"] = "cpu"
    torch.manual_seed(3)
    master = GTN_Master(cfg, bohb_id=-1, engine=OracleNesEngine(), transport="file")
    master.clean_working_dir()
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.file_worker", str(i), "--initial-sleep", "0.01"], env=env,
                              cwd=str(tmp_path)) for i in range(W)]
    try:
        seen = []
        orig_read = master.read_worker_results

        def spy():
            # snapshot the raw result payloads before the master consumes (and deletes) them
            import time
            while not all(os.path.isfile(master.get_result_check_file_name(i)) for i in range(W)):
                time.sleep(0.02)
            seen.append([torch.load(master.get_result_file_name(i)) for i in range(W)])
            orig_read()
        master.read_worker_results = spy
        theta0 = master.theta.clone().numpy()
        mean_score, mean_list, _ = master.run()
        assert len(mean_list) == iters and len(seen) == iters
        # recompute the NES update from the payloads: eps arrives sign-folded (GTN_worker.py:180-185), sign = +1
        theta = theta0
        for payload in seen:
            eps = np.stack([flat_linear(d["eps"]) for d in payload])
            w = orc.score_transform(cfg["agents"]["gtn"]["score_transform_type"], [d["score"] for d in payload],
                                    [d["score_orig"] for d in payload])
            theta = orc.update_env(theta, eps, np.ones(W, np.float32), w, cfg["agents"]["gtn"]["step_size"])
        assert np.array_equal(master.theta.numpy(), theta)
        assert not np.array_equal(theta, theta0)
        assert master.score_list == [d["score"] for d in seen[-1]]
        assert mean_score == np.mean([d["score_orig"] for d in seen[-1]])
        for p in procs:                       # quit_flag on the last iteration (bohb_id < 0) ends the workers
            assert p.wait(timeout=60) == 0
        assert os.listdir(master.sync_dir) == []
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def _hip_worker_thread(id, cwd_config, errors):
    try:
        from learning_environments_amd.agents.GTN import GTN_Worker
        GTN_Worker(id, bohb_id=-1, seed=100 + id).run()
    except Exception as e:  # noqa
        errors.append(e)


def _configs():
    from learning_environments_amd import configs
    c2 = _tiny_cartpole(2, 2)
    c4 = configs.cliff_reward_env_ql(num_workers=2, max_iterations=2)
    c4["agents"]["ql"]["train_episodes"] = 20
    c4["agents"]["gtn"].update(mode="single", time_sleep_master=0.05, time_sleep_worker=0.1, quit_when_solved=False)
    c5 = configs.halfcheetah_reward_env_td3(num_workers=2, max_iterations=2)
    c5["agents"]["td3"].update(train_episodes=3, init_episodes=1, batch_size=32, hidden_size=32)
    c5["envs"]["HalfCheetah-v3"]["max_steps"] = 20
    c5["agents"]["gtn"].update(mode="single", time_sleep_master=0.05, time_sleep_worker=0.1, quit_when_solved=False)
    return {"cfg2_cartpole_ddqn": c2, "cfg4_cliff_ql": c4, "cfg5_cheetah_td3": c5}


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", ["cfg2_cartpole_ddqn", "cfg4_cliff_ql", "cfg5_cheetah_td3"])
def test_file_master_drives_hip_workers(tmp_path, monkeypatch, name):
    """This package's master <-> this package's GTN_Worker (HIP) through the sync directory, configs 2, 4 and 5."""
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.models.model_utils import linear_params
    monkeypatch.chdir(tmp_path)
    cfg = _configs()[name]
    W = cfg["agents"]["gtn"]["num_workers"]
    torch.manual_seed(5)
    master = GTN_Master(cfg, bohb_id=-1, transport="file")
    master.clean_working_dir()
    theta0 = master.theta.clone()
    errors = []
    threads = [threading.Thread(target=_hip_worker_thread, args=(i, None, errors), daemon=True) for i in range(W)]
    for t in threads:
        t.start()
    mean_score, mean_list, _ = master.run()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive()
    assert errors == []
    assert len(mean_list) == 2 and np.isfinite(mean_score)
    assert master.eps.shape == (W, master.p_theta) and bool(torch.isfinite(master.theta).all())
    moved = not torch.equal(master.theta, theta0)
    assert moved or all(w == 0 for w in master.get_score_transform_list())
    # the env wrapper the master saves sees the updated parameters (flat-buffer aliasing)
    flat = torch.cat([p.detach().reshape(-1) for p in linear_params(master.synthetic_env_orig)])
    assert torch.equal(flat, master.theta)


def test_score_transform_and_update_env_keep_the_mirrored_sign(tmp_path, monkeypatch):
    """the reference-compatible call sequence `score_transform(); update_env()` (no arguments) must apply the
    eps of a worker whose -eps scored better with sign -1 (in the reference eps_list already holds the inverted eps,
    GTN_worker.py:180-185).  Compared against the update recomputed by the oracle with the signs of the generation."""
    from oracle import oracle as orc
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    cfg = _tiny_cartpole(4, 1)
    cfg["device"] = "cpu"
    torch.manual_seed(7)
    m = GTN_Master(cfg, bohb_id=0, engine=OracleNesEngine(), seed=3)
    theta0 = m.theta.clone().numpy()
    gathered = m.evaluate_population(0)
    m._gathered = gathered
    host = gathered.numpy()
    m.score_list, m.score_orig_list = host[:, 0].tolist(), host[:, 1].tolist()
    assert (host[:, 2] == -1).any() or (host[:, 2] == 1).all()        # at least exercised; the seed below gives both signs
    m.score_transform()
    m.update_env()
    w = orc.score_transform(cfg["agents"]["gtn"]["score_transform_type"], host[:, 0], host[:, 1])
    want = orc.update_env(theta0, m.eps.numpy(), host[:, 2].astype(np.float32), w, cfg["agents"]["gtn"]["step_size"])
    assert np.array_equal(m.theta.numpy(), want)
    assert m.score_transform_list == w.tolist()
    # editing the public score lists is honoured, the signs stay
    m.theta.copy_(torch.from_numpy(theta0))
    m.score_list = [s + 1.0 for s in m.score_list]
    m.update_env()
    w2 = orc.score_transform(cfg["agents"]["gtn"]["score_transform_type"], host[:, 0] + 1.0, host[:, 1])
    assert np.array_equal(m.theta.numpy(), orc.update_env(theta0, m.eps.numpy(), host[:, 2].astype(np.float32), w2,
                                                          cfg["agents"]["gtn"]["step_size"]))


# ---- G11: the payload files the REFERENCE wrote (oracle/gen_golden.py gen_g11; reference agents/GTN_master.py:147-176 write_worker_inputs,
# agents/GTN_worker.py:139-154 write_worker_result) meet this package's worker and master ----
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _flat(envw):
    from learning_environments_amd.models.model_utils import linear_params
    return torch.cat([p.detach().reshape(-1) for p in linear_params(envw)]).cpu().numpy()


def _place(src, dst, check):
    import shutil
    shutil.copyfile(os.path.join(GOLDEN, src), dst)
    torch.save({}, check)


def _g11():
    return np.load(os.path.join(GOLDEN, "g11_file_transport.npz"))


def _sd_signature(sd):
    return [(k, tuple(v.shape), v.dtype) for k, v in sd.items()]


def _check_worker_reads_reference_input(engine, tmp_path, monkeypatch):
    from learning_environments_amd.agents.GTN import GTN_Worker
    monkeypatch.chdir(tmp_path)
    g = _g11()
    ref_in = torch.load(os.path.join(GOLDEN, "g11_ref_master_input_w1.pt"))
    w = GTN_Worker(1, bohb_id=-1, engine=engine, seed=4)
    _place("g11_ref_master_input_w1.pt", w.get_input_file_name(1), w.get_input_check_file_name(1))
    w.time_sleep_worker = 0.0
    w.read_worker_input()
    assert w.timeout == float(g["timeout"]) == ref_in["timeout"] and w.quit_flag is True and w.config == ref_in["config"]
    assert np.array_equal(_flat(w.synthetic_env_orig), g["theta0"]) and np.array_equal(_flat(w.synthetic_env), g["theta0"])
    assert not os.path.exists(w.get_input_file_name(1)) and not os.path.exists(w.get_input_check_file_name(1))
    # ... and answers with a result payload of the reference's shape: same keys, same state-dict keys / shapes / dtypes
    best, orig = w.evaluate()
    w.write_worker_result(score=best, score_orig=orig, time_elapsed=0.5)
    mine = torch.load(w.get_result_file_name(1))
    ref_out = torch.load(os.path.join(GOLDEN, "g11_ref_worker_result_w1.pt"))
    assert list(mine.keys()) == list(ref_out.keys())
    for k in ("eps", "synthetic_env"):
        assert _sd_signature(mine[k]) == _sd_signature(ref_out[k])
    assert all(type(mine[k]) is type(ref_out[k]) for k in ("time_elapsed", "score", "score_orig"))
    assert os.path.isfile(w.get_result_check_file_name(1))
    # the eps it reports is the one its synthetic_env carries: synthetic_env = theta0 + eps (mirrored pick folded in)
    assert np.array_equal(_flat_sd(mine["synthetic_env"]), (torch.from_numpy(g["theta0"]) + torch.from_numpy(_flat_sd(mine["eps"]))).numpy())


def _flat_sd(sd):
    return np.concatenate([v.reshape(-1).numpy() for k, v in sd.items() if (k[:-6] + "bias") in sd or k.endswith("bias")])


def _check_master_reads_reference_results(engine, device, tmp_path, monkeypatch):
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    g = _g11()
    cfg = copy.deepcopy(torch.load(os.path.join(GOLDEN, "g11_ref_master_input_w0.pt"))["config"])
    cfg["device"] = device
    cfg["agents"]["gtn"]["time_sleep_master"] = 0.0
    m = GTN_Master(cfg, bohb_id=-1, engine=engine, transport="file")
    m.clean_working_dir()
    m.theta.copy_(torch.from_numpy(g["theta0"]).to(m.theta.device))
    # (a) what this master writes for its workers has the reference master's format ...
    m.write_worker_inputs(0)
    mine = torch.load(m.get_input_file_name(0))
    ref_in = torch.load(os.path.join(GOLDEN, "g11_ref_master_input_w0.pt"))
    assert list(mine.keys()) == list(ref_in.keys()) and mine["quit_flag"] is True and mine["timeout"] == ref_in["timeout"]
    assert _sd_signature(mine["synthetic_env_orig"]) == _sd_signature(ref_in["synthetic_env_orig"])
    assert all(torch.equal(mine["synthetic_env_orig"][k], ref_in["synthetic_env_orig"][k]) for k in ref_in["synthetic_env_orig"])
    m.clean_working_dir()
    # (b) ... and from the reference WORKERS' result payloads it reproduces the reference master's lists and its theta after
    # score_transform + update_env bit for bit
    for i in range(2):
        _place("g11_ref_worker_result_w%d.pt" % i, m.get_result_file_name(i), m.get_result_check_file_name(i))
    m.read_worker_results()
    assert m.score_list == g["score"].tolist() and m.score_orig_list == g["score_orig"].tolist()
    assert m.time_elapsed_list == g["time_elapsed"].tolist()
    assert np.array_equal(m.eps.cpu().numpy(), g["eps"])
    assert os.listdir(m.sync_dir) == []
    m.score_transform()
    assert m.score_transform_list == g["weights"].tolist()
    m.update_env()
    assert np.array_equal(m.theta.cpu().numpy(), g["theta1"])
    assert np.array_equal(_flat(m.synthetic_env_orig), g["theta1"])
    assert m.calc_worker_timeout() == float(np.mean(g["time_elapsed"])) * cfg["agents"]["gtn"]["time_mult"]


def test_master_reads_reference_worker_results_cpu(tmp_path, monkeypatch):
    from oracle.engine_standin import OracleNesEngine
    _check_master_reads_reference_results(OracleNesEngine(), "cpu", tmp_path, monkeypatch)


def test_worker_reads_reference_master_input_cpu(tmp_path, monkeypatch):
    from oracle.engine_standin import OracleNesEngine
    _check_worker_reads_reference_input(OracleNesEngine(), tmp_path, monkeypatch)


@pytest.mark.gpu
def test_master_reads_reference_worker_results_hip(tmp_path, monkeypatch):
    from learning_environments_amd.engine import HipNesEngine
    _check_master_reads_reference_results(HipNesEngine(), "cuda", tmp_path, monkeypatch)


@pytest.mark.gpu
def test_worker_reads_reference_master_input_hip(tmp_path, monkeypatch):
    from learning_environments_amd.engine import HipNesEngine
    _check_worker_reads_reference_input(HipNesEngine(), tmp_path, monkeypatch)


# ---- worker process entry points (reference experiments/GTN_Worker.py:13-16, experiments/GTN_Worker_single_pc.py:12-30) ----
def test_worker_cli_argument_plumbing(monkeypatch, capsys):
    from learning_environments_amd.experiments import GTN_Worker as cli, GTN_Worker_single_pc as cli_n
    import learning_environments_amd.agents.GTN as gtn
    made = []

    class FakeWorker(object):
        def __init__(self, id, bohb_id=-1, seed=None):
            made.append((bohb_id, id, seed))

        def run(self):
            made.append("ran")
    monkeypatch.setattr(gtn, "GTN_Worker", FakeWorker)
    assert cli.main(["20003", "7"]) == 0
    assert made == [(20003, 7, None), "ran"]
    assert capsys.readouterr().out.split() == ["20003", "7"]          # the reference echoes its arguments
    with pytest.raises(ValueError):
        cli.main(["0", "-1"])
    with pytest.raises(SystemExit):
        cli.main(["0"])                                                # both ids are required, as in the reference
    a = cli_n.parse_args([])
    assert (a.num_workers, a.bohb_id) == (16, 0)                       # the reference's single-PC defaults
    cmd = cli_n.worker_command(3, 5, seed=100)
    assert cmd[1:] == ["-m", "learning_environments_amd.experiments.GTN_Worker", "3", "5", "--seed", "105"]


def _wait_for_lines(stream, needle, count, timeout_s=180):
    """read `stream` until `count` lines containing `needle` have gone by"""
    import time
    seen, t0 = 0, time.time()
    while seen < count:
        line = stream.readline()
        if not line:
            if time.time() - t0 > timeout_s:
                raise TimeoutError("workers did not start")
            time.sleep(0.05)
            continue
        seen += needle in line
    return seen


@pytest.mark.timeout(300)
def test_worker_cli_fails_loudly_without_a_device(tmp_path, monkeypatch):
    """No CPU fallback behind the command line either: the process reads the master's input and dies on the first use of the engine."""
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    cfg = _tiny_cartpole(1, 1)
    cfg["device"] = "cpu"
    master = GTN_Master(cfg, bohb_id=-1, engine=OracleNesEngine(), transport="file")
    master.clean_working_dir()
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.Popen([sys.executable, "-m", "learning_environments_amd.experiments.GTN_Worker", "-1", "0"], env=env, cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        _wait_for_lines(p.stdout, "Starting GTN Worker", 1)
        master.write_worker_inputs(0)
        _, err = p.communicate(timeout=120)
        assert p.returncode != 0
        assert "needs a HIP device" in err and "no CPU fallback" in err
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_file_master_drives_worker_processes_started_by_the_single_pc_launcher(tmp_path, monkeypatch):
    """`python -m learning_environments_amd.experiments.GTN_Worker_single_pc 2`: two HIP worker PROCESSES that share the GPU serve this
    package's file-transport master for two generations and end on its quit_flag."""
    from learning_environments_amd.agents.GTN import GTN_Master
    monkeypatch.chdir(tmp_path)
    cfg = _tiny_cartpole(2, 2)
    torch.manual_seed(5)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    launcher = subprocess.Popen([sys.executable, "-m", "learning_environments_amd.experiments.GTN_Worker_single_pc", "2", "--bohb-id", "-1", "--seed", "100"],
                                env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, text=True)
    try:
        _wait_for_lines(launcher.stdout, "Starting GTN Worker", 2)
        master = GTN_Master(cfg, bohb_id=-1, transport="file")
        theta0 = master.theta.clone()
        mean_score, mean_list, _ = master.run()
        assert launcher.wait(timeout=120) == 0
        assert len(mean_list) == 2 and np.isfinite(mean_score)
        assert bool(torch.isfinite(master.theta).all())
        assert not torch.equal(master.theta, theta0) or all(w == 0 for w in master.get_score_transform_list())
        assert os.listdir(master.sync_dir) == []
    finally:
        if launcher.poll() is None:
            launcher.terminate()


def test_worker_calc_best_score_matches_the_reference_fixture(tmp_path, monkeypatch, golden):
    """GTN_Worker.calc_best_score (reference agents/GTN_worker.py:234-254) through the engine's worker-best routine: scores equal to the
    reference's (fixture G6M: num_grad_evals 3, 'mean' / 'minmax', mirrored or not) and the reference's side effects -- eps inverted
    when -eps won, synthetic_env = theta + eps afterwards."""
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Worker
    from learning_environments_amd.models.model_utils import linear_params
    monkeypatch.chdir(tmp_path)
    g = golden("g6m_worker_best_multi")
    cfg = _tiny_cartpole(1, 1)
    cfg["device"] = "cpu"
    w = GTN_Worker(0, bohb_id=-1, engine=OracleNesEngine(), seed=1)
    w.late_init(cfg)
    assert w.time_sleep_worker == pytest.approx(0.01)              # mode 'single': a tenth of the configured 0.1
    flat = lambda e: torch.cat([p.detach().reshape(-1) for p in linear_params(e)])
    for gt in ("mean", "minmax"):
        for m in (1, 0):
            w.grad_eval_type, w.mirrored_sampling = gt, bool(m)
            for row in range(g["score_add"].shape[0]):
                w.get_random_noise()
                eps0 = flat(w.eps).clone()
                w.subtract_noise_from_synthetic_env()          # the state run() is in before the pick
                best = w.calc_best_score(score_sub=list(g["score_sub"][row]), score_add=list(g["score_add"][row]))
                assert best == g["best_%s_%d" % (gt, m)][row]
                sign = float(g["sign_%s_%d" % (gt, m)][row])
                assert torch.equal(flat(w.eps), sign * eps0)
                assert torch.equal(flat(w.synthetic_env), flat(w.synthetic_env_orig) + flat(w.eps))
    w.grad_eval_type = "median"
    with pytest.raises(NotImplementedError):
        w.calc_best_score(score_sub=[1.0], score_add=[2.0])
    bad = _tiny_cartpole(1, 1)
    bad["agents"]["gtn"]["synthetic_env_type"] = 2
    with pytest.raises(NotImplementedError):
        GTN_Worker(1, bohb_id=-1, engine=OracleNesEngine(), seed=1).late_init(bad)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_worker_relaunches_a_team_launch_that_could_not_assemble(tmp_path, monkeypatch):
    """several worker processes share one GPU, and a worker's three-chain TD3 launch picks teams of six.  Next to a foreign
    kernel that holds most CUs the teams cannot assemble, the launch reports status -10, and GTN_Worker._run_chains repeats it with one
    workgroup per chain instead of raising: the evaluation's scores equal those of an undisturbed worker with the same seed."""
    import time
    from learning_environments_amd import configs
    from learning_environments_amd.agents.GTN import GTN_Worker
    from tools import diag
    monkeypatch.chdir(tmp_path)
    cfg = configs.fixed_work(configs.halfcheetah_reward_env_td3(num_workers=1, max_iterations=1), 2)
    cfg["agents"]["td3"]["init_episodes"] = 1
    cfg["envs"]["HalfCheetah-v3"]["max_steps"] = 40
    cfg["agents"]["gtn"].update(mode="single", time_sleep_worker=0.1)

    def evaluate(disturb):
        w = GTN_Worker(0, bohb_id=-1, seed=321)
        w.late_init(cfg)
        torch.manual_seed(9)                       # the noise draw of get_random_noise
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        if disturb:
            with torch.cuda.stream(side):          # 248 of 256 CUs for 1.5 s: a few team members start, most cannot
                diag.occupy_cus(248, 150 * 1024, 150_000_000, side.cuda_stream)      # tools/diag: test aid, not in the product ABI
            time.sleep(0.05)
        out = w.evaluate()
        side.synchronize()
        return out, w.team_fallbacks, int(getattr(next(iter(w._inner.values())).cfg, "team_size", -1))

    ref, fb0, ts0 = evaluate(False)
    got, fb1, ts1 = evaluate(True)
    assert fb0 == 0 and ts0 == 0
    assert got == ref                               # (score_best, score_orig): deterministic functions of the seed either way
    assert fb1 in (0, 1) and ts1 == (1 if fb1 else 0)


def test_single_pc_launcher_ends_with_the_first_failing_worker(monkeypatch, tmp_path):
    """the launcher polls all its children -- a worker that exits non-zero while worker 0 is still waiting for the master ends
    the launcher with that code (and the surviving worker is terminated) instead of blocking behind a sequential wait."""
    import time
    from learning_environments_amd.experiments import GTN_Worker_single_pc as cli_n
    monkeypatch.chdir(tmp_path)
    marker = tmp_path / "sleeper_started"
    cmds = {0: [sys.executable, "-c", "import time, pathlib; pathlib.Path(%r).write_text('x'); time.sleep(120)" % str(marker)],
            1: [sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(7)"]}
    monkeypatch.setattr(cli_n, "worker_command", lambda bohb_id, id, seed=None: cmds[id])
    t0 = time.time()
    rc = cli_n.main(["2", "--bohb-id", "-1"])
    assert rc == 7 and time.time() - t0 < 30
    assert marker.exists()                      # worker 0 was running (and has been terminated: main() returned long before its sleep ends)
