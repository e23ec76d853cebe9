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
