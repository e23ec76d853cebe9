"""world_size-2 gloo test of the N>1 path: population sharding, the single all-gather, redundant rank update.
The per-chain scorer is the CPU oracle (oracle/engine_standin.py); on the GPU box the same GTN_Master code drives the
HIP engine over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_config(num_workers):
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=num_workers, max_iterations=2), 2)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 12
    cfg["agents"]["ddqn"]["test_episodes"] = 3
    cfg["agents"]["ddqn"]["batch_size"] = 32
    return cfg


def _vary_config(num_workers):
    from learning_environments_amd.configs import with_vary
    cfg = with_vary(_small_config(num_workers))
    cfg["agents"]["ddqn"].update(batch_size=24, hidden_size=16, init_episodes=1)
    cfg["agents"]["gtn"]["max_iterations"] = 1
    return cfg


def _run(rank, world, port, tmp, q, seed_per_rank=False, vary=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.chdir(tmp)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from oracle.engine_standin import OracleNesEngine
    from learning_environments_amd.agents.GTN import GTN_Master
    # a launcher that seeds every rank differently must not matter: rank 0's theta / model name are broadcast at construction
    torch.manual_seed(1000 * rank if seed_per_rank else 0)
    import random
    random.seed(rank if seed_per_rank else 0)
    m = GTN_Master(_vary_config(4) if vary else _small_config(5), bohb_id=0, engine=OracleNesEngine(), seed=11)
    with torch.no_grad():
        m.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)
    mean_score, mean_list, _ = m.run()
    q.put((rank, m.theta.numpy().copy(), list(m.score_list), list(m.score_orig_list), float(mean_score), (m.w_lo, m.w_hi),
           os.path.basename(m.model_name)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, tmp_path, port, seed_per_rank=False, vary=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, str(tmp_path), q, seed_per_rank, vary)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.timeout(400)
def test_two_rank_run_matches_single_rank(tmp_path):
    (tmp_path / "w1").mkdir()
    (tmp_path / "w2").mkdir()
    single = _launch(1, tmp_path / "w1", 29611)[0]
    double = _launch(2, tmp_path / "w2", 29612)
    # uneven split of 5 workers over 2 ranks: [0,3) and [3,5)
    assert double[0][5] == (0, 3) and double[1][5] == (3, 5)
    for r in double:
        # every rank ends with bit-identical theta and fitness lists == the single-process run
        assert np.array_equal(r[1], single[1])
        assert r[2] == single[2] and r[3] == single[3] and r[4] == single[4]
    assert not np.array_equal(single[1], np.zeros_like(single[1]))


@pytest.mark.timeout(400)
def test_vary_agents_two_ranks_match_single_rank(tmp_path):
    """DDQN_vary through GTN_Master on CPU tensors (oracle-backed engine): every chain's hyper-parameter draw and fresh agent
    are functions of its (seed, generation, worker, kind) key, so the sharded run equals the single-rank run bit for bit."""
    (tmp_path / "v1").mkdir(); (tmp_path / "v2").mkdir()
    one = _launch(1, tmp_path / "v1", 29671, vary=True)[0]
    two = _launch(2, tmp_path / "v2", 29672, vary=True)
    for r in two:
        assert np.array_equal(r[1], one[1])                    # theta after the generation
        assert r[2] == one[2] and r[3] == one[3]               # score lists
    assert two[0][5] != two[1][5]                              # different shards
    # the draws really differ between chains (otherwise this would be the plain-agent test again)
    sys.path.insert(0, ROOT)
    from learning_environments_amd.agents import vary
    from oracle import oracle as orc
    hps = {tuple(sorted(vary.vary_hyperparameters(_vary_config(4)["agents"]["ddqn"], vary.chain_units(orc.chain_key(11, 0, w, k))).items()))
           for w in range(4) for k in range(3)}
    assert len(hps) == 12


def test_ranks_seeded_differently_still_agree(tmp_path):
    """theta is replicated and never exchanged per generation, so it must start identical.  Ranks that build
    their initial theta under different torch / random seeds end with bit-identical theta, fitness lists and model name."""
    (tmp_path / "w").mkdir()
    double = _launch(2, tmp_path / "w", 29613, seed_per_rank=True)
    assert np.array_equal(double[0][1], double[1][1]) and double[0][2] == double[1][2] and double[0][3] == double[1][3]
    assert double[0][6] == double[1][6]


def _run_hip(rank, world, port, tmp, q, graph=None):
    """Same as _run but with the product's HIP engine; both ranks share cuda:0 and talk through gloo (plumbing check of the
    sharded path with the real kernels on a 1-GPU box -- the product path on a multi-GPU node is RCCL, one GPU per rank)."""
    sys.path.insert(0, ROOT)
    os.chdir(tmp)
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from learning_environments_amd.agents.GTN import GTN_Master
    torch.manual_seed(0)
    m = GTN_Master(_small_config(5), bohb_id=0, seed=11, graph=graph)
    with torch.no_grad():
        m.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)
    mean_score, mean_list, _ = m.run()
    q.put((rank, m.theta.cpu().numpy().copy(), list(m.score_list), list(m.score_orig_list), float(mean_score), (m.w_lo, m.w_hi),
           (bool(m.use_graph), int(m.graph_replays), len(mean_list))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_hip_engine_matches_single_rank(tmp_path):
    ctx = mp.get_context("spawn")

    def launch(world, sub, port, graph=None):
        (tmp_path / sub).mkdir()
        q = ctx.Queue()
        procs = [ctx.Process(target=_run_hip, args=(r, world, port, str(tmp_path / sub), q, graph)) for r in range(world)]
        for p in procs:
            p.start()
        out = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        return sorted(out, key=lambda t: t[0])

    single = launch(1, "g1", 29621)[0]
    double = launch(2, "g2", 29622)
    eager = launch(2, "g3", 29623, graph=False)
    assert double[0][5] == (0, 3) and double[1][5] == (3, 5)
    # the captured paths really ran: one graph per generation in one process, two (around the eager all-gather) with two ranks
    used, replays, gens = single[6]
    assert used and replays == gens
    for r in double:
        assert r[6][0] and r[6][1] == 2 * r[6][2]
    for r in eager:
        assert not r[6][0] and r[6][1] == 0
    for r in double + eager:
        assert np.array_equal(r[1], single[1])
        assert r[2] == single[2] and r[3] == single[3] and r[4] == single[4]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_rank_rccl_group_runs_the_collective_path(tmp_path):
    """RCCL on the hardware a one-GPU box has: a child process creates a ONE-RANK `nccl` (= RCCL) process group
    before any other GPU call and runs two GTN_Master generations.  With a group the fitness records go through
    dist.all_gather_into_tensor (float64 device tensor) and the captured generation is the two graphs around that collective -- the
    path N ranks run; theta and the score lists must equal, bit for bit, the child that has no process group (one graph, no collective).
    The eager (graph=False) variant of the grouped run is checked too."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for name, group, graph, port in (("plain", 0, 1, 0), ("rccl", 1, 1, 29631), ("rccl_eager", 1, 0, 29632)):
        work = tmp_path / name
        work.mkdir()
        out = str(tmp_path / (name + ".npz"))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_child.py"), "--group", str(group), "--graph", str(graph),
                            "--port", str(port or 29630), "--out", out, "--workdir", str(work)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        outs[name] = np.load(out)
    plain, rccl, eager = outs["plain"], outs["rccl"], outs["rccl_eager"]
    assert int(plain["has_group"]) == 0 and int(plain["collectives"]) == 0 and int(plain["two_graphs"]) == 0
    gens = len(plain["mean"])
    assert gens == 2
    for r in (rccl, eager):
        assert int(r["has_group"]) == 1 and str(r["backend"]) == "nccl"
        assert int(r["collectives"]) == gens                      # ONE all-gather per generation, really issued
        for k in ("theta", "score", "score_orig", "mean"):
            assert np.array_equal(r[k], plain[k]), k
    # the captured path: two graphs per generation around the eager collective (unless this stack refused the capture, which the
    # master survives by running eagerly -- then it must say why)
    if int(rccl["use_graph"]):
        assert int(rccl["two_graphs"]) == 1 and int(rccl["replays"]) == 2 * gens
    else:
        assert str(rccl["capture_error"]) != ""
    assert int(eager["use_graph"]) == 0 and int(eager["replays"]) == 0
    assert int(plain["use_graph"]) == 1 and int(plain["replays"]) == gens
