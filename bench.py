#!/usr/bin/env python3
"""bench.py -- NES worker-evaluations/sec of the fused MI355X inner loop (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one NES generation of GTN_Master: noise draw, ONE fused-kernel launch evaluating this rank's share of the
population (3 inner loops per worker: train DDQN on the perturbed CartPole SE with per-episode real-env tests + final
test), mirrored-sampling pick, one all-gather of the fitness triples, rank transform + theta update.
Workload = BASELINE configs[1] ("CartPole-v0 SE, DDQN inner agent, NES pop=64 on 1xMI355X") in the fixed-work form of
BASELINE.md §3: theta = torch default Linear init under seed 0 with done-net output bias -10, early-out disabled,
train_episodes=20.  Weak scaling: every GPU evaluates 64 workers (global population 64*N).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POP_PER_GPU = 64
TRAIN_EPISODES = 20
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_master(world):
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=POP_PER_GPU * world), TRAIN_EPISODES)
    torch.manual_seed(0)                      # theta: torch default Linear init under seed 0
    cwd = os.getcwd()
    os.makedirs("/tmp/lenv_bench", exist_ok=True)
    os.chdir("/tmp/lenv_bench")               # GTN_Base creates ./results/GTN_sync relative to cwd
    try:
        master = GTN_Master(cfg, bohb_id=0, seed=1234)
    finally:
        os.chdir(cwd)
    with torch.no_grad():
        master.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)   # SE never terminates: fixed work per episode
    return master, cfg


def algorithmic_bytes(master, stats):
    """SURVEY.md §8(d) algorithmic byte model for the fused kernel (per launch = this rank's chains):
    chain setup 4*2*P_theta; per train env-step 4*[(2S+ad+2) + (A+S) + (S+2)]; per learn step additionally
    4*[B*(2S+ad+2) + 8*P_agent]; per test env-step 4*(P_act + 2S + 2)."""
    c = master.cfg
    S, A, B = c.state_dim, c.num_actions, c.batch_size
    row = 2 * S + 1 + 2
    p_agent = master.inner.p_agent
    chains = stats.shape[0]
    train_steps, learn_steps, test_steps = int(stats[:, 1].sum()), int(stats[:, 2].sum()), int(stats[:, 3].sum())
    b = chains * 4 * 2 * master.p_theta
    b += train_steps * 4 * (row + (A + S) + (S + 2))
    b += learn_steps * 4 * (B * row + 8 * p_agent)
    b += test_steps * 4 * (p_agent + 2 * S + 2)
    return b, train_steps, learn_steps, test_steps


def measured_traffic():
    """HBM bytes per fused-kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_summary.json, written by tools/summarize_profiles.py: 2*FETCH_SIZE + WRITE_SIZE, KB -> bytes, with the
    gfx950 read-side correction of MI355X_MICROARCH.md).  PMC counters cannot be read from inside the run, hence the file."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(f))
            if "hbm_traffic_bytes_per_launch" in d:
                best = (d["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def cpu_baseline(master, cfgd):
    """The oracle (CPU port of the same path, oracle/lenv_oracle.c) timed on this box's host cores on a bounded sample
    of the same workload: `pop_s` workers (3 chains each, same theta/eps/agent-init recipe), one thread per core."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 256)
    pop_s = max(2, min(64, threads // 3 if threads >= 6 else 2))
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=master.cfg.grad_chunk, rng_mode=0)
    theta = master.theta.detach().cpu().numpy()
    eps = (np.random.RandomState(1).randn(pop_s, theta.size) * cfgd["agents"]["gtn"]["noise_std"]).astype(np.float32)
    bounds = master.agent_bounds.cpu().numpy()
    init = ((np.random.RandomState(2).rand(3 * pop_s, bounds.size).astype(np.float32) * 2 - 1) * bounds).astype(np.float32)
    t0 = time.time()
    orc.ddqn_se_population(ocfg, theta, eps, init, seed=1234, generation=0, threads=threads)
    dt = time.time() - t0
    return {"value": pop_s / dt, "unit": "worker-evaluations/s", "cores": threads, "kind": "port",
            "sample": "%d workers (=%d chains) of the same fixed-work CartPole-SE/DDQN workload, %d threads, %.1f s wall"
                      % (pop_s, 3 * pop_s, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); the product path has no CPU fallback")
    # LENV_BENCH_BACKEND=gloo is a plumbing check only (several ranks sharing one GPU on a 1-GPU box); the product path is RCCL
    backend = os.environ.get("LENV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    master, cfgd = build_master(world)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        master.step(it)
        it += 1
    barrier()
    # HIP events around every fused-kernel launch (same stream the kernel is enqueued on = torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    orig_inner = master.engine.inner_scores
    slot = {"i": 0}

    def timed_inner(*a, **k):
        e0, e1 = ev[slot["i"]]
        e0.record()
        out = orig_inner(*a, **k)
        e1.record()
        slot["i"] += 1
        return out

    master.engine.inner_scores = timed_inner
    t0 = time.perf_counter()
    for _ in range(args.steps):
        master.step(it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    master.engine.inner_scores = orig_inner

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        stats = master.inner.stats.cpu().numpy()
        bytes_launch, train_steps, learn_steps, test_steps = algorithmic_bytes(master, stats)
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        total_evals = POP_PER_GPU * world * args.steps
        line = {
            "metric": "NES worker-evaluations/sec (full inner-loop train+eval) at pop=64 per GPU",
            "value": total_evals / dt, "unit": "worker-evaluations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CartPole-v0 SE (6-83-{4,1,1} leakyrelu) + DDQN (4-57-2 tanh, B=199), "
                                   "NES pop=64 per GPU, fixed-work: train_episodes=%d x 200 steps, 10 real-env test episodes "
                                   "per train episode + final test, early-out off" % TRAIN_EPISODES,
                       "pop_per_gpu": POP_PER_GPU, "global_pop": POP_PER_GPU * world, "chains_per_gpu": 3 * POP_PER_GPU,
                       "train_episodes": TRAIN_EPISODES, "parallelism": "population-sharded x%d, 1 all-gather/generation" % world,
                       "env_steps_per_s": (train_steps + test_steps) * world / (dt / args.steps),
                       "kernel_launches_per_generation": 1 + 3},
            "roofline": {"bound": "hbm", "kernel": "ddqn_se_inner_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_launch": bytes_launch, "kernel_ms": kernel_ms,
                         "note": "latency/issue-bound small-MLP chains: weights+activations live in LDS, only the replay "
                                 "buffer touches HBM/L2 (see DESIGN.md)"},
        }
        tr = measured_traffic()
        if tr is not None:
            line["roofline"]["traffic"] = tr[0]
            line["roofline"]["traffic_source"] = tr[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(master, cfgd)
        print(json.dumps(line))

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
