#!/usr/bin/env python3
"""bench.py -- NES worker-evaluations/sec of the fused MI355X inner loop (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one NES generation of GTN_Master: noise draw, ONE fused-kernel launch evaluating this rank's share of the
population (3 inner loops per worker: train DDQN on the perturbed CartPole SE with per-episode real-env tests + final
test), mirrored-sampling pick, one all-gather of the fitness records, rank transform + theta update.
Workload = BASELINE configs[1] ("CartPole-v0 SE, DDQN inner agent, NES pop=64 on 1xMI355X") in the fixed-work form of
BASELINE.md §3: theta = torch default Linear init under seed 0 with done-net output bias -10, early-out disabled,
train_episodes=20.

Multi-GPU: one process per GPU over RCCL.  Under torchrun the RANK/LOCAL_RANK/WORLD_SIZE environment is used as is; a
bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts the N ranks itself as CHILD processes -- the launcher parent
never touches the GPU, nothing re-execs.  With N > 1 two population layouts are timed back to back, K steps each:
    weak   : 64 workers per GPU (global population 64*N)      -> `value`, "scaling": "weak"
    strong : global population 64 (64/N workers per GPU)       -> `strong.value`   (BASELINE's "pop=64 at 1/2/4/8 GPU")
At N = 1 the two coincide.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POP = 64                    # BASELINE metric: pop = 64
TRAIN_EPISODES = 20
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: vector fp32 (256 CU x 4 SIMD x 16 lanes x 2 (fma) x 2 (packed) x 2.4 GHz)
PLUMBING_ENV = "LENV_BENCH_PLUMBING_ENGINE"   # tests only: "module:Class" of a stand-in engine -> CPU/gloo, tiny workload


# ----------------------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N without a torchrun environment
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start n ranks of this script as child processes (one per GPU) and wait for them.  The parent only counts devices
    (torch.cuda.device_count() does not initialise HIP on this image) and never creates a context."""
    if not os.environ.get(PLUMBING_ENV) and os.environ.get("LENV_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("LENV_BENCH_LAUNCH_TIMEOUT", "3600"))
    try:
        # poll every rank: the first failure (or the overall deadline) ends the job at once instead of leaving the other
        # ranks blocked in a collective until the RCCL watchdog fires
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------------------
def bench_config(num_workers, plumbing=False):
    from learning_environments_amd.configs import cartpole_syn_env_ddqn, fixed_work
    cfg = fixed_work(cartpole_syn_env_ddqn(num_workers=num_workers), 2 if plumbing else TRAIN_EPISODES)
    if plumbing:       # tests/test_bench_launcher.py: seconds on the CPU oracle, never reported as a measurement
        cfg["envs"]["CartPole-v0"]["max_steps"] = 10
        cfg["agents"]["ddqn"].update(test_episodes=2, batch_size=16)
    return cfg


def team_size(master):
    """Workgroups per chain the fused DDQN launch of this master uses (1 unless the launch under-fills the GPU)."""
    try:
        import ctypes as C
        from learning_environments_amd import _lib
        return int(_lib.lib().lenv_ddqn_se_team_size(C.byref(master.cfg), int(master.cpw * master.n_local)))
    except Exception:
        return None


def build_master(num_workers, engine=None, plumbing=False, team_size=0):
    from learning_environments_amd.agents.GTN import GTN_Master
    cfg = bench_config(num_workers, plumbing)
    cfg["agents"]["gtn"]["team_size"] = int(team_size)     # workgroups per chain: 0 = automatic (lenv_ddqn_cfg::team_size)
    torch.manual_seed(0)                      # theta: torch default Linear init under seed 0
    cwd = os.getcwd()
    work = os.path.join("/tmp", "lenv_bench_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)
    os.chdir(work)                            # GTN_Base creates ./results/GTN_sync relative to cwd
    try:
        master = GTN_Master(cfg, bohb_id=0, seed=1234, engine=engine)
    finally:
        os.chdir(cwd)
    with torch.no_grad():
        master.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)   # SE never terminates: fixed work per episode
    return master, cfg


def host_theta(cfgd):
    """The bench's theta built on the host (same recipe as build_master) for the CPU baseline, before the GPU is touched."""
    from learning_environments_amd.envs.env_factory import EnvFactory
    from learning_environments_amd.models.model_utils import linear_params
    torch.manual_seed(0)
    env = EnvFactory(cfgd).generate_virtual_env()
    with torch.no_grad():
        env.env.done_net[-1].bias.fill_(-10.0)
    return env, torch.cat([p.detach().reshape(-1) for p in linear_params(env)]).numpy().astype(np.float32)


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


def algorithmic_flops(master, train_steps, learn_steps, test_steps):
    """fp32 FLOPs the path needs (2 per multiply-add, activations not counted): SE step 3 nets, batch-1 Q forward per
    env step, and per learn step 3 minibatch forwards + the backward (2x one forward) + ~12 per parameter for Adam/Polyak."""
    c = master.cfg
    S, A, B, Hq, Hse = c.state_dim, c.num_actions, c.batch_size, c.q_hidden, c.se_hidden
    q_fwd = 2 * (S * Hq + Hq * A)
    se = 2 * (3 * (S + A) * Hse + (S + 2) * Hse)
    return train_steps * (se + q_fwd) + learn_steps * (5 * B * q_fwd + 12 * master.inner.p_agent) + test_steps * q_fwd


def profile_summary():
    """Numbers that cannot be read from inside the run (PMC counters, launch counts) come from the committed rocprofv3
    passes of this same command: profiles/rNN_summary.json (tools/summarize_profiles.py)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(f))
            if "hbm_traffic_bytes_per_launch" in d:
                best = (d, os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (the only place bench.py touches oracle/)
# ----------------------------------------------------------------------------------------------------------------------
def available_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (v2 cpu.max / v1 cfs_quota)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfgd, theta, grad_chunk, file_io=True):
    """The oracle (CPU port of the same path, oracle/lenv_oracle.c) timed on this box's host cores on a bounded sample of
    the same workload.  Leg (i) "threads": one chain per available core, all at once, inside one process.  Leg (ii)
    "file_io" (SURVEY.md §8(d)(ii)): the reference's deployment shape -- one single-threaded worker process per core
    talking to a master through the sync-file protocol with the reference's 'single'-mode sleeps."""
    from oracle import oracle as orc
    cores, quota = available_cores()
    model = cpu_model()
    noise_std = cfgd["agents"]["gtn"]["noise_std"]
    ocfg = orc.ddqn_cfg_from_config(cfgd, grad_chunk=grad_chunk, rng_mode=0)
    from oracle.file_worker import agent_bounds
    bounds = agent_bounds(ocfg)

    def population(pop_s, threads):
        eps = (np.random.RandomState(1).randn(pop_s, theta.size) * noise_std).astype(np.float32)
        init = ((np.random.RandomState(2).rand(3 * pop_s, bounds.size).astype(np.float32) * 2 - 1) * bounds).astype(np.float32)
        t0 = time.time()
        orc.ddqn_se_population(ocfg, theta, eps, init, seed=1234, generation=0, threads=threads)
        return time.time() - t0

    # single-thread calibration: one worker-evaluation = 3 chains on one core
    t1 = population(1, 1)
    # all cores, whole rounds: `cores` workers = 3*cores chains of equal length handed to the largest thread count <= cores that
    # divides the chain count (16 cores: 48 chains on 16 threads = exactly three rounds)
    pop_s = max(1, min(POP, cores))
    threads = max(t for t in range(1, cores + 1) if (3 * pop_s) % t == 0)
    dt = population(pop_s, threads)
    out = {"value": pop_s / dt, "unit": "worker-evaluations/s", "cores": threads, "kind": "port",
           "cpu_model": model, "affinity_cores": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota,
           "single_core_value": 1.0 / t1, "parallel_speedup": (pop_s / dt) * t1,
           "sample": "%d workers (=%d chains) of the same fixed-work CartPole-SE/DDQN workload on %d threads, %.1f s wall; single-thread calibration: 1 worker (3 chains) in %.1f s" % (pop_s, 3 * pop_s, threads, dt, t1)}
    if file_io:
        out["file_io"] = cpu_baseline_file_io(cfgd, theta, cores, deadline_s=max(60.0, 8.0 * t1), grad_chunk=grad_chunk)
    return out


def cpu_baseline_file_io(cfgd, theta, cores, deadline_s, grad_chunk):
    """One generation of a file-transport GTN_Master driving W oracle workers (oracle/file_worker.py), W = min(cores, 32)
    (each worker process imports torch for the .pt files: 32 bounds the memory).  evals/s = W / wall of the generation."""
    import copy
    import tempfile
    from learning_environments_amd.agents.GTN import GTN_Master
    from oracle.engine_standin import OracleNesEngine       # CPU-side master of the CPU baseline (oracle infrastructure)
    W = max(1, min(cores, 32))
    cfg = copy.deepcopy(cfgd)
    cfg["device"] = "cpu"
    cfg["agents"]["gtn"].update(num_workers=W, max_iterations=1, mode="single")
    work = tempfile.mkdtemp(prefix="lenv_fileio_")
    cwd = os.getcwd()
    os.chdir(work)
    procs = []
    try:
        torch.manual_seed(0)
        master = GTN_Master(cfg, bohb_id=-1, engine=OracleNesEngine(), transport="file")
        with torch.no_grad():
            master.theta.copy_(torch.from_numpy(theta))
        env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for i in range(W):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.file_worker", str(i), "--max-generations", "1",
                                           "--grad-chunk", str(int(grad_chunk))],      # the same canonical summation order as leg (i) and the kernel
                                          env=env, cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        time.sleep(min(20.0, 2.0 + 0.25 * W))     # let the workers import torch and start polling (not part of the sample)
        t0 = time.time()
        master.write_worker_inputs(0)
        deadline = t0 + deadline_s
        while time.time() < deadline:
            if all(os.path.isfile(master.get_result_check_file_name(i)) for i in range(W)):
                break
            time.sleep(master.time_sleep_master)
        else:
            return {"value": None, "cores": W, "note": "file-IO leg did not finish within %.0f s; not reported" % deadline_s}
        master.read_worker_results()
        dt = time.time() - t0
        return {"value": W / dt, "unit": "worker-evaluations/s", "cores": W, "kind": "port",
                "sample": "1 generation: file-transport GTN_Master + %d single-threaded oracle worker processes through "
                          "results/GTN_sync (reference protocol, 'single'-mode sleeps 0.02 s master / 0.2 s worker), %.1f s "
                          "wall, mean worker time %.1f s" % (W, dt, float(np.mean(master.time_elapsed_list)))}
    finally:
        os.chdir(cwd)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        import shutil
        shutil.rmtree(work, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------------------------
# the other BASELINE configurations, one shard each at the per-GPU size the config names (N = 1 only, after the headline)
# ----------------------------------------------------------------------------------------------------------------------
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD = the fp32 vector rate


def dueling_model(cfg, st):
    """SURVEY.md §8(d) algorithmic bytes and the fp32 FLOPs of the layer products of one launch, BASELINE configs[2]."""
    a = cfg["agents"]["duelingddqn"]
    S, A, H, F, L, B = 6, 3, a["hidden_size"], a["feature_dim"], a["hidden_layer"], a["batch_size"]
    P = S * H + H + (L - 1) * (H * H + H) + H * F + F + 2 * (F * F + F) + F + 1 + A * F + A
    f = 2 * (S * H + (L - 1) * H * H + H * F + 2 * F * F + F * (1 + A))                      # FLOPs of one forward row
    learn, train, test = float(st[:, 2].sum()), float(st[:, 1].sum()), float(st[:, 3].sum())
    nbytes = 4 * (learn * (B * (2 * S + 3) + 8 * P) + train * ((2 * S + 3) + (A + S) + (S + 2)) + test * (P + 2 * S + 2))
    return nbytes, learn * 5 * B * f + (train + test) * f


def td3_model(cfg, st):
    """The same for BASELINE configs[4] (actor 17-128-128-6, twin critics 23-128-128-1, policy_delay 1)."""
    a = cfg["agents"]["td3"]
    S, A, H, B = 17, 6, a["hidden_size"], a["batch_size"]
    Pa = S * H + H + H * H + H + H * A + A
    Pc = (S + A) * H + H + H * H + H + H + 1
    fa, fc = 2 * (S * H + H * H + H * A), 2 * ((S + A) * H + H * H + H)
    learn, train, test = float(st[:, 2].sum()), float(st[:, 1].sum()), float(st[:, 3].sum())
    nbytes = 4 * (learn * (B * (2 * S + A + 2) + 8 * (Pa + 2 * Pc)) + train * (2 * S + A + 2) + test * (Pa + 2 * S + 2))
    return nbytes, learn * B * (4 * fa + 10 * fc) + (train + test) * fa


def ql_model(cfg, st):
    """Tabular Q-learning on the Cliff RewardEnv (BASELINE configs[3]): 80 B per env step + the reward net read once per chain
    (SURVEY.md §8(d)); no floating-point products worth a FLOP count."""
    train, test = float(st[:, 1].sum()), float(st[:, 3].sum())
    return 80.0 * (train + test) + st.shape[0] * 4 * 2 * 1602, 0.0


def secondary_configs(only=None, team_size=0):
    """One GTN_Master per configuration, its §8(d) fixed-work form, 1 untimed + K timed generations each, HIP events around
    the generation's device work.  Sized to finish in about a minute.  only = 2 / 3 / 4: just that BASELINE configs[] entry (the
    rocprofv3 passes profile one configuration per run)."""
    from learning_environments_amd import configs as C
    from learning_environments_amd.agents.GTN import GTN_Master
    out = []

    def run(name, key, kernel, cfg, model, steps, warmup=1, se=True):
        if only is not None and key != "BASELINE configs[%d]" % only:
            return
        cfg["agents"]["gtn"]["team_size"] = int(team_size)      # A/B aid: workgroups per chain (0 = automatic = what ships)
        torch.manual_seed(0)
        cwd = os.getcwd()
        work = os.path.join("/tmp", "lenv_bench_%d" % os.getpid())
        os.makedirs(work, exist_ok=True)
        os.chdir(work)
        try:
            m = GTN_Master(cfg, bohb_id=0, seed=1234)
        finally:
            os.chdir(cwd)
        if se:
            with torch.no_grad():
                m.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)       # the SE never terminates: fixed work per episode
        dt, kernel_ms = timed_generations(m, steps, warmup, torch.cuda.synchronize, 1, True)
        st = m.inner.stats.cpu().numpy()
        pop = cfg["agents"]["gtn"]["num_workers"]
        nbytes, flops = model(cfg, st)
        learn = float(st[:, 2].mean())
        rec = {"config": key, "workload": name, "kernel": kernel, "pop_on_this_gpu": pop, "chains": int(st.shape[0]), "steps": steps,
               "ms_per_step": dt / steps * 1e3, "kernel_ms": kernel_ms, "value": pop * steps / dt, "unit": "worker-evaluations/s",
               "train_steps": int(st[:, 1].sum()), "learn_steps": int(st[:, 2].sum()), "test_steps": int(st[:, 3].sum()),
               "us_per_learn_step_per_chain": kernel_ms * 1e3 / learn if learn else None,
               "algorithmic_GBps": nbytes / (kernel_ms * 1e-3) / 1e9, "hbm_frac": nbytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "graph": bool(getattr(m, "use_graph", False)), "team_fallbacks": int(getattr(m, "team_fallbacks", 0)),
               # which launch path was timed: inside a process group -- also a one-rank one -- a generation is two graphs around
               # an eager all-gather, in a bare process one graph and no collective; records of the two launch styles are not comparable
               "graphs_per_generation": (0 if not getattr(m, "use_graph", False) else (2 if getattr(m, "_graph2", None) is not None else 1)),
               "collective_ran": bool(getattr(m, "collectives_run", 0) > 0)}
        if flops:
            busy = min(int(st.shape[0]), 256)
            if kernel in ("td3_wavechain_kernel", "dueling_wavechain_kernel"):    # a chain is run by a team of workgroups when the whole launch stays resident
                import ctypes
                from learning_environments_amd import _lib
                fn = _lib.lib().lenv_td3_rn_team_size if kernel == "td3_wavechain_kernel" else _lib.lib().lenv_dueling_team_size
                team = int(fn(ctypes.byref(m.cfg), int(st.shape[0])))
                rec["workgroups_per_chain"] = team
                busy = min(busy * max(team, 1), 256)
            tf = flops / (kernel_ms * 1e-3) / 1e12
            rec.update({"mfma_f32_TFLOPs": tf, "mfma_f32_frac": tf / MFMA_F32_PEAK_TFLOPS, "busy_cus": busy,
                        "mfma_f32_frac_of_busy_cus": tf / (MFMA_F32_PEAK_TFLOPS * busy / 256.0)})
        out.append(rec)
        del m
        torch.cuda.empty_cache()

    # configs[1]'s own 8-GPU form: BASELINE's metric is pop 64 at 1/2/4/8 GPUs, so at N = 8 a GPU holds 8 workers = 24 chains, each on a
    # team of workgroups (automatic team size).  What this box can measure is that shard's generation time; the 8-GPU value it implies is a
    # PROJECTION (64 workers / this time; the all-gather of 8 x 32 B adds microseconds), reported as such and never as `value`.
    if only is None or only == 1:
        from learning_environments_amd import _lib
        import ctypes
        m8, _ = build_master(8, team_size=team_size)
        dt8, k8 = timed_generations(m8, 10, 2, torch.cuda.synchronize, 1, True)
        st8 = m8.inner.stats.cpu().numpy()
        out.append({"config": "BASELINE configs[1] strong-scaling shard", "kernel": "ddqn_se_inner_kernel (TEAM)",
                    "workload": "CartPole-v0 SE + DDQN, pop 64 over 8 GPUs = 8 workers = 24 chains on this GPU, the headline's fixed-work form",
                    "pop_on_this_gpu": 8, "chains": int(st8.shape[0]), "steps": 10, "ms_per_step": dt8 / 10 * 1e3, "kernel_ms": k8,
                    "workgroups_per_chain": int(_lib.lib().lenv_ddqn_se_team_size(ctypes.byref(m8.cfg), int(st8.shape[0]))),
                    "us_per_learn_step_per_chain": k8 * 1e3 / float(st8[:, 2].mean()),
                    "value": 8 * 10 / dt8, "unit": "worker-evaluations/s (this shard alone)",
                    "projected_8gpu_value": POP * 10 / dt8,
                    "projected_8gpu_note": "PROJECTION, not a measurement: 64 workers / this shard's generation time, as if eight GPUs each ran "
                                           "this shard and exchanged one 8 x 32 B all-gather; the driver's SCALE run is the measurement",
                    "team_fallbacks": int(getattr(m8, "team_fallbacks", 0)), "graph": bool(getattr(m8, "use_graph", False)),
                    "graphs_per_generation": (0 if not getattr(m8, "use_graph", False) else (2 if getattr(m8, "_graph2", None) is not None else 1)),
                    "collective_ran": bool(getattr(m8, "collectives_run", 0) > 0)})
        del m8
        torch.cuda.empty_cache()
        # ... and the shards of the 2- and 4-GPU forms (32 / 16 workers = 96 / 48 chains, teams of 2 / 4), so that the record carries the whole
        # projected strong-scaling curve of BASELINE's metric -- projections all, from one-GPU measurements of each shard
        # (not under `--only-config 1`: that is the run rocprofv3 profiles, and the other shards launch the same kernel instantiations)
        curve = [{"n_gpus": 8, "workers_per_gpu": 8, "chains_per_gpu": 24, "workgroups_per_chain": out[-1]["workgroups_per_chain"],
                  "ms_per_generation": out[-1]["ms_per_step"], "projected_value": out[-1]["projected_8gpu_value"]}]
        for n_gpus, workers in ((4, 16), (2, 32)) if only is None else ():
            ms_, _ = build_master(workers, team_size=team_size)
            dts, _k = timed_generations(ms_, 10, 2, torch.cuda.synchronize, 1, True)
            curve.append({"n_gpus": n_gpus, "workers_per_gpu": workers, "chains_per_gpu": 3 * workers,
                          "workgroups_per_chain": int(_lib.lib().lenv_ddqn_se_team_size(ctypes.byref(ms_.cfg), 3 * workers)),
                          "ms_per_generation": dts / 10 * 1e3, "projected_value": POP * 10 / dts})
            del ms_
            torch.cuda.empty_cache()
        out[-1]["projected_strong_scaling"] = sorted(curve, key=lambda r: r["n_gpus"])
    # configs[2]: Acrobot SE + DuelingDDQN, pop 256 over 8 GPUs = 32 workers = 96 chains per GPU; 20 train episodes x 500 steps
    # (init_episodes 10 as published: the second half learns), 10 lock-step test episodes after every train episode
    c3 = C.fixed_work(C.acrobot_syn_env_duelingddqn(32), 20)
    run("Acrobot-v1 SE + DuelingDDQN (6-128-128 / 128 / 3, B=128): one 8-GPU shard of pop 256 = 32 workers, 20 x 500 train steps",
        "BASELINE configs[2]", "dueling_wavechain_kernel", c3, dueling_model, steps=2)
    # configs[3]: Cliff RewardEnv + QL, pop 128 on one GPU, 100 episodes (published early-out)
    c4 = C.cliff_reward_env_ql(128)
    c4["agents"]["gtn"]["quit_when_solved"] = False
    run("Cliff RewardEnv (potential-shaped) + QL, pop 128 = 384 chains, 100 episodes", "BASELINE configs[3]", "ql_rn_inner_kernel",
        c4, ql_model, steps=50, warmup=5, se=False)
    # configs[4]: HalfCheetah stand-in RewardEnv + TD3, pop 64 over 8 GPUs = 8 workers = 24 chains per GPU; 5 episodes x 1000 steps
    # (init_episodes 1 instead of the published 20, so that four of the five episodes learn)
    c5 = C.fixed_work(C.halfcheetah_reward_env_td3(8), 5)
    c5["agents"]["td3"]["init_episodes"] = 1
    run("HalfCheetah stand-in RewardEnv + TD3 (17-128-128-6, twin critics, B=192): one 8-GPU shard of pop 64 = 8 workers, 5 x 1000 "
        "train steps, init_episodes 1 instead of the published 20 (four of the five episodes learn)", "BASELINE configs[4]", "td3_wavechain_kernel", c5, td3_model, steps=2, se=False)
    return out


def next_row_records():
    """SURVEY.md §8(f) rows measured next to the contract line (never part of `value`): the evaluation harness at the experiment's size
    (reference experiments/syn_env_evaluate_cartpole_vary_hp_2.py __main__: 40 checkpoints x 10 DDQN_vary agents, mode 2, through run_vary_hp
    = ONE fused launch of 400 chains) and the two shipped small-net shapes that got their own kernel paths in round 6."""
    import copy
    import shutil
    import tempfile
    from learning_environments_amd import configs as C
    from learning_environments_amd.agents.GTN import GTN_Master
    from learning_environments_amd.experiments import syn_env_run_vary_hp as rv
    from learning_environments_amd.experiments.syn_env_evaluate import load_envs_and_config, train_test_agents
    out = []
    ck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "ckpt_cartpole_se_reference_b.pt")
    if os.path.exists(ck):
        base = torch.load(ck, map_location="cpu", weights_only=False)
        d = tempfile.mkdtemp(prefix="lenv_bench_harness_")
        try:
            gen = torch.Generator().manual_seed(1)
            for m in range(40):
                sd = {k: (v + 0.01 * torch.randn(v.shape, generator=gen)) if v.dtype.is_floating_point else v for k, v in base["model"].items()}
                cfg = copy.deepcopy(base["config"])
                cfg["envs"]["CartPole-v0"].update(max_steps=200, solved_reward=195.0)
                cfg["agents"]["ddqn_vary"]["vary_hp"] = True
                torch.save({"model": sd, "config": cfg}, os.path.join(d, "CartPole-v0_%d_%06d.pt" % (m, m)))
            rv.run_vary_hp(2, "warm", 1, 10, d, load_envs_and_config, train_test_agents, "CartPole", out_dir=d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rewards, steps, episodes = rv.run_vary_hp(2, "b", 40, 10, d, load_envs_and_config, train_test_agents, "CartPole", out_dir=d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            from learning_environments_amd.experiments.syn_env_evaluate import train_test_agents_generalization_gap
            rv.run_vary_hp(2, "warm2", 1, 10, d, load_envs_and_config, train_test_agents_generalization_gap, "CartPole", out_dir=d)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, steps_g, _ = rv.run_vary_hp(2, "g", 40, 10, d, load_envs_and_config, train_test_agents_generalization_gap, "CartPole", out_dir=d)
            torch.cuda.synchronize()
            dtg = time.perf_counter() - t1
            out.append({"row": "the same with the *_eval_generalization_gap script's agents (vary_hp off, the optimised DDQN 4-57-2 tanh, batch 199 = the headline "
                               "kernel's shape with test_mode 1): 400 agents in one launch of ddqn_se_inner_kernel", "agents": 400, "seconds": dtg,
                        "agents_per_s": 400 / dtg, "train_steps": int(sum(s_[0] for s_ in steps_g))})
            out.append({"row": "SURVEY 8(f).1 evaluation harness: run_vary_hp mode 2, 40 SE checkpoints x 10 DDQN_vary agents (drawn shapes) in one fused launch, "
                               "checkpoint loading and the result file inside the time", "agents": 400, "seconds": dt, "agents_per_s": 400 / dt,
                        "train_steps": int(sum(s_[0] for s_ in steps)), "mean_episodes": float(np.mean([e[0] for e in episodes])),
                        "models": "stand-ins for trained SEs: default_config_cartpole.yaml's shape, reward ~1 per step, 200-step episodes"})
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def generation(name, cfg, kernel, gens=3):
        torch.manual_seed(0)
        cwd = os.getcwd()
        work = os.path.join("/tmp", "lenv_bench_%d" % os.getpid())
        os.makedirs(work, exist_ok=True)
        os.chdir(work)
        try:
            m = GTN_Master(cfg, bohb_id=0, seed=7)
        finally:
            os.chdir(cwd)
        m.step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(1, 1 + gens):
            m.step(it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / gens
        st = m.inner.stats.cpu().numpy()
        out.append({"row": name, "kernel": kernel, "chains": int(st.shape[0]), "ms_per_generation": dt * 1e3,
                    "train_steps": int(st[:, 1].sum()), "learn_steps": int(st[:, 2].sum()), "test_steps": int(st[:, 3].sum())})
        del m
        torch.cuda.empty_cache()
    c = C.fixed_work(C.cmc_syn_env_td3(16), 3)
    c["agents"]["td3"].update(init_episodes=1, hidden_size=64, hidden_layer=1, activation_fn="leakyrelu")
    c["envs"]["MountainCarContinuous-v0"].update(max_steps=200, hidden_size=128, hidden_layer=3, activation_fn="relu")
    generation("default_config_cmc_syn_env_opt.yaml-like TD3 (actor 2-64-1, critics 3-64-1, B 256) on a 128x3 VirtualEnv, pop 16 = 48 chains, "
               "3 episodes x 100 agent steps", c, "td3_rn_inner_kernel<DIRECT>")
    c = C.fixed_work(C.cartpole_reward_env_ddqn(16), 6)
    c["agents"]["gtn"]["quit_when_solved"] = False
    generation("default_config_cartpole_reward_env.yaml: DDQN 4-64-2 on the real CartPole + reward net, pop 16 = 48 chains, 6 episodes", c,
               "ddqn_se_inner_kernel<RENV>")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------
def timed_generations(master, steps, warmup, barrier, world, use_events):
    """W untimed + K timed generations, barrier + synchronize on both sides, MAX over ranks.  Returns (seconds, kernel_ms)."""
    it = 0
    for _ in range(warmup):
        master.step(it)
        it += 1
    barrier()
    ev = []
    orig_inner = master.engine.inner_scores

    def timed_inner(*a, **k):
        # HIP events on the stream the fused kernel is enqueued on (= torch's current stream, see engine._stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_inner(*a, **k)
        e1.record()
        ev.append((e0, e1))
        return out

    if getattr(master, "use_graph", False) and master._graph is None:
        try:
            master._capture_generation()       # (no warm-up generation ran: capture before the event hooks go in)
        except RuntimeError as e:              # the capture is an optimisation: run eagerly if this stack refuses it
            master.use_graph, master._graph, master._graph2 = False, None, None
            master.graph_capture_error = str(e)
            torch.cuda.synchronize()
    graph = getattr(master, "_graph", None) if getattr(master, "use_graph", False) else None
    orig_replay = graph.replay if graph is not None else None

    def timed_replay():
        # captured generation: HIP events around the replay on the launching stream = device time of the generation's seven
        # kernels, of which the fused inner loop is > 99 %: an upper bound of its duration, taken over the timed region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_replay()
        e1.record()
        ev.append((e0, e1))

    if use_events and graph is not None:
        graph.replay = timed_replay
    elif use_events:
        master.engine.inner_scores = timed_inner
    t0 = time.perf_counter()
    for _ in range(steps):
        master.step(it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    master.engine.inner_scores = orig_inner
    if graph is not None and use_events:
        del graph.replay
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=master.engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None
    return dt, kernel_ms


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    plumbing = os.environ.get(PLUMBING_ENV)
    engine = None
    if plumbing:
        import importlib
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        mod, cls = plumbing.split(":")
        engine = getattr(importlib.import_module(mod), cls)()
        backend = "gloo"
    else:
        backend = os.environ.get("LENV_BENCH_BACKEND", "nccl")   # gloo: several ranks sharing one GPU (plumbing check only)

    # ---- CPU baseline first (rank 0, N = 1), so that the GPU section below is one contiguous busy interval ----
    cpu = None
    if not args.no_cpu_baseline and world == 1 and not plumbing:
        from learning_environments_amd.config import ddqn_cfg_from_config
        cfgd = bench_config(POP)
        _, theta_host = host_theta(cfgd)
        cpu_grad_chunk = ddqn_cfg_from_config(cfgd).grad_chunk      # host-side query of the library, no device call
        cpu = cpu_baseline(cfgd, theta_host, grad_chunk=cpu_grad_chunk, file_io=not args.no_file_io)

    if not plumbing:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (MI355X); the product path has no CPU fallback")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    # a launcher (torchrun, or bench.py's own) set the rendezvous variables: there is a process group, also for ONE rank -- a one-rank
    # RCCL communicator runs the same all-gather path as N ranks.  A bare `python bench.py` has none and says so in `ranks`.
    grouped = world > 1 or (os.environ.get("WORLD_SIZE") == "1" and "RANK" in os.environ and "MASTER_PORT" in os.environ)
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    def barrier():
        if grouped:
            import torch.distributed as dist
            dist.barrier()
        if not plumbing:
            torch.cuda.synchronize()

    use_events = not plumbing
    # weak: 64 workers per GPU.  (grad_chunk of the CPU baseline == the kernel's, asserted below)
    master, cfgd = build_master(POP * world, engine, bool(plumbing))
    if cpu is not None:
        assert master.cfg.grad_chunk == cpu_grad_chunk      # CPU baseline and kernel use the same canonical summation order
    t_start = time.perf_counter()
    dt, kernel_ms = timed_generations(master, args.steps, args.warmup, barrier, world, use_events)
    strong = None
    if world > 1:
        smaster, _ = build_master(POP, engine, bool(plumbing))
        sdt, skernel_ms = timed_generations(smaster, args.steps, args.warmup, barrier, world, use_events)
        strong = {"value": POP * args.steps / sdt, "unit": "worker-evaluations/s", "global_pop": POP,
                  "workers_per_gpu": smaster.w_per,
                  "ms_per_step": sdt / args.steps * 1e3, "kernel_ms": skernel_ms,
                  "workgroups_per_chain": team_size(smaster),
                  "note": "a chain's %d serial learn steps bound the generation; launches that under-fill the GPU run every chain on a "
                          "team of workgroups (DESIGN.md section 5), which shortens a learn step by 6-10 %%, not by the team "
                          "size" % (TRAIN_EPISODES * 200)}
    others = None
    if world == 1 and not plumbing and not args.no_configs:
        others = secondary_configs()           # the other BASELINE configurations, one shard each (not part of `value`)
    next_rows = None
    if world == 1 and not plumbing and not args.no_configs:
        try:
            next_rows = next_row_records()
        except Exception as e:                 # (never let an extra record cost the contract line)
            next_rows = [{"error": repr(e)[:300]}]
    gpu_section_s = time.perf_counter() - t_start

    if rank == 0:
        stats = master.inner.stats.cpu().numpy()
        total_evals = POP * world * args.steps
        line = {
            "metric": "NES worker-evaluations/sec (full inner-loop train+eval) at pop=64 per GPU",
            "value": total_evals / dt, "unit": "worker-evaluations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not plumbing else "PLUMBING TEST (CPU stand-in engine, tiny workload) -- not a measurement",
            "timed_region_s": dt, "gpu_section_s": gpu_section_s,
            # `backend` names a communicator that EXISTS; collective_ran = the fitness all-gather really executed in the timed generations
            "ranks": {"world_size": world, "backend": (("rccl" if backend == "nccl" else backend) if grouped else "none"),
                      "collective_ran": bool(getattr(master, "collectives_run", 0) > 0),
                      "collectives_per_generation": (1 if getattr(master, "collectives_run", 0) > 0 else 0),
                      "team_fallbacks": int(getattr(master, "team_fallbacks", 0)),
                      "launcher": "bench.py child processes" if os.environ.get("LENV_BENCH_SPAWNED") else ("torchrun/env" if grouped else "single process")},
            "weak": {"value": total_evals / dt, "global_pop": POP * world, "workers_per_gpu": POP, "ms_per_step": dt / args.steps * 1e3},
            "strong": strong if strong is not None else {"value": total_evals / dt, "global_pop": POP, "workers_per_gpu": POP,
                                                         "ms_per_step": dt / args.steps * 1e3, "note": "N=1: identical to weak"},
        }
        if not plumbing:
            bytes_launch, train_steps, learn_steps, test_steps = algorithmic_bytes(master, stats)
            flops_launch = algorithmic_flops(master, train_steps, learn_steps, test_steps)
            achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
            tflops = flops_launch / (kernel_ms * 1e-3) / 1e12
            chains = stats.shape[0]
            line["config"] = {
                "workload": "BASELINE configs[1]: CartPole-v0 SE (6-83-{4,1,1} leakyrelu) + DDQN (4-57-2 tanh, B=199), "
                            "NES pop=64 per GPU, fixed-work: train_episodes=%d x 200 steps, 10 real-env test episodes "
                            "per train episode + final test, early-out off" % TRAIN_EPISODES,
                "pop_per_gpu": POP, "global_pop": POP * world, "chains_per_gpu": 3 * POP, "train_episodes": TRAIN_EPISODES,
                "parallelism": ("population-sharded x%d, 1 all-gather/generation" % world) if grouped else
                               "one process, one GPU: the whole population in one launch, no process group and no collective",
                "env_steps_per_s": (train_steps + test_steps) * world / (dt / args.steps),
                "us_per_learn_step_per_chain": kernel_ms * 1e3 / (learn_steps / chains) if learn_steps else None,
                "kernel_launches_per_generation": None,
                # how a generation is launched: one captured graph in one process, two graphs around the eager all-gather with N > 1
                # ranks, 0 = eager launches (capture refused: the reason is in graph_capture_error)
                "graphs_per_generation": (0 if not getattr(master, "use_graph", False) else (2 if getattr(master, "_graph2", None) is not None else 1))}
            if getattr(master, "graph_capture_error", None):
                line["config"]["graph_capture_error"] = master.graph_capture_error[:300]
            line["roofline"] = {"bound": "hbm", "kernel": "ddqn_se_inner_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                                "algorithmic_bytes_per_launch": bytes_launch, "kernel_ms": kernel_ms,
                                "note": "latency/issue-bound small-MLP chains: weights+activations live in LDS, only the replay "
                                        "buffer touches HBM/L2 (see DESIGN.md); the binding resource is vector-instruction issue (DESIGN.md section 5, docs/notebook_r05.md section 1), see roofline_valu"}
            busy = min(chains, 256)
            line["roofline_valu"] = {"bound": "valu issue (fp32 vector)", "kernel": "ddqn_se_inner_kernel", "achieved": tflops,
                                     "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VALU_PEAK_TFLOPS,
                                     "algorithmic_flops_per_launch": flops_launch, "busy_cus": busy,
                                     "frac_of_busy_cus": tflops / (FP32_VALU_PEAK_TFLOPS * busy / 256.0)}
            ps = profile_summary()
            if ps is not None:
                d, src = ps
                line["roofline"]["traffic"] = d["hbm_traffic_bytes_per_launch"]
                line["roofline"]["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                if "launches_per_generation" in d:
                    line["config"]["kernel_launches_per_generation"] = d["launches_per_generation"]
                    line["config"]["kernel_launches_source"] = src + " (rocprofv3 --kernel-trace: all dispatches / fused-kernel dispatches)"
        else:
            line["config"] = {"workload": "plumbing self-test of the multi-rank path"}
        if others is not None:
            line["configs"] = others
        if next_rows is not None:
            line["next_rows"] = next_rows
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)

    if grouped:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-file-io", action="store_true", help="skip the file-IO worker-mode leg of the CPU baseline")
    ap.add_argument("--no-configs", action="store_true", help="skip the shards of the other BASELINE configurations")
    ap.add_argument("--only-config", type=int, default=None, choices=(1, 2, 3, 4),
                    help="run only the shard of BASELINE configs[N] (profiling aid; prints its record alone)")
    ap.add_argument("--team-size", type=int, default=0,
                    help="with --only-config: workgroups per chain (lenv_*_cfg::team_size; 0 = automatic, the shipped launch)")
    args = ap.parse_args()
    if args.only_config is not None:
        # profiling aid: one shard of one of the other BASELINE configurations, nothing else
        torch.cuda.set_device(0)
        print(json.dumps(secondary_configs(only=args.only_config, team_size=args.team_size)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["LENV_BENCH_SPAWNED"] = "1"
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
