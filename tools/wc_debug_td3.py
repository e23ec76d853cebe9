#!/usr/bin/env python3
"""Debug aid: the wave-chain TD3 kernel (production launch) against the generic GEMM-queue kernel (a launch that asks for a step
trace) on the same inputs -- scores, counters, per-episode test means and the final parameters, bit for bit.
usage: tools/wc_debug_td3.py [episodes] [max_steps] [chains]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_environments_amd import configs, engine as eng  # noqa: E402
from learning_environments_amd.config import td3_cfg_from_config  # noqa: E402
from learning_environments_amd.agents.nes_common import chain_keys  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = int(sys.argv[2]) if len(sys.argv) > 2 else 6
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(1), E)
cfgd["agents"]["td3"]["init_episodes"] = 1
cfgd["envs"]["HalfCheetah-v3"]["max_steps"] = M
cfg = td3_cfg_from_config(cfgd)
rng = np.random.RandomState(5)
P_rn = 17 * 128 + 128 + 128 + 1
theta = (rng.randn(P_rn) * 0.2).astype(np.float32)
pop = (chains + 2) // 3
eps = (rng.randn(pop, P_rn) * 0.1).astype(np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
worker = (np.arange(chains) // 3).astype(np.int32)
sign = np.array([[0.0, 1.0, -1.0][c % 3] for c in range(chains)], np.float32)
keys = chain_keys(78, 1, worker, np.arange(chains) % 3)


def run(trace_cap):
    il = eng.Td3InnerLoop(cfg, chains, trace_cap=trace_cap, want_final_params=True, want_episode_stats=True)
    init = rng0.uniform(-0.08, 0.08, (chains, il.p_agent)).astype(np.float32)
    t0 = time.time()
    il.run(dev(theta), dev(eps), dev(worker), dev(sign), dev(init), rng_keys=dev(keys.view(np.int64)))
    torch.cuda.synchronize()
    dt = time.time() - t0
    return dict(score=il.score.cpu().numpy(), stats=il.stats.cpu().numpy(), etm=il.episode_test_mean.cpu().numpy(),
                fo=il.final_params.cpu().numpy(), status=il.status.cpu().numpy(), dt=dt, pa=il.p_actor, pc=il.p_critic)


rng0 = np.random.RandomState(9)
a = run(0)
rng0 = np.random.RandomState(9)
b = run(2)
print("wavechain %.3fs generic %.3fs" % (a["dt"], b["dt"]))
print("status", a["status"], b["status"])
print("stats\n", a["stats"], "\n", b["stats"])
print("score", a["score"], b["score"])
ok = True
for k in ("score", "stats", "etm", "fo"):
    same = np.array_equal(a[k], b[k], equal_nan=True)
    ok &= same
    print(k, "EQUAL" if same else "DIFF")
    if not same and k == "fo":
        d = np.nonzero(a[k] != b[k])
        print("  first diffs (chain, param):", list(zip(d[0][:8].tolist(), d[1][:8].tolist())), "count", d[0].size, "of", a[k].size)
        bad = np.unique(d[1])
        pa, pc = a["pa"], a["pc"]
        segs = []
        for name, base, i_dim, o_dim in (("actor", 0, 17, 6), ("critic1", pa, 23, 1), ("critic2", pa + pc, 23, 1)):
            o = base
            for nm, n in (("W0", 128 * i_dim), ("b0", 128), ("W1", 128 * 128), ("b1", 128), ("Wout", o_dim * 128), ("bout", o_dim)):
                segs.append((name + "." + nm, o, o + n)); o += n
        for n, lo, hi in segs:
            c = int(((bad >= lo) & (bad < hi)).sum())
            if c:
                sel = bad[(bad >= lo) & (bad < hi)][:3]
                print("   %-14s %6d differing; e.g." % (n, c), [(int(p) - lo, float(a[k][0, p]), float(b[k][0, p])) for p in sel])
sys.exit(0 if ok else 1)
