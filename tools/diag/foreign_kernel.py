#!/usr/bin/env python3
"""Diagnostic (GPU box): does a team launch next to a foreign kernel give up within LENV_TEAM_GIVEUP_TICKS?  Prints timings + statuses."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from learning_environments_amd import _lib, configs, engine as eng
from learning_environments_amd.config import td3_cfg_from_config
from learning_environments_amd.agents.nes_common import chain_keys
from tools import diag
if os.environ.get("LENV_TIMING_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LENV_TIMING_LIB"])
L = _lib.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
side = torch.cuda.Stream()
# 1. plain concurrency: a torch op on the default stream while the foreign kernel runs
with torch.cuda.stream(side):
    diag.occupy_cus(128, 150 * 1024, 100_000_000, side.cuda_stream)
t0 = time.time(); x = torch.ones(1 << 20, device="cuda"); y = (x * 2).sum().item(); print("torch op next to the foreign kernel: %.3f s (foreign done: %s)" % (time.time() - t0, side.query()))
side.synchronize(); print("foreign kernel ended after %.3f s" % (time.time() - t0))
cfgd = configs.fixed_work(configs.halfcheetah_reward_env_td3(8), 3)
cfgd["agents"]["td3"]["init_episodes"] = 1
cfgd["envs"]["HalfCheetah-v3"]["max_steps"] = 20
cfg = td3_cfg_from_config(cfgd)
chains = 24
rng = np.random.RandomState(91)
P_rn = 17 * 128 + 128 + 128 + 1
theta = (rng.randn(P_rn) * 0.2).astype(np.float32); eps = (rng.randn(8, P_rn) * 0.1).astype(np.float32)
worker = (np.arange(chains) // 3).astype(np.int32); sign = np.tile(np.array([0.0, 1.0, -1.0], np.float32), 8)
keys = chain_keys(80, 3, worker, np.arange(chains) % 3)
init = rng.uniform(-0.08, 0.08, (chains, 59016)).astype(np.float32)
args = (dev(theta), dev(eps), dev(worker), dev(sign), dev(init)); kw = dict(rng_keys=dev(keys.view(np.int64)))
cfg.team_size = 1
il = eng.Td3InnerLoop(cfg, chains, want_final_params=True)
il.run(*args, **kw); torch.cuda.synchronize()
ref = [t.cpu().numpy().copy() for t in (il.score, il.stats, il.final_params)]
cfg.team_size = 0
for occupy in (232,):
    il = eng.Td3InnerLoop(cfg, chains, want_final_params=True)
    torch.cuda.synchronize()
    print("team size", L.lenv_td3_rn_team_size(C.byref(cfg), chains))
    if occupy:
        with torch.cuda.stream(side):
            diag.occupy_cus(occupy, 150 * 1024, 500_000_000, side.cuda_stream)
        time.sleep(0.05)
    t0 = time.time()
    il.run(*args, **kw)
    torch.cuda.current_stream().synchronize()
    dt = time.time() - t0
    print("occupy %3d CUs: team launch returned after %.3f s, statuses %s, foreign done %s" % (occupy, dt, sorted(set(il.status.cpu().tolist())), side.query()))
    out = [t.cpu().numpy().copy() for t in (il.score, il.stats, il.final_params)]
    bad = [c for c in range(chains) if not (out[0][c] == ref[0][c] and np.array_equal(out[2][c], ref[2][c]))]
    print("   chains that differ from the one-workgroup launch:", bad, "statuses", il.status.cpu().tolist())
    if os.environ.get("LENV_TIMING_LIB") and occupy:
        fp = il.final_params.cpu().numpy()
        for c in range(chains):
            print("   chain %2d:" % c, " ".join("g%d[x%d %.1f %.1f %.1f]" % (g, int(fp[c, 8 * g + 3]), fp[c, 8 * g], fp[c, 8 * g + 1], fp[c, 8 * g + 2]) for g in range(6)))
    side.synchronize()
