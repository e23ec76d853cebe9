#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of chain 0 of the GEMM-queue DDQN / DuelingDDQN kernel (dueling_se_inner_kernel) on the shipped
configurations no wave-chain shape serves (default_config_mountaincar.yaml's DDQN 2-256-256-3, default_config_cartpole_reward_env.yaml's
DDQN 4-64-2 on a RewardEnv).  Needs a -DLENV_PHASE_TIMING build of dueling_se_inner_loop.hip (tools/build_variant.sh) given as
LENV_TIMING_LIB.  Never used by the product path or by bench.py."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_environments_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["LENV_TIMING_LIB"])
import torch
os.makedirs("/tmp/lenv_bench", exist_ok=True)
os.chdir("/tmp/lenv_bench")
from learning_environments_amd.agents.GTN import GTN_Master
from learning_environments_amd import configs

names = ["act-select fwd(I=1)", "SE step+append", "replay gather", "3x forward", "TD error", "heads backward", "feature backward",
         "adam+polyak", "tests", "other"]


def run(label, c):
    m = GTN_Master(c, bohb_id=0, seed=7)
    m.step(0)
    torch.cuda.synchronize()
    t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
    buf = (C.c_ulonglong * 16)()
    _lib.lib().lenv_debug_duel_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    assert _lib.lib().lenv_debug_duel_phase_cycles(buf) == 0
    st = m.inner.stats[0].tolist()
    tot = sum(buf[i] for i in range(10))
    print("%s: generation wall %.1f ms; stats %s; total %.1f Mcycles" % (label, dt * 1e3, st, tot / 1e6))
    for i, n in enumerate(names):
        per = buf[i] / max(1, st[2]) if 2 <= i <= 7 else (buf[i] / max(1, st[1]) if i < 2 else buf[i] / max(1, st[3]))
        print("  %-22s %12d cycles  %5.1f%%  %9.0f per %s" % (n, buf[i], 100.0 * buf[i] / max(1, tot), per,
                                                               "learn step" if 2 <= i <= 7 else ("env step" if i < 2 else "test step")))


c = configs.fixed_work(configs.mountaincar_syn_env_ddqn(16), 3)
c["agents"]["ddqn"]["init_episodes"] = 1
c["envs"]["MountainCar-v0"]["max_steps"] = 100
run("default_config_mountaincar.yaml: DDQN 2-256-256-3, B 128, ten test episodes", c)
c = configs.fixed_work(configs.cartpole_reward_env_ddqn(16), 6)
c["agents"]["gtn"]["quit_when_solved"] = False
run("default_config_cartpole_reward_env.yaml: DDQN 4-64-2 on the real CartPole + reward net, B 192", c)
