#!/usr/bin/env python3
"""Diagnostic: build a -DLENV_PHASE_TIMING library and print per-phase shader-clock shares of chain 0 of the DuelingDDQN
kernel at config 3's shapes.  Never used by the product path or by bench.py."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "learning_environments_amd", "csrc")
OUT = "/tmp/liblenv_hip_timing.so"
srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-fPIC", "-shared", "-DLENV_PHASE_TIMING", "-o", OUT] + [a for a in sys.argv[1:] if a.startswith("-D")] + [os.path.join(CSRC, s) for s in srcs])
from learning_environments_amd import _lib
_lib.LIB_PATH = OUT
import time
import torch
os.makedirs("/tmp/lenv_bench", exist_ok=True)
os.chdir("/tmp/lenv_bench")
from learning_environments_amd.agents.GTN import GTN_Master
from learning_environments_amd import configs

c = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(32), 3)
c["agents"]["duelingddqn"]["init_episodes"] = 1
c["envs"]["Acrobot-v1"]["max_steps"] = 100
m = GTN_Master(c, bohb_id=0, seed=7)
m.step(0)
torch.cuda.synchronize()
t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
buf = (C.c_ulonglong * 16)()
_lib.lib().lenv_debug_duel_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
assert _lib.lib().lenv_debug_duel_phase_cycles(buf) == 0
names = ["act-select fwd(I=1)", "SE step+append", "replay gather", "3x forward", "TD error", "heads backward", "feature backward",
         "adam+polyak", "tests", "other"]
tot = sum(buf[i] for i in range(10))
print("generation wall %.1f ms; stats %s; total %.1f Mcycles" % (dt * 1e3, m.inner.stats[0].tolist(), tot / 1e6))
for i, n in enumerate(names):
    print("%-22s %12d cycles  %5.1f%%" % (n, buf[i], 100.0 * buf[i] / max(1, tot)))

# ---- config 5: TD3 kernel ----
c = configs.fixed_work(configs.halfcheetah_reward_env_td3(32), 3)
c["agents"]["td3"]["init_episodes"] = 1
c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
m = GTN_Master(c, bohb_id=0, seed=7)
m.step(0)
torch.cuda.synchronize()
t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
buf = (C.c_ulonglong * 16)()
_lib.lib().lenv_debug_td3_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
assert _lib.lib().lenv_debug_td3_phase_cycles(buf) == 0
names = ["act+env step+append", "replay gather", "actor_t fwd+noise", "4 critic fwds", "TD error", "critics backward", "critic adam",
         "policy fwd/bwd", "actor adam+polyak", "tests", "other"]
tot = sum(buf[i] for i in range(11))
print("TD3 generation wall %.1f ms; stats %s; total %.1f Mcycles" % (dt * 1e3, m.inner.stats[0].tolist(), tot / 1e6))
for i, n in enumerate(names):
    print("%-22s %12d cycles  %5.1f%%" % (n, buf[i], 100.0 * buf[i] / max(1, tot)))
