#!/usr/bin/env python3
"""Diagnostic: -DLENV_PHASE_TIMING build, per-phase shader-clock shares of chain 0 of the wave-chain TD3 kernel at BASELINE
configs[4]'s shapes (96 chains).  Never used by the product path or by bench.py."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "learning_environments_amd", "csrc")
OUT = "/tmp/liblenv_hip_timing.so"
srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
if os.environ.get("LENV_TIMING_LIB"):                    # a diagnostic build made elsewhere (tools/build_variant.sh): just load it
    OUT = os.path.abspath(os.environ["LENV_TIMING_LIB"])
else:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-fPIC", "-shared", "-DLENV_PHASE_TIMING", "-o", OUT] + [a for a in sys.argv[1:] if a.startswith("-D")] +
                      [os.path.join(CSRC, s) for s in srcs])
from learning_environments_amd import _lib
_lib.LIB_PATH = OUT
import torch
os.makedirs("/tmp/lenv_bench", exist_ok=True)
os.chdir("/tmp/lenv_bench")
from learning_environments_amd.agents.GTN import GTN_Master
from learning_environments_amd import configs

c = configs.fixed_work(configs.halfcheetah_reward_env_td3(int(os.environ.get("LENV_TIMING_POP", "32"))), 3)
c["agents"]["td3"]["init_episodes"] = 1
c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
m = GTN_Master(c, bohb_id=0, seed=7, graph=False)
m.step(0)
torch.cuda.synchronize()
t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
buf = (C.c_ulonglong * 48)()
_lib.lib().lenv_debug_t3w_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
assert _lib.lib().lenv_debug_t3w_phase_cycles(buf) == 0
names = ["act+env step+append", "replay gather", "actor_t fwd+noise", "4 critic fwds", "TD error", "critics backward", "critic adam",
         "policy fwd/bwd", "actor adam+polyak", "tests", "other"]
st = m.inner.stats[0].tolist()
tot = sum(buf[i] for i in range(11))
print("team barriers: %d cycles total, %.0f per learn step" % (buf[11], buf[11] / max(1, st[2])))
print("TD3 generation wall %.1f ms; stats %s; total %.1f Mcycles" % (dt * 1e3, st, tot / 1e6))
for i, n in enumerate(names):
    per = buf[i] / max(1, st[2]) if 1 <= i <= 8 else (buf[i] / max(1, st[1]) if i == 0 else buf[i] / max(1, st[3]))
    print("%-22s %12d cycles  %5.1f%%  %9.0f per %s" % (n, buf[i], 100.0 * buf[i] / max(1, tot), per,
                                                         "learn step" if 1 <= i <= 8 else ("env step" if i == 0 else "test step")))

sub = {16: "fwd smalls + W2 image", 17: "fwd L1", 18: "fwd L2 chain", 19: "fwd epilogue + output layer", 20: "fwd final barrier wait",
       24: "bwd output-layer grads (VALU)", 25: "bwd dz2 + W2^T image", 26: "bwd chain (+dX) | gb2", 27: "bwd W2 weight grads (2 halves)",
       28: "bwd layer-1 grads (2 halves)",
       30: "split fwd: smalls + W2 image + barrier", 31: "split fwd: L1 + exchange + barrier", 32: "split fwd: L2 tile chain + exchange + barrier",
       33: "split fwd: output layer (wave 0)", 34: "split fwd: final barrier",
       35: "split bwd: Wo + W2^T loads + barrier", 36: "split bwd: dz2 + image store + barrier", 37: "split bwd: tile chain + exchange + barrier",
       38: "split bwd: dX (wave 0)", 39: "split bwd: final barrier"}
# the sub-phase counters accumulate over both generations and over all calls of a learn step (7 forwards, 4 backwards)
for i, n in sub.items():
    calls = 7 if (i < 24 or 30 <= i <= 34) else (4 if (i in (25, 26) or i >= 35) else 3)
    print("%-36s %12d  %9.0f per call" % (n, buf[i], buf[i] / max(1, 2 * st[2] * calls)))
