#!/usr/bin/env python3
"""Diagnostic: -DLENV_PHASE_TIMING build, per-phase shader-clock shares of chain 0 of the wave-chain DuelingDDQN kernel at
BASELINE configs[2]'s shapes (96 chains).  Never used by the product path or by bench.py.
usage: tools/phase_timing_wc.py [-DFLAG ...]"""
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

c = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(32), 3)
c["agents"]["duelingddqn"]["init_episodes"] = 1
c["envs"]["Acrobot-v1"]["max_steps"] = 100
m = GTN_Master(c, bohb_id=0, seed=7, graph=False)
m.step(0)
torch.cuda.synchronize()
t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
buf = (C.c_ulonglong * 64)()
_lib.lib().lenv_debug_wc_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
assert _lib.lib().lenv_debug_wc_phase_cycles(buf) == 0
names = ["act-select fwd(I=1)", "SE step+append", "replay gather", "target+online forward", "TD error", "-", "backward", "adam+polyak",
         "tests", "other"]
st = m.inner.stats[0].tolist()
tot = sum(buf[i] for i in range(10))
print("generation wall %.1f ms; stats %s; total %.1f Mcycles" % (dt * 1e3, st, tot / 1e6))
for i, n in enumerate(names):
    per = ""
    if i in (2, 3, 4, 6, 7):
        per = "  %8.0f per learn step" % (buf[i] / max(1, st[2]))
    elif i == 8:
        per = "  %8.0f per lock-step test forward" % (buf[i] / max(1, st[3] / 10.0))
    elif i in (0, 1):
        per = "  %8.0f per env step" % (buf[i] / max(1, st[1]))
    print("%-22s %12d cycles  %5.1f%%%s" % (n, buf[i], 100.0 * buf[i] / max(1, tot), per))

sub = {16: "bwd (a) dz + W^T image | team: head output-layer grads (member 0)", 17: "bwd (b) input-grad chain | colsum/head grads | team: layer-1 grads (member 1)", 18: "bwd (c) input image | team: the two feature-layer gradient routines (member 1)", 19: "bwd (d) weight-grad tiles",
       22: "bwd head output-layer weight grads", 20: "bwd layer 1", 21: "bwd adam tail (layers 1, 2, heads)", 24: "thin L1", 25: "thin L2", 26: "thin L3", 27: "thin L4 (v1 | a1)", 28: "thin heads", 32: "fwd smalls + W2 image",
       40: "  (b) wave 0: chain", 41: "  (b) wave 0: epilogue", 44: "  (b) wave 4: colsum", 45: "  (b) wave 4: head grads", 46: "  (b) wave 4: adam, half of layer q-1", 33: "fwd L1+L2", 34: "fwd L3", 35: "fwd L4v + V", 36: "fwd L4a + Adv"}
sub.update({48: "team fwd: smalls + W2 image", 49: "team fwd: L1 + exchange", 50: "team fwd: layer tiles (4 layers)", 51: "team fwd: exchange / staging / heads",
            52: "team fwd: final barrier", 53: "team bwd chain: first image load", 54: "team bwd chain: dz + image store + barrier", 55: "team bwd chain: chain + epilogue",
            56: "team bwd chain: final barrier", 57: "team wgrad layer: images", 58: "team wgrad layer: tiles", 59: "team wgrad layer: colsum + barrier",
            60: "team barrier behind the forward, member 0 waits", 61: "team barrier behind the chain, member 0 waits", 62: "team barrier behind the weight gradients, member 0 waits",
            63: "team barrier behind Adam, member 0 waits", 12: "  ... member 1 waits (forward)", 13: "  ... member 1 waits (chain)", 14: "  ... member 1 waits (weight gradients)",
            15: "  ... member 1 waits (Adam)"})
nthin = st[3] / 10.0 + buf[0] * 0  # lock-step test forwards; greedy forwards are counted on top
for i, n in sub.items():
    div = 2 * st[2] if i < 24 or i >= 32 else 1      # the sub-phase counters accumulate over both generations
    print("%-46s %12d  %9.0f per %s" % (n, buf[i], buf[i] / max(1, div), "learn step" if div != 1 else "(total)"))
