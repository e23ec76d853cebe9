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
       30: "team fwd: weights + L1 + exchange + barrier", 31: "team fwd: L2 chain + exchange + barrier", 32: "team fwd: output layer (16x16x4)",
       33: "team fwd: final barrier", 35: "team bwd: weights + dz2 + exchange + barrier", 36: "team bwd: chain + exchange + barrier",
       37: "team bwd: dX", 38: "team bwd: final barrier", 40: "team wgrad+adam jobs (wave 0)", 41: "team wgrad: final barrier",
       42: "  fwd detail: entry -> all weight loads back", 43: "  fwd detail: L1 + epilogue + put", 44: "  fwd detail: get + L2 chain drained", 45: "  fwd detail: epilogue + h2 store + put"}
# the sub-phase counters accumulate over both generations and over all calls of a learn step (7 forwards, 4 backwards)
for i, n in sub.items():
    calls = 5 if (30 <= i <= 34 or 42 <= i <= 45) else (3 if 35 <= i <= 39 else (2 if i >= 40 else (7 if i < 24 else (4 if i in (25, 26) else 3))))
    print("%-36s %12d  %9.0f per call" % (n, buf[i], buf[i] / max(1, 2 * st[2] * calls)))

if hasattr(_lib.lib(), "lenv_debug_t3v_jobs"):
    jb = (C.c_ulonglong * 20)()
    _lib.lib().lenv_debug_t3v_jobs.argtypes = [C.POINTER(C.c_ulonglong)]
    if _lib.lib().lenv_debug_t3v_jobs(jb) == 0:
        for ph, pn in enumerate(("critics", "actor")):
            for cl, cn in enumerate(("W2 tile job (chain + optimizer epilogue)", "W1 tile job", "bias job", "output-layer job", "  tile chain alone")):
                sm, ct = jb[(ph * 5 + cl) * 2], jb[(ph * 5 + cl) * 2 + 1]
                print("wgrad %-8s %-44s %9.0f cycles per job (%d jobs)" % (pn, cn, sm / max(1, ct), ct))

for i, n in ((24, "  actor: layer 1 (wave 0)"), (25, "  actor: barrier 1"), (26, "  actor: layer 2"), (27, "  actor: barrier 2"), (28, "  actor: output layer"), (29, "  actor: barrier 3 (incl. the noise wave)"), (12, "test step: obs"), (13, "test step: one-row actor"), (14, "test step: action noise (det_normal) + clamp"), (15, "test step: env dynamics + reward")):
    if buf[i] and not buf[0]:
        print("%-44s %9.0f cycles per test step" % (n, buf[i] / max(1, 2 * st[3])))
