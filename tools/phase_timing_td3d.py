"""-DLENV_PHASE_TIMING build of td3_discrete_inner_loop.hip: shader-clock shares per phase of chain 0.
usage (GPU box): LENV_TIMING_LIB=gpurun_out_lib_td3dtiming.so python tools/phase_timing_td3d.py"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_environments_amd import _lib  # noqa: E402
if os.environ.get("LENV_TIMING_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LENV_TIMING_LIB"])
os.makedirs("/tmp/lenv_bench", exist_ok=True)
os.chdir("/tmp/lenv_bench")
from learning_environments_amd.agents.GTN import GTN_Master  # noqa: E402
from learning_environments_amd import configs  # noqa: E402

names = ["select action (actor row + noise)", "SE step + append", "critics: critic_2 backward (rest of the phase)", "critic adam", "policy update", "tests", "-", "other",
         "critics: gather + actor_target forward", "critics: gumbel / noise + four critic forwards", "critics: TD error + critic_1 backward"]


def run(label, c):
    m = GTN_Master(c, bohb_id=0, seed=7)
    m.step(0)
    torch.cuda.synchronize()
    t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
    buf = (C.c_ulonglong * 16)()
    _lib.lib().lenv_debug_td3d_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    assert _lib.lib().lenv_debug_td3d_phase_cycles(buf) == 0
    st = m.inner.stats[0].tolist()
    tot = sum(buf)
    print("%s: generation wall %.1f ms; stats %s; total %.1f Mcycles" % (label, dt * 1e3, st, tot / 1e6))
    for i, n in enumerate(names):
        if buf[i]:
            per = buf[i] / max(1, st[1] if i in (0, 1) else (st[2] if i in (2, 3, 4, 8, 9, 10) else st[3]))
            print("  %-36s %12d cycles  %5.1f%%  %9.0f per step" % (n, buf[i], 100.0 * buf[i] / max(1, tot), per))


c = configs.fixed_work(configs.cartpole_syn_env_td3_discrete(32, hidden_size=128, batch_size=128, use_layer_norm=True, activation_fn="relu"), 3)
c["envs"]["CartPole-v0"]["max_steps"] = 100
run("TD3_discrete_vary + LayerNorm on a CartPole SE, pop 32 (128-wide relu nets, batch 128)", c)
c = configs.fixed_work(configs.acrobot_syn_env_td3_discrete(32), 3)
c["envs"]["Acrobot-v1"]["max_steps"] = 100
run("TD3_discrete_vary on an Acrobot SE, pop 32 (shipped 510-wide nets)", c)
