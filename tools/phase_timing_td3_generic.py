#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of chain 0 of the GEMM-queue TD3 kernel (td3_rn_inner_kernel) on VirtualEnv configurations that
no wave-chain shape serves (small nets, td3_vary).  Needs a -DLENV_PHASE_TIMING build of td3_rn_inner_loop.hip (tools/build_variant.sh) given
as LENV_TIMING_LIB.  Never used by the product path or by bench.py."""
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

names = ["act+env step+append", "replay gather", "actor_t fwd+noise", "4 critic fwds", "TD error", "critics backward", "critic adam",
         "policy fwd/bwd", "actor adam+polyak", "tests", "other"]


def run(label, c):
    c["agents"]["gtn"]["kernel_variant"] = _lib.VARIANT_NO_WAVECHAIN
    m = GTN_Master(c, bohb_id=0, seed=7)
    m.step(0)
    torch.cuda.synchronize()
    buf0 = (C.c_ulonglong * 32)()
    _lib.lib().lenv_debug_td3_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    assert _lib.lib().lenv_debug_td3_phase_cycles(buf0) == 0      # (the sub-phase counters are never reset: the first generation's share is subtracted)
    t0 = time.time(); m.step(1); torch.cuda.synchronize(); dt = time.time() - t0
    buf = (C.c_ulonglong * 32)()
    _lib.lib().lenv_debug_td3_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    assert _lib.lib().lenv_debug_td3_phase_cycles(buf) == 0
    st = m.inner.stats[0].tolist()
    tot = sum(buf[i] for i in range(11))
    print("%s: generation wall %.1f ms; stats %s; total %.1f Mcycles" % (label, dt * 1e3, st, tot / 1e6))
    for i, n in enumerate(names):
        per = buf[i] / max(1, st[2]) if 1 <= i <= 8 else (buf[i] / max(1, st[1]) if i == 0 else buf[i] / max(1, st[3]))
        print("  %-22s %12d cycles  %5.1f%%  %9.0f per %s" % (n, buf[i], 100.0 * buf[i] / max(1, tot), per,
                                                               "learn step" if 1 <= i <= 8 else ("env step" if i == 0 else "test step")))
    sub = ["stage actor_t", "actor_t fwd + noise", "stage target critics", "target critics fwd", "stage critics", "critics fwd", "TD error",
           "critic_2 fwd again into the matrix (split)", "output-layer grads (x2)", "in-place dz (x2, split)", "first-layer grads (x2)", "-",
           "env step: cumulative at 'action selected'", "env step: cumulative at 'env stepped'", "env step: cumulative at 'appended'"]
    for i in range(16):
        buf[16 + i] -= buf0[16 + i]
    if any(buf[16 + i] for i in range(16)):
        print("  t3_direct_critics sub-phases (thread 0 of chain 0), cycles per learn step:")
        for i, n in enumerate(sub):
            if buf[16 + i]:
                print("    %-46s %9.0f" % (n, buf[16 + i] / max(1, st[2] if i < 12 else st[1])))


c = configs.fixed_work(configs.cmc_syn_env_td3(16), 3)
c["agents"]["td3"].update(init_episodes=1, hidden_size=64, hidden_layer=1, activation_fn="leakyrelu")
c["envs"]["MountainCarContinuous-v0"].update(max_steps=200, hidden_size=128, hidden_layer=3, activation_fn="relu")
run("cmc_syn_env_opt-like: TD3 64x1, SE 128x3, B 256, same_action_num 2", c)
c = configs.fixed_work(configs.halfcheetah_syn_env_td3(16), 3)
c["agents"]["td3"]["init_episodes"] = 1
c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
run("halfcheetah.yaml td3 section on the GEMM-queue kernel: TD3 128x2, SE 128x3, B 256", c)
