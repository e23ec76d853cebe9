import json, sys, numpy as np, torch
sys.path.insert(0, '.')
from learning_environments_amd import _lib
import os
if os.environ.get("LENV_TIMING_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["LENV_TIMING_LIB"])
from learning_environments_amd import engine
from oracle import oracle as orc
g = np.load('tests/golden/g8t_calc_score_cheetah_td3.npz')
cfgd = json.loads(str(g["config_json"]))
H, B = 24, 32
cfgd["agents"]["td3"].update(hidden_size=H, hidden_layer=1, batch_size=B, activation_fn="relu", policy_delay=1, train_episodes=2, init_episodes=1, test_episodes=1)
cfgd["envs"]["HalfCheetah-v3"].update(max_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3, hidden_size=24, reward_env_type=0)
ocfg = orc.td3_cfg_from_config(cfgd, rng_mode=0)
res = {}
runs = []
for variant in (0, 0, _lib.VARIANT_NO_DIRECT):
    c = _lib.Td3Cfg()
    for f, _ in _lib.Td3Cfg._fields_: setattr(c, f, getattr(ocfg, f, 0))
    c.kernel_variant = variant
    il = engine.Td3InnerLoop(c, 1, want_final_params=True)
    rng = np.random.RandomState(11)
    theta = (rng.randn(1) * 0.2).astype(np.float32)
    init = rng.uniform(-0.2, 0.2, (1, il.p_agent)).astype(np.float32)
    keys = np.array([orc.chain_key(21, 4, 0, 0)], np.uint64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    il.run(t(theta), None, None, None, t(init), rng_keys=t(keys.view(np.int64)))
    torch.cuda.synchronize()
    res[variant] = il.final_params.cpu().numpy()[0].copy()
    runs.append(res[variant])
    print(variant, il.status.cpu().tolist(), il.stats.cpu().tolist(), float(il.score[0]))
print('direct run-to-run max diff', float(np.abs(runs[0] - runs[1]).max()))
d = np.abs(res[0] - res[_lib.VARIANT_NO_DIRECT])
S, A = 17, 6
Pa = S*H + H + A*H + A; Pc = (S+A)*H + H + H + 1
def blocks(off, inn, out):
    return [("W0", off, off + H*inn), ("b0", off + H*inn, off + H*inn + H), ("Wo", off + H*inn + H, off + H*inn + H + out*H), ("bo", off + H*inn + H + out*H, off + H*inn + H + out*H + out)]
for name, off, inn, out in (("actor", 0, S, A), ("critic1", Pa, S+A, 1), ("critic2", Pa+Pc, S+A, 1)):
    for bn, lo, hi in blocks(off, inn, out):
        print(name, bn, float(d[lo:hi].max()), int((d[lo:hi] > 0).sum()), hi - lo)
