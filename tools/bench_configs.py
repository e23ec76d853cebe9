#!/usr/bin/env python3
"""Throughput of the fused kernels at the FULL shapes of BASELINE configs 2-5 on a reduced episode budget (extra
information for DESIGN.md; bench.py's contract line is config 2 only).  Prints one JSON object per config."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.makedirs("/tmp/lenv_bench", exist_ok=True)
os.chdir("/tmp/lenv_bench")

from learning_environments_amd.agents.GTN import GTN_Master  # noqa: E402
from learning_environments_amd import configs  # noqa: E402


def run(name, cfg, gens=2, extra=None, force_gemm=False):
    torch.manual_seed(0)
    m = GTN_Master(cfg, bohb_id=0, seed=7)
    if force_gemm:       # A/B aid: one sequential batch gradient (grad_chunk 0) = the GEMM-queue kernel instead of the register-resident one
        m.cfg.grad_chunk = 0
        m.inner = m.task.make_inner(m.cpw * m.n_local)
        assert m.inner.dueling
    m.step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, 1 + gens):
        m.step(it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / gens
    st = m.inner.stats.cpu().numpy()
    out = dict(config=name, pop=cfg["agents"]["gtn"]["num_workers"], chains=int(st.shape[0]), s_per_generation=dt,
               evals_per_s=cfg["agents"]["gtn"]["num_workers"] / dt, train_steps=int(st[:, 1].sum()), learn_steps=int(st[:, 2].sum()),
               test_steps=int(st[:, 3].sum()), us_per_learn_step_per_chain=1e6 * dt / max(1.0, st[:, 2].mean()))
    out.update(extra(cfg, st, dt) if extra else {})
    print(json.dumps(out))


import bench  # noqa: E402  (the byte / FLOP models live next to the contract line)
HBM_PEAK_GBPS, MFMA_F32_PEAK_TFLOPS = bench.HBM_PEAK_GBPS, bench.MFMA_F32_PEAK_TFLOPS


def _frac(model):
    def f(cfg, st, dt):
        nbytes, flops = model(cfg, st)
        return dict(algorithmic_GBps=nbytes / dt / 1e9, hbm_frac=nbytes / dt / 1e9 / HBM_PEAK_GBPS, fp32_TFLOPs=flops / dt / 1e12,
                    mfma_f32_frac=flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, busy_cus=int(st.shape[0]))
    return f


dueling_model, td3_model = _frac(bench.dueling_model), _frac(bench.td3_model)


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "3", "4", "5", "td3d"]
    if "2" in which:
        run("cfg2 CartPole SE + DDQN pop 64 (20 episodes)", configs.fixed_work(configs.cartpole_syn_env_ddqn(64), 20))
    if "2full" in which:
        # every CU busy: 85 workers = 255 chains on 256 CUs (BASELINE's metric is quoted at pop 64 = 192 chains)
        run("cfg2 CartPole SE + DDQN pop 85 = 255 chains (20 episodes)", configs.fixed_work(configs.cartpole_syn_env_ddqn(85), 20))
    if "4" in which:
        c = configs.cliff_reward_env_ql(128)
        c["agents"]["gtn"]["quit_when_solved"] = False
        run("cfg4 Cliff RN + QL pop 128 (100 episodes, early-out on)", c)
    if "3" in which:
        c = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(32), 3)
        c["agents"]["duelingddqn"]["init_episodes"] = 1
        c["envs"]["Acrobot-v1"]["max_steps"] = 100
        run("cfg3 Acrobot SE + DuelingDDQN pop 32 (3 episodes x 100 steps)", c, gens=1, extra=dueling_model)
    if "3full" in which:
        # the same workload with every CU busy (85 workers = 255 chains on 256 CUs): what the kernel delivers per chip rather
        # than per BASELINE's 8-GPU shard of 32 workers
        c = configs.fixed_work(configs.acrobot_syn_env_duelingddqn(85), 3)
        c["agents"]["duelingddqn"]["init_episodes"] = 1
        c["envs"]["Acrobot-v1"]["max_steps"] = 100
        run("cfg3 Acrobot SE + DuelingDDQN pop 85 = 255 chains (3 episodes x 100 steps)", c, gens=1, extra=dueling_model)
    if "5full" in which:
        c = configs.fixed_work(configs.halfcheetah_reward_env_td3(85), 3)
        c["agents"]["td3"]["init_episodes"] = 1
        c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
        run("cfg5 HalfCheetah-standin RN + TD3 pop 85 = 255 chains (3 episodes x 100 steps)", c, gens=1, extra=td3_model)
    if "acrobot_ddqn" in which:
        # default_config_acrobot.yaml's ddqn section (Critic_DQN 6-128-128-3, B = 128) on 96 chains: the wave-chain kernel's plain-DQN
        # shape, then the same launch on the GEMM-queue kernel (gtn.kernel_variant = NO_WAVECHAIN)
        from learning_environments_amd import _lib
        for variant, label in ((0, "wave-chain kernel"), (_lib.VARIANT_NO_WAVECHAIN, "GEMM-queue kernel")):
            c = configs.fixed_work(configs.acrobot_syn_env_ddqn(32), 3)
            c["envs"]["Acrobot-v1"]["max_steps"] = 100
            c["agents"]["gtn"]["kernel_variant"] = variant
            run("Acrobot SE + DDQN 6-128-128-3 pop 32 (3 episodes x 100 steps), " + label, c, gens=2)
    if "pendulum_td3" in which:
        # default_config_pendulum_reward_env.yaml (TD3 3-128-128-1 / 4-128-128-1 leakyrelu, B = 192, ten test episodes, reward net with two
        # hidden layers) at its own population (16 workers = 48 chains) and at 32 workers = 96 chains: the wave-chain kernel's second TD3
        # shape (teams of 3 / 2), then the same launches on the GEMM-queue kernel
        from learning_environments_amd import _lib
        for pop in (16, 32):
            for variant, label in ((0, "wave-chain kernel"), (_lib.VARIANT_NO_WAVECHAIN, "GEMM-queue kernel")):
                c = configs.fixed_work(configs.pendulum_reward_env_td3(pop), 3)
                c["agents"]["td3"]["init_episodes"] = 1
                c["envs"]["Pendulum-v0"]["max_steps"] = 100
                c["agents"]["gtn"]["kernel_variant"] = variant
                run("Pendulum RN + TD3 pop %d (3 episodes x 100 steps), %s" % (pop, label), c, gens=2)
    if "cmc_td3" in which:
        # default_config_cmc_reward_env.yaml (TD3 2-128-128-1 / 3-128-128-1 leakyrelu, B = 192, same_action_num 2, one test episode) at its own
        # population (16 workers = 48 chains): the wave-chain kernel's third TD3 shape, then the GEMM-queue kernel
        from learning_environments_amd import _lib
        for pop in (16, 32):
            for variant, label in ((0, "wave-chain kernel"), (_lib.VARIANT_NO_WAVECHAIN, "GEMM-queue kernel")):
                c = configs.fixed_work(configs.cmc_reward_env_td3(pop), 3)
                c["agents"]["td3"]["init_episodes"] = 1
                c["envs"]["MountainCarContinuous-v0"]["max_steps"] = 200
                c["agents"]["gtn"]["kernel_variant"] = variant
                run("MountainCarContinuous RN + TD3 pop %d (3 episodes x 100 agent steps), %s" % (pop, label), c, gens=2)
    if "cmc_venv_td3" in which:
        # default_config_cmc.yaml (TD3 2-128-128-1 / 3-128-128-1 relu, B = 256, policy_delay 2, same_action_num 2, trained on a VirtualEnv
        # of three 3-96-96-x nets) at its own population (128 workers = 384 chains, one workgroup per chain), as one 8-GPU shard (16 workers =
        # 48 chains, teams of 4) and at 4 workers (teams of 8): the wave-chain kernel's fourth TD3 shape, then the GEMM-queue kernel
        from learning_environments_amd import _lib
        for pop in (128, 16, 4):
            for variant, label in ((0, "wave-chain kernel"), (_lib.VARIANT_NO_WAVECHAIN, "GEMM-queue kernel")):
                c = configs.fixed_work(configs.cmc_syn_env_td3(pop), 3)
                c["agents"]["td3"]["init_episodes"] = 1
                c["envs"]["MountainCarContinuous-v0"]["max_steps"] = 200
                c["agents"]["gtn"]["kernel_variant"] = variant
                run("MountainCarContinuous SE + TD3 (B 256, policy_delay 2) pop %d (3 episodes x 100 agent steps), %s" % (pop, label), c, gens=2)
    if "venv_td3" in which:
        # the td3 sections of default_config_pendulum.yaml (16 workers) and default_config_halfcheetah.yaml (one 8-GPU shard of its 128 workers)
        # as fixed-shape agents (`td3_vary` with vary_hp off): wave-chain shapes 5 / 6 (teams of 4), then the GEMM-queue kernel
        from learning_environments_amd import _lib
        for make, env_name, label0, steps in ((configs.pendulum_syn_env_td3, "Pendulum-v0", "Pendulum SE 4-32-32-x", 100), (configs.halfcheetah_syn_env_td3, "HalfCheetah-v3", "HalfCheetah-standin SE 23-128-128-128-x", 100)):
            for variant, label in ((0, "wave-chain kernel"), (_lib.VARIANT_NO_WAVECHAIN, "GEMM-queue kernel")):
                c = configs.fixed_work(make(16), 3)
                c["agents"]["td3"]["init_episodes"] = 1
                c["envs"][env_name]["max_steps"] = steps
                c["agents"]["gtn"]["kernel_variant"] = variant
                run("%s + TD3 (B 256, policy_delay 2, ten test episodes) pop 16 (3 episodes x %d steps), %s" % (label0, steps, label), c, gens=2)
    if "mountaincar_ddqn" in which:
        # default_config_mountaincar.yaml (DDQN 2-256-256-3 relu, B = 128, ten test episodes, 16 workers = 48 chains): GEMM-queue kernel, no wave-chain shape
        c = configs.fixed_work(configs.mountaincar_syn_env_ddqn(16), 3)
        c["agents"]["ddqn"]["init_episodes"] = 1
        c["envs"]["MountainCar-v0"]["max_steps"] = 100
        run("MountainCar SE + DDQN 2-256-256-3 pop 16 (3 episodes x 100 steps), GEMM-queue kernel", c, gens=2)
    if "cartpole_rn_ddqn" in which:
        # default_config_cartpole_reward_env.yaml (DDQN 4-64-2 leakyrelu, B = 192, trained on the real CartPole with a learned reward, 16 workers)
        # round 6: the register-resident kernel's RENV instantiation (real-env training step + reward net on the env wave), then the GEMM-queue kernel
        for pop in (16, 64):
            for force, label in ((False, "register-resident kernel (RENV)"), (True, "GEMM-queue kernel")):
                c = configs.fixed_work(configs.cartpole_reward_env_ddqn(pop), 6)
                c["agents"]["gtn"]["quit_when_solved"] = False
                run("CartPole RewardEnv + DDQN 4-64-2 pop %d (6 episodes), %s" % (pop, label), c, gens=2, force_gemm=force)
    if "cmc_opt_td3" in which:
        # default_config_cmc_syn_env_opt.yaml-like: TD3 with ONE 64-wide hidden layer on a VirtualEnv of three 3-128-128-128-x nets, B = 256
        # round 6: the DIRECT instantiation (narrow nets skip the product queue), then the queued path (gtn.kernel_variant = NO_DIRECT)
        from learning_environments_amd import _lib
        for pop in (16, 64):
            for variant, label in ((0, "GEMM-queue kernel, DIRECT layer products"), (_lib.VARIANT_NO_DIRECT, "GEMM-queue kernel, queued products")):
                c = configs.fixed_work(configs.cmc_syn_env_td3(pop), 3)
                c["agents"]["td3"].update(init_episodes=1, hidden_size=64, hidden_layer=1, activation_fn="leakyrelu")
                c["envs"]["MountainCarContinuous-v0"].update(max_steps=200, hidden_size=128, hidden_layer=3, activation_fn="relu")
                c["agents"]["gtn"]["kernel_variant"] = variant
                run("MountainCarContinuous SE 128x3 + TD3 64x1 (B 256) pop %d (3 episodes x 100 agent steps), %s" % (pop, label), c, gens=2)
    if "td3d" in which:
        # TD3_discrete_vary as the syn-env YAMLs ship it (510-wide tanh nets, batch 122, hard Gumbel softmax) on an Acrobot SE
        c = configs.fixed_work(configs.acrobot_syn_env_td3_discrete(32), 3)
        c["envs"]["Acrobot-v1"]["max_steps"] = 100
        run("TD3_discrete_vary on an Acrobot SE, pop 32 (3 episodes x 100 steps, shipped 510-wide nets)", c, gens=1)
        c = configs.fixed_work(configs.cartpole_syn_env_td3_discrete(32, hidden_size=128, batch_size=128, use_layer_norm=True, activation_fn="relu"), 3)
        c["envs"]["CartPole-v0"]["max_steps"] = 100
        run("TD3_discrete_vary + LayerNorm on a CartPole SE, pop 32 (128-wide relu nets, batch 128)", c, gens=1)
    if "5" in which:
        c = configs.fixed_work(configs.halfcheetah_reward_env_td3(32), 3)
        c["agents"]["td3"]["init_episodes"] = 1
        c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
        run("cfg5 HalfCheetah-standin RN + TD3 pop 32 (3 episodes x 100 steps)", c, gens=1, extra=td3_model)
