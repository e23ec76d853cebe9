"""Times the REFERENCE's train_test_agents (experiments/syn_env_evaluate_cartpole_vary_hp_2.py:25-48) on this container's CPU for context
next to tools/bench_harness.py: `agents_num` DDQN_vary agents on a CartPole SE of the same kind (reward ~1 per step, 200-step episodes).
Imports /root/reference: runs in the build container only (test infrastructure, like gen_golden.py).  usage: python oracle/time_reference_harness.py [agents_num]"""
import os
import shutil
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg            # noqa: E402  (path setup, shims, seed_all, quiet)
import torch                       # noqa: E402


def main():
    agents_num = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import importlib
    import ConfigSpace
    ev = importlib.import_module("experiments.syn_env_evaluate_cartpole_vary_hp_2")
    from envs.env_factory import EnvFactory
    cfg = gg.load_cfg("default_config_cartpole.yaml")
    cfg["envs"]["CartPole-v0"].update(max_steps=200, solved_reward=195.0)
    gg.seed_all(4200)
    with gg.quiet():
        venv = EnvFactory(cfg).generate_virtual_env()
    with torch.no_grad():
        venv.env.reward_net[-1].bias.add_(1.0)
        venv.env.done_net[-1].bias.add_(0.3)
    tmp = tempfile.mkdtemp(prefix="lenv_time_")
    torch.save({'model': venv.state_dict(), 'config': cfg}, os.path.join(tmp, "model.pt"))
    cwd = os.getcwd()
    os.chdir(os.path.join(gg.REF, "experiments"))
    try:
        with gg.quiet():
            venv, real_env, config = ev.load_envs_and_config(file_name="model.pt", model_dir=tmp, device="cpu")
    finally:
        os.chdir(cwd)
    shutil.rmtree(tmp, ignore_errors=True)
    torch.set_num_threads(1)                                  # one process per model in the reference's pool
    for mode, env in ((2, venv), (0, real_env)):
        gg.seed_all(7)
        ConfigSpace.RANDOM.seed(7)
        t0 = time.time()
        with gg.quiet():
            rewards, steps, episodes = ev.train_test_agents(train_env=env, test_env=real_env, config=config, agents_num=agents_num)
        dt = time.time() - t0
        print("reference train_test_agents on 1 CPU thread, mode %d: %d agents in %.1f s = %.3f agents/s; train steps %s, episodes %s"
              % (mode, agents_num, dt, agents_num / dt, [s[0] for s in steps], [e[0] for e in episodes]), flush=True)


if __name__ == "__main__":
    main()
