import sys, runpy, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench_configs.py", "none"]
ns = runpy.run_path(os.path.join(sys.path[0], "tools/bench_configs.py"), run_name="bc")
from learning_environments_amd import configs
c = configs.fixed_work(configs.halfcheetah_syn_env_td3(32), 3)
c["agents"]["td3"].update(init_episodes=1, test_episodes=1)
c["envs"]["HalfCheetah-v3"]["max_steps"] = 100
ns["run"]("halfcheetah VirtualEnv + TD3 pop 32 (3 episodes x 100 steps)", c)
