"""learning_environments_amd -- MI355X-native NES inner loop for learned Synthetic Environments.

Drop-in for the hot path of automl/learning_environments (reference module paths in brackets):

    learning_environments_amd.envs.env_factory.EnvFactory      [envs/env_factory.py]
    learning_environments_amd.envs.env_wrapper.EnvWrapper      [envs/env_wrapper.py]
    learning_environments_amd.envs.virtual_env.VirtualEnv      [envs/virtual_env.py]
    learning_environments_amd.agents.GTN.GTN_Master/GTN_Worker [agents/GTN*.py]

All arithmetic of the path runs in hand-written gfx950 kernels behind the C-ABI of
include/lenv_hip.h (liblenv_hip.so, built from learning_environments_amd/csrc).  There is no CPU
fallback: importing the compute entry points without the built library, or calling them without
a HIP device, raises.
"""
__version__ = "0.1.0"
