"""EnvFactory (reference envs/env_factory.py:10-93): builds real / virtual / reward envs from the YAML config dict.

The method names and the keys of the `kwargs` dict an env receives are the drop-in surface (VirtualEnv / RewardEnv read
`state_dim`, `action_dim`, `observation_space`, `action_space`, `reset_env`, `env_name`, `device` plus every entry of
config["envs"][env_name], where a `[low, default, high, ...]` list stands for its default).  The real envs are this package's
own: a device-resident gym-classic-control env (`DeviceRealEnv`) or a compiled gridworld (`GridEnv`)."""
from .env_wrapper import EnvWrapper
from .grid_env import GridEnv
from .gridworld import LAYOUTS
from .real_env import DeviceRealEnv
from .reward_env import RewardEnv
from .virtual_env import VirtualEnv

_SHAPE_KEYS = ("state_dim", "action_dim", "observation_space", "action_space")


def _default_of(entry):
    """A hyper-parameter written as a search range `[low, default, high]` is used at its default."""
    return float(entry[1]) if isinstance(entry, list) else entry


class EnvFactory:
    def __init__(self, config):
        self.env_name, self.device = config["env_name"], config["device"]
        self.env_config = config["envs"][self.env_name]
        probe = self.generate_real_env(print_str='EnvFactory (dummy_env): ')       # the shapes come from a throw-away real env
        self.state_dim, self.action_dim = probe.get_state_dim(), probe.get_action_dim()
        self.observation_space, self.action_space = probe.env.observation_space, probe.env.action_space

    # ---- the three products ----
    def generate_real_env(self, print_str=''):
        return EnvWrapper(env=self._generate_real_env_with_kwargs(self._get_default_parameters(False), self.env_name))

    def generate_virtual_env(self, print_str=''):
        return EnvWrapper(env=VirtualEnv(self._get_default_parameters(True)))

    def generate_reward_env(self, print_str=''):
        params = self._get_default_parameters(True)
        return EnvWrapper(env=RewardEnv(real_env=self._generate_real_env_with_kwargs(params, self.env_name), kwargs=params))

    # ---- helpers ----
    def _get_default_parameters(self, virtual_env):
        params = dict(env_name=self.env_name, device=self.device)
        if virtual_env:
            params.update({k: getattr(self, k) for k in _SHAPE_KEYS})
            params["reset_env"] = self.generate_real_env()
        params.update({k: _default_of(v) for k, v in self.env_config.items()})
        return params

    def _generate_real_env_with_kwargs(self, kwargs, env_name):
        env = GridEnv(env_name) if env_name in LAYOUTS else DeviceRealEnv(env_name)
        vars(env).update(kwargs)                  # every config entry is readable as an attribute (solved_reward, max_steps, ...)
        env._max_episode_steps = int(kwargs["max_steps"])
        env.kwargs = kwargs
        return env
