"""EnvFactory (reference envs/env_factory.py:10-93): builds real / virtual envs from the YAML config dict."""
from .env_wrapper import EnvWrapper
from .grid_env import GridEnv
from .gridworld import LAYOUTS
from .real_env import DeviceRealEnv
from .reward_env import RewardEnv
from .virtual_env import VirtualEnv


class EnvFactory:
    def __init__(self, config):
        self.env_name = config["env_name"]
        self.device = config["device"]
        self.env_config = config["envs"][self.env_name]
        dummy_env = self.generate_real_env(print_str='EnvFactory (dummy_env): ')
        self.state_dim = dummy_env.get_state_dim()
        self.action_dim = dummy_env.get_action_dim()
        self.observation_space = dummy_env.env.observation_space
        self.action_space = dummy_env.env.action_space

    def generate_real_env(self, print_str=''):
        kwargs = self._get_default_parameters(virtual_env=False)
        env = self._generate_real_env_with_kwargs(kwargs=kwargs, env_name=self.env_name)
        return EnvWrapper(env=env)

    def generate_virtual_env(self, print_str=''):
        kwargs = self._get_default_parameters(virtual_env=True)
        env = VirtualEnv(kwargs)
        return EnvWrapper(env=env)

    def generate_reward_env(self, print_str=''):
        kwargs = self._get_default_parameters(virtual_env=True)
        real_env = self._generate_real_env_with_kwargs(kwargs=kwargs, env_name=self.env_name)
        reward_env = RewardEnv(real_env=real_env, kwargs=kwargs)
        return EnvWrapper(env=reward_env)

    def _get_default_parameters(self, virtual_env):
        kwargs = {"env_name": self.env_name, "device": self.device}
        if virtual_env:
            kwargs["state_dim"] = self.state_dim
            kwargs["action_dim"] = self.action_dim
            kwargs["observation_space"] = self.observation_space
            kwargs["action_space"] = self.action_space
            kwargs["reset_env"] = self.generate_real_env()
        for key, value in self.env_config.items():
            if isinstance(value, list):
                kwargs[key] = float(value[1])
            else:
                kwargs[key] = value
        return kwargs

    def _generate_real_env_with_kwargs(self, kwargs, env_name):
        env = GridEnv(env_name) if env_name in LAYOUTS else DeviceRealEnv(env_name)
        for key, value in kwargs.items():
            setattr(env, key, value)
        env._max_episode_steps = int(kwargs["max_steps"])
        env.kwargs = kwargs
        return env
