"""Device-resident real environments (CartPole-v0 / Acrobot-v1 / MountainCar-v0 of gym==0.17.3, third party; dynamics restated in
csrc/lenv_device.cuh and UNPINNED -- see DESIGN.md).  One instance = one environment whose float64 state lives in HBM
and is stepped by lenv_real_env_step; the fused inner loop uses the same device functions for its scoring rollouts."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..engine import require_device, _ptr, _stream
from .spaces import Box, Discrete

_SPECS = {
    "CartPole-v0": dict(S=4, A=2, max_steps=200,
                        high=[4.8, np.finfo(np.float32).max, 24 * 2 * math.pi / 360, np.finfo(np.float32).max]),
    "Acrobot-v1": dict(S=6, A=3, max_steps=500, high=[1.0, 1.0, 1.0, 1.0, 4 * math.pi, 9 * math.pi]),
    "MountainCar-v0": dict(S=2, A=3, max_steps=200, low=[-1.2, -0.07], high=[0.6, 0.07]),
    # HalfCheetah-v3 is served by the documented STAND-IN (tools/gen_cheetah_standin.py): MuJoCo cannot be installed
    "HalfCheetah-v3": dict(S=17, A=6, SD=17, max_steps=1000, high=[np.inf] * 17, continuous=True, max_action=1.0),
    "Pendulum-v0": dict(S=3, A=1, SD=2, max_steps=200, high=[1.0, 1.0, 8.0], continuous=True, max_action=2.0),
    "MountainCarContinuous-v0": dict(S=2, A=1, SD=2, max_steps=999, low=[-1.2, -0.07], high=[0.6, 0.07], continuous=True, max_action=1.0),
}


class DeviceRealEnv(object):
    """gym.Env look-alike: reset() -> obs (np.float32[S]); step(a) -> (obs, reward, done, info)."""

    def __init__(self, env_name, seed=0):
        if env_name not in _SPECS:
            raise NotImplementedError("real env '%s' has no device implementation yet" % env_name)
        spec = _SPECS[env_name]
        self.env_name = env_name
        self.env_id = _lib.ENV[env_name]
        self.observation_space = Box(np.asarray(spec["low"]) if "low" in spec else -np.asarray(spec["high"]), np.asarray(spec["high"]))
        self.continuous = bool(spec.get("continuous", False))
        self.action_space = Box(-np.ones(spec["A"]) * spec["max_action"], np.ones(spec["A"]) * spec["max_action"]) if self.continuous else Discrete(spec["A"])
        self._A = spec["A"]
        self._SD = spec["SD"] if self.continuous else 4  # float64 state words
        self._max_episode_steps = spec["max_steps"]
        self._S = spec["S"]
        self._seed = int(seed)
        self._episode = 0
        self._dev = None

    def _alloc(self):
        if self._dev is None:
            dev = require_device()
            self._dev = dict(state=torch.zeros(self._SD, dtype=torch.float64, device=dev), elapsed=torch.zeros(1, dtype=torch.int32, device=dev),
                             obs=torch.zeros(self._S, dtype=torch.float32, device=dev), reward=torch.zeros(1, device=dev),
                             done=torch.zeros(1, device=dev),
                             action=torch.zeros(self._A if self.continuous else 1, dtype=torch.float32 if self.continuous else torch.int32, device=dev),
                             key=torch.zeros(1, dtype=torch.int64, device=dev), episode=torch.zeros(1, dtype=torch.int64, device=dev))
        return self._dev

    def seed(self, seed=None):
        self._seed = int(seed or 0)
        self._episode = 0
        return [self._seed]

    def reset(self):
        d = self._alloc()
        key = _lib.lib().lenv_chain_key(self._seed, 0, 0, 3)
        d["key"].fill_(np.array([key], np.uint64).view(np.int64)[0].item())
        d["episode"].fill_(self._episode)
        self._episode += 1
        if self.continuous:
            rc = _lib.lib().lenv_cont_env_reset(self.env_id, _ptr(d["key"]), _ptr(d["episode"]), 1, _ptr(d["state"]), _ptr(d["obs"]),
                                                _ptr(d["elapsed"]), _stream())
        else:
            rc = _lib.lib().lenv_real_env_reset(self.env_id, _ptr(d["key"]), _ptr(d["episode"]), 1, _ptr(d["state"]), _ptr(d["obs"]),
                                                _ptr(d["elapsed"]), _stream())
        _lib.check(rc, "lenv_real_env_reset")
        return d["obs"].cpu().numpy()

    def step(self, action):
        d = self._alloc()
        if self.continuous:
            d["action"].copy_(torch.as_tensor(np.asarray(action, np.float32).reshape(-1)))
            rc = _lib.lib().lenv_cont_env_step(self.env_id, int(self._max_episode_steps), 1, _ptr(d["action"]), _ptr(d["state"]),
                                               _ptr(d["elapsed"]), _ptr(d["obs"]), _ptr(d["reward"]), _ptr(d["done"]), _stream())
        else:
            d["action"].fill_(int(action))
            rc = _lib.lib().lenv_real_env_step(self.env_id, int(self._max_episode_steps), 1, _ptr(d["action"]), _ptr(d["state"]),
                                               _ptr(d["elapsed"]), _ptr(d["obs"]), _ptr(d["reward"]), _ptr(d["done"]), _stream())
        _lib.check(rc, "lenv_real_env_step")
        info = {}
        if self.env_name == "HalfCheetah-v3":
            # info dict of the stand-in (same keys/order as HalfCheetah-v3: x_position, x_velocity, reward_run, reward_ctrl);
            # consumed by the RewardEnv types 3,4,7,8,101,102 (reward_env.py:95-131)
            st = d["state"].cpu().numpy().reshape(-1)
            ctrl = 0.0
            for v in np.asarray(action, np.float32).reshape(-1):
                ctrl = ctrl + float(v) * float(v)
            info = {"x_position": float(st[0]), "x_velocity": float(st[8]), "reward_run": float(st[8]), "reward_ctrl": -0.1 * ctrl}
        return d["obs"].cpu().numpy(), float(d["reward"].item()), bool(d["done"].item() > 0.5), info

    def render(self, mode='human'):
        return None

    def close(self):
        return None
