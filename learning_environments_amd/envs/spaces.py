"""Minimal action/observation space descriptors (stand in for gym.spaces in the reference's EnvWrapper getters,
envs/env_wrapper.py:94-104,118-128)."""
import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()

    def contains(self, x):
        return 0 <= int(x) < self.n


class Box(object):
    def __init__(self, low, high):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape
