"""Minimal stand-in for gym==0.17.3 (TEST INFRASTRUCTURE ONLY).

The reference (/root/reference, requirements.txt:47) imports gym, which is not
installed in this image and cannot be fetched.  This package restates, from the
published gym 0.17.3 sources, exactly the subset the NES hot path touches
(SURVEY.md Appendix B): Env, Wrapper, spaces.Discrete/Box, utils.seeding,
wrappers.TimeLimit, make() for CartPole-v0 / Acrobot-v1 (+ a documented
HalfCheetah stand-in).  It is only ever put on PYTHONPATH by oracle/gen_golden.py
so that the read-only reference can be imported to produce golden vectors.
Real-env physics restated here is "parity unpinned" (no reference test pins it).
"""
from gym.core import Env, Wrapper
from gym import spaces, utils, wrappers, envs
from gym.envs import make

__version__ = "0.17.3-shim"
