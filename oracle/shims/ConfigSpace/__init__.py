"""Stub: /root/reference/agents/agent_utils.py:2-8 imports the *_vary agents, which
import ConfigSpace at module level (agents/DDQN_vary.py:3-4).  Never exercised."""
from ConfigSpace import hyperparameters


class ConfigurationSpace(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("ConfigSpace stub (oracle shim)")
