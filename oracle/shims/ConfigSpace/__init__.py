"""Stand-in for ConfigSpace 0.4.13 (requirements.txt:22; third-party, not under /root/reference) so that the reference's
*_vary agents (agents/DDQN_vary.py:26-59 ...) can be RUN by oracle/gen_golden.py.  Only what those agents use: an unconditional
ConfigurationSpace of uniform float / integer hyper-parameters and sample_configuration().  The sampling rule is the
published one (see hyperparameters.py); hyper-parameters are drawn in name order like ConfigSpace's sorted space.  The
real library seeds its RandomState from OS entropy when no seed is given; this stand-in draws from the module-level
`RANDOM` so that a fixture is reproducible (gen_golden seeds it)."""
import numpy as np

from ConfigSpace import hyperparameters

RANDOM = np.random.RandomState(0)


class ConfigurationSpace(object):
    def __init__(self, seed=None):
        self._hps = {}
        self.random = np.random.RandomState(seed) if seed is not None else RANDOM

    def add_hyperparameter(self, hp):
        self._hps[hp.name] = hp
        return hp

    def get_hyperparameters(self):
        return [self._hps[k] for k in sorted(self._hps)]

    def sample_configuration(self):
        return {hp.name: hp.sample(self.random) for hp in self.get_hyperparameters()}
