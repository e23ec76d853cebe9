"""ConfigSpace 0.4.13 hyper-parameter sampling, restated from the published algorithm (third-party; see __init__.py):
UniformFloatHyperparameter: u ~ U[0,1) -> value = lower + u*(upper-lower) on the (log-)scale, back-transformed and clipped;
UniformIntegerHyperparameter: the same on the float range [lower-0.49999, upper+0.49999], then numpy.rint."""
import numpy as np


class UniformFloatHyperparameter(object):
    def __init__(self, name, lower, upper, default_value=None, q=None, log=False):
        if q is not None:
            raise NotImplementedError("quantised hyper-parameters are not used by the reference agents")
        self.name, self.lower, self.upper, self.log, self.default_value = name, float(lower), float(upper), bool(log), default_value
        if self.log and self.lower <= 0:
            raise ValueError("log-scale hyper-parameter needs a positive lower bound")
        self._lower = np.log(self.lower) if self.log else self.lower
        self._upper = np.log(self.upper) if self.log else self.upper

    def _transform(self, u):
        v = u * (self._upper - self._lower) + self._lower
        if self.log:
            v = np.exp(v)
        return float(min(self.upper, max(self.lower, v)))

    def sample(self, rs):
        return self._transform(rs.uniform())


class UniformIntegerHyperparameter(object):
    def __init__(self, name, lower, upper, default_value=None, q=None, log=False):
        self.name, self.lower, self.upper, self.log, self.default_value = name, int(lower), int(upper), bool(log), default_value
        self.ufhp = UniformFloatHyperparameter(name, self.lower - 0.49999, self.upper + 0.49999, log=log)

    def sample(self, rs):
        v = int(np.rint(self.ufhp._transform(rs.uniform())))
        return min(self.upper, max(self.lower, v))


class CategoricalHyperparameter(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("ConfigSpace stand-in (oracle shim): categorical hyper-parameters are not used on the hot path")
