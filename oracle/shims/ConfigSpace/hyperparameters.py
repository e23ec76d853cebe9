class _HP(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("ConfigSpace stub (oracle shim)")


UniformFloatHyperparameter = UniformIntegerHyperparameter = CategoricalHyperparameter = _HP
