"""Hyper-parameter variation of the *_vary agents.

Reference: agents/DDQN_vary.py:26-59, agents/DuelingDDQN_vary.py:24-69, agents/TD3_vary.py:24-58 -- every agent that
select_agent builds (one per calc_score, i.e. one per chain here) draws
    lr           log-uniform float in [lr / 3, lr * 3]
    batch_size   log-uniform integer in [int(batch_size / 3), int(batch_size * 3)]
    hidden_size  log-uniform integer in [int(hidden_size / 3), int(hidden_size * 3)]
    hidden_layer uniform integer in [hidden_layer - 1, hidden_layer + 1]
from a ConfigSpace.ConfigurationSpace() and trains with those values.

ConfigSpace (requirements.txt:22 pins 0.4.13) is a third-party dependency that is not under /root/reference; its sampling
rule is restated here from the published algorithm:
  * UniformFloatHyperparameter(log=True): value = exp(u * (log(upper) - log(lower)) + log(lower)), clipped to the bounds;
  * UniformIntegerHyperparameter: the same draw on the float range [lower - 0.49999, upper + 0.49999] (log scale if log=True),
    then rounded to the nearest integer.
The reference's ConfigurationSpace is created WITHOUT a seed (its RandomState comes from OS entropy), so no bit stream can
be matched: parity is the distribution, plus a deterministic replay of recorded draws (fixture G8V).  Here u comes from the
chain's counter RNG (stream STREAM_VARY_HP, one index per hyper-parameter in ConfigSpace's alphabetical order), so a
generation is reproducible from (seed, generation, worker, kind) like everything else in the inner loop."""
import math

STREAM_VARY_HP = 11                                   # csrc/lenv_device.cuh
HP_ORDER = ("batch_size", "hidden_layer", "hidden_size", "lr")


def log_uniform_float(u, lower, upper):
    lo, hi = math.log(lower), math.log(upper)
    return min(upper, max(lower, math.exp(u * (hi - lo) + lo)))


def uniform_int(u, lower, upper, log):
    lo, hi = lower - 0.49999, upper + 0.49999
    if log:
        v = math.exp(u * (math.log(hi) - math.log(lo)) + math.log(lo))
    else:
        v = u * (hi - lo) + lo
    v = min(hi, max(lo, v))
    return int(min(upper, max(lower, round(v))))      # round(): half to even, as numpy.rint


def hp_bounds(agent_section):
    """{name: (lower, upper)} exactly as the reference writes them."""
    lr, b, h, l = agent_section["lr"], agent_section["batch_size"], agent_section["hidden_size"], agent_section["hidden_layer"]
    return {"lr": (lr / 3, lr * 3), "batch_size": (int(b / 3), int(b * 3)), "hidden_size": (int(h / 3), int(h * 3)),
            "hidden_layer": (l - 1, l + 1)}


def vary_hyperparameters(agent_section, units):
    """units: four uniforms in [0, 1) in HP_ORDER.  Returns {lr, batch_size, hidden_size, hidden_layer}."""
    bd = hp_bounds(agent_section)
    u = dict(zip(HP_ORDER, units))
    return {"lr": log_uniform_float(u["lr"], *bd["lr"]),
            "batch_size": uniform_int(u["batch_size"], *bd["batch_size"], log=True),
            "hidden_size": uniform_int(u["hidden_size"], *bd["hidden_size"], log=True),
            "hidden_layer": uniform_int(u["hidden_layer"], *bd["hidden_layer"], log=False)}


def chain_units(key):
    """The four draws of the chain with counter-RNG key `key` (host function of the library, no device work)."""
    from .. import _lib
    L = _lib.lib()
    return [L.lenv_rng_unit(int(key), STREAM_VARY_HP, i) for i in range(len(HP_ORDER))]
