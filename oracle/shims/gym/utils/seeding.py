"""gym.utils.seeding subset.  gym 0.17.3 draws the seed from os.urandom when
seed is None; the golden-vector harness needs reproducibility, so here an unset
seed comes from a harness-controlled counter."""
import numpy as np

_counter = [12345]


def set_counter(value):
    _counter[0] = int(value)


def np_random(seed=None):
    if seed is None:
        seed = _counter[0]
        _counter[0] += 1
    rng = np.random.RandomState()
    rng.seed(int(seed) % (2 ** 32))
    return rng, seed
