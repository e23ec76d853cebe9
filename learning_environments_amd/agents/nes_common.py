"""Shared host logic of the NES outer loop: sharding, seeds, agent initialisation, rank utilities."""
import math

import numpy as np
import torch

_G = np.uint64(0x9e3779b97f4a7c15)


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
    x ^= x >> np.uint64(31)
    return x


def chain_keys(seed, generation, workers, kinds):
    """Vectorised lenv_chain_key (csrc/lenv_api.hip): counter-RNG key of chain (worker, kind)."""
    with np.errstate(over="ignore"):
        k = _mix64(np.full(len(workers), seed, np.uint64) + _G)
        k = _mix64(k ^ (np.uint64(generation) + _G * np.uint64(2)))
        k = _mix64(k ^ (np.asarray(workers, np.uint64) + _G * np.uint64(3)))
        k = _mix64(k ^ (np.asarray(kinds, np.uint64) + _G * np.uint64(4)))
    return k


def shard_bounds(pop, rank, world):
    """Workers [lo, hi) of `rank`: contiguous blocks of ceil(pop/world) (the all-gather is rank-major = worker order)."""
    per = (pop + world - 1) // world
    lo = min(rank * per, pop)
    return lo, min(lo + per, pop), per


def rank_table(score_transform_type, n):
    """Per-rank values consumed by lenv_nes_rank_update for the rank-only transforms, computed with numpy exactly as
    reference agents/GTN_master.py:205-227 does.  type 1: weight of ascending rank i; types 2/3: raw NES utility
    max(0, log(n/2+1) - log(rank)) of descending rank i (the normalisations run on device in worker order)."""
    t = np.zeros(n, np.float64)
    if score_transform_type == 1:
        for i in range(n):
            t[i] = i / (n - 1)
    elif score_transform_type in (2, 3):
        ranks = np.arange(1, n + 1).astype(float)
        for i in range(n):
            t[i] = max(0, np.log(n / 2 + 1) - np.log(ranks[i]))
    return t


def linear_init_bounds(layer_dims):
    """Per-parameter bound of nn.Linear's default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    for weight and bias) for an MLP given as [(fan_in, fan_out), ...], flat state-dict order."""
    parts = []
    for fan_in, fan_out in layer_dims:
        b = 1.0 / math.sqrt(fan_in)
        parts.append(np.full(fan_in * fan_out + fan_out, b, np.float32))
    return np.concatenate(parts)


def with_layer_norm_block(bounds, ln_slice):
    """The bounds vector of a net whose shared nn.LayerNorm (weight | bias, 2 H values) sits at `ln_slice` = (offset, H) -- or of several
    nets, `ln_slice` = [(offset, H), ...] ascending, offsets in the FINAL vector: bound 0 there -- the draw leaves zeros,
    set_layer_norm_init then writes the weight's ones (nn.LayerNorm: weight 1, bias 0)."""
    if not ln_slice:
        return bounds
    slices = [ln_slice] if isinstance(ln_slice, tuple) else list(ln_slice)
    parts, pos, inserted = [], 0, 0
    for off, H in slices:
        orig = off - inserted
        parts += [bounds[pos:orig], np.zeros(2 * H, np.float32)]
        pos, inserted = orig, inserted + 2 * H
    parts.append(bounds[pos:])
    return np.concatenate(parts)


def set_layer_norm_init(agent_init, ln_slice):
    if ln_slice and agent_init is not None:
        for off, H in ([ln_slice] if isinstance(ln_slice, tuple) else ln_slice):
            agent_init[:, off:off + H] = 1.0
    return agent_init


def fresh_agent_init(bounds, chains, generator, device):
    """`chains` freshly initialised agents (reference: select_agent -> DDQN() per calc_score, agents/agent_utils.py:15-66)."""
    u = torch.rand((chains, bounds.numel()), generator=generator, device=device, dtype=torch.float32)
    return (u * 2.0 - 1.0) * bounds


def host_worker_best(score_add, score_sub, mirrored, grad_eval_type):
    """GTN_Worker.calc_best_score's arithmetic for ONE worker on the host (reference agents/GTN_worker.py:234-254): statistics.mean
    (exactly rounded) or the minimum of each side, then the mirrored pick.  Returns (score_best, sign): sign -1 = the -eps side won.
    The population form of the same rule is lenv_nes_worker_best_multi (csrc/nes_update.hip); a GPU test holds the two together."""
    import statistics
    if grad_eval_type == 'mean':
        sub, add = statistics.mean(score_sub), statistics.mean(score_add)
    elif grad_eval_type == 'minmax':
        sub, add = min(score_sub), min(score_add)
    else:
        raise NotImplementedError('Unknown parameter for grad_eval_type: ' + str(grad_eval_type))
    if mirrored and sub > add:
        return sub, -1.0
    return add, 1.0
