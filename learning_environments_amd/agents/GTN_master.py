"""placeholder, completed below"""
import numpy as np


def rank_table(score_transform_type, n):
    """Per-rank values consumed by lenv_nes_rank_update for the rank-only transforms
    (reference: agents/GTN_master.py:205-227), computed with numpy exactly as the reference does.
    type 1: final weight of ascending rank i; types 2/3: raw NES utility of descending rank i."""
    t = np.zeros(n, np.float64)
    if score_transform_type == 1:
        for i in range(n):
            t[i] = i / (n - 1)
    elif score_transform_type in (2, 3):
        ranks = np.arange(1, n + 1).astype(float)
        for i in range(n):
            t[i] = max(0, np.log(n / 2 + 1) - np.log(ranks[i]))
    return t
