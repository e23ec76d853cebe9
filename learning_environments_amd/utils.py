"""Helpers with the exact semantics of the reference's utils.py (the hot-path subset)."""
import torch


def to_one_hot_encoding(normal, one_hot_dim):
    """reference utils.py:108-126 -- scalar / 1-element tensor -> [one_hot_dim]; 1-D vector -> [n, one_hot_dim];
    float indices are int()-truncated."""
    if torch.is_tensor(normal):
        normal = normal.squeeze()
    if not torch.is_tensor(normal):
        one_hot = torch.zeros(one_hot_dim)
        one_hot[int(normal)] = 1
    elif normal.dim() == 0 or (normal.dim() == 1 and len(normal) == 1):
        one_hot = torch.zeros(one_hot_dim)
        one_hot[int(normal.item())] = 1
    elif normal.dim() == 1:
        idx = normal.to(torch.int64)          # int() truncation toward zero
        one_hot = torch.zeros(len(normal), one_hot_dim)
        one_hot[torch.arange(len(normal)), idx.cpu()] = 1
    else:
        raise NotImplementedError('One hot encoding supported only for scalar values and 1D vectors')
    return one_hot


def from_one_hot_encoding(one_hot):
    """reference utils.py:129-130"""
    return torch.tensor([torch.argmax(one_hot)])


def calc_abs_param_sum(model):
    """reference utils.py:133-137"""
    sm = 0
    for param in model.parameters():
        sm += abs(param).sum()
    return sm


class AverageMeter:
    """Running record of per-episode values with windowed means -- the behaviour of reference utils.py:75-105 (host-side
    bookkeeping; the device kernels implement the window mean themselves).  Parity-critical detail: a window mean divides by
    `len(window) + 1e-9`, so an empty window gives 0.0 and early-out comparisons see the slightly shrunk mean."""

    def __init__(self, print_str):
        self.print_str, self.vals, self.it = print_str, [], 0

    def _window(self, size, skip):
        """The `size` values that precede the newest `skip` ones (fewer at the start of a run)."""
        end = max(len(self.vals) - skip, 0)
        return self.vals[max(end - size, 0):end]

    def _mean(self, num, ignore_last):
        win = self._window(num, ignore_last)
        return sum(win) / (len(win) + 1e-9)

    def update(self, val, print_rate=10):
        self.vals.append(val.item() if torch.is_tensor(val) else val)
        self.it += 1
        if self.it % print_rate == 0:
            print(self.print_str + "{:15.6f} {:>25} {}".format(self._mean(print_rate, 0), "Total updates: ", self.it))

    def get_mean(self, num=10):
        return self._mean(num, 0)

    def get_mean_last(self, num=10):
        return self._mean(num, num)

    def get_raw_data(self):
        return self.vals


def save_lists(mode, config, reward_list, train_steps_needed, episode_length_needed, env_reward_overview, experiment_name=None, out_dir=None):
    """The harness's result file `<mode>_<experiment_name>.pt` (reference utils.py:144-160: same keys, the overview as a DataFrame with one
    row per model / real-env repetition)."""
    import os
    import pandas as pd
    if experiment_name is None:
        experiment_name = "_experiment_"
    if out_dir is None:
        out_dir = os.getcwd()
    file_name = os.path.join(out_dir, str(mode) + '_' + experiment_name + '.pt')
    save_dict = {'config': config, 'reward_list': reward_list, 'train_steps_needed': train_steps_needed,
                 'episode_length_needed': episode_length_needed,
                 'env_reward_overview': pd.DataFrame.from_dict(env_reward_overview, orient="index")}
    torch.save(save_dict, file_name)
    return file_name
