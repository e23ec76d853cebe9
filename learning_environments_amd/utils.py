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
    """reference utils.py:75-105 (host-side bookkeeping; the device kernels implement `_mean` themselves)."""

    def __init__(self, print_str):
        self.print_str = print_str
        self.vals = []
        self.it = 0

    def update(self, val, print_rate=10):
        if torch.is_tensor(val):
            val = val.item()
        self.vals.append(val)
        self.it += 1
        if self.it % print_rate == 0:
            mean_val = self._mean(num=print_rate, ignore_last=0)
            print(self.print_str + "{:15.6f} {:>25} {}".format(mean_val, "Total updates: ", self.it))

    def get_mean(self, num=10):
        return self._mean(num, ignore_last=0)

    def get_mean_last(self, num=10):
        return self._mean(num, ignore_last=num)

    def get_raw_data(self):
        return self.vals

    def _mean(self, num, ignore_last):
        vals = self.vals[max(len(self.vals) - num - ignore_last, 0): max(len(self.vals) - ignore_last, 0)]
        return sum(vals) / (len(vals) + 1e-9)
