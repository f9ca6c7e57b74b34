"""Small torch helpers (reference: lib/core/torch_ext.py): Gaussian KL, list mean, episode-stat meter,
checkpoint save/load with retry."""
import os
import time

import numpy as np
import torch


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    """torch_ext.py:27-36"""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    kl = (c1 + c2 - 0.5).sum(dim=-1)
    return kl.mean() if reduce else kl


def mean_list(val):
    return torch.mean(torch.stack(val))


def safe_filesystem_op(func, *args, **kwargs):
    """5 attempts with 2^k s back-off (torch_ext.py:51-72)."""
    num_attempts = 5
    for attempt in range(num_attempts):
        try:
            return func(*args, **kwargs)
        except Exception as exc:
            print(f"Exception {exc} when trying to execute {func} with args:{args} and kwargs:{kwargs}...")
            wait_sec = 2 ** attempt
            print(f"Waiting {wait_sec} before trying again...")
            time.sleep(wait_sec)
    raise RuntimeError(f"Could not execute {func}, give up after {num_attempts} attempts...")


def save_checkpoint(filename, state):
    print("=> saving checkpoint '{}'".format(filename + ".pth"))
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    safe_filesystem_op(torch.save, state, filename + ".pth")


def load_checkpoint(filename):
    print("=> loading checkpoint '{}'".format(filename))
    # reference checkpoints pickle numpy scalars -> weights_only=False
    return safe_filesystem_op(torch.load, filename, map_location="cpu", weights_only=False)


class AverageMeter:
    """Mean over (at most) the last `max_size` finished episodes (torch_ext.py:270-296); host-side,
    fed once per epoch from device-side per-step sums (no per-step nonzero() sync)."""

    def __init__(self, in_shape, max_size):
        self.max_size = max_size
        self.current_size = 0
        self.mean = np.zeros(in_shape, dtype=np.float64)

    def update_from_sum(self, value_sum, count):
        """Equivalent to update(values) with values.mean = value_sum / count, len(values) = count."""
        size = int(count)
        if size == 0:
            return
        new_mean = np.asarray(value_sum, dtype=np.float64) / size
        size = int(np.clip(size, 0, self.max_size))
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = (self.mean * old_size + new_mean * size) / size_sum

    def update_from_sums(self, pairs):
        """update_from_sum for a sequence of (value_sum, count) pairs of a one-wide meter, in plain float arithmetic (the same
        IEEE double operations in the same order: bit-identical) - the per-call numpy overhead of 72 calls per epoch was 0.35 ms
        during which the GPU had nothing queued (`profiles/r05_epoch_gaps.md`)."""
        mean, cur, cap = float(self.mean[0]), self.current_size, self.max_size
        for value_sum, count in pairs:
            size = int(count)
            if size == 0:
                continue
            new_mean = float(value_sum) / size
            size = min(max(size, 0), cap)
            old_size = min(cap - size, cur)
            cur = old_size + size
            mean = (mean * old_size + new_mean * size) / cur
        self.current_size = cur
        self.mean = np.full_like(self.mean, mean)

    def clear(self):
        self.current_size = 0
        self.mean.fill(0)

    def __len__(self):
        return self.current_size

    def get_mean(self):
        return self.mean


def explained_variance(y_pred, y):
    """1 - Var[y - y_pred] / Var[y] (lib/core/torch_ext.py:149-166, masks=None branch; unbiased variances)."""
    return 1.0 - torch.var(y - y_pred) / torch.var(y)


def policy_clip_fraction(new_neglogp, old_neglogp, clip_param):
    """Fraction of samples whose probability ratio left [1 - clip, 1 + clip] (lib/core/torch_ext.py:168-178)."""
    import math
    logratio = old_neglogp - new_neglogp
    return torch.logical_or(logratio < math.log(1.0 - clip_param), logratio > math.log(1.0 + clip_param)).float().mean()
