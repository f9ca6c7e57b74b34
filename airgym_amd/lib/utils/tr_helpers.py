"""DefaultRewardsShaper (reference: lib/utils/tr_helpers.py:16-42): shift, scale, clip, optional log."""
import math

import torch


class DefaultRewardsShaper:
    def __init__(self, scale_value=1, shift_value=0, min_val=-math.inf, max_val=math.inf, log_val=False, is_torch=True):
        self.scale_value = scale_value
        self.shift_value = shift_value
        self.min_val = min_val
        self.max_val = max_val
        self.log_val = log_val

    def __call__(self, reward):
        reward = (reward + self.shift_value) * self.scale_value
        if self.min_val != -math.inf or self.max_val != math.inf:
            reward = torch.clamp(reward, self.min_val, self.max_val)
        if self.log_val:
            reward = torch.log(reward)
        return reward
