"""Minimal stand-ins for gym.spaces.Box / Dict.  The reference only reads .shape/.low/.high/.spaces
(lib/agent/a2c_base.py:205-210, lib/agent/a2c_continuous.py:41-48, lib/core/experience.py:301-313) and gym
0.23.1 is not a dependency of this build."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        self.low, self.high = low, high
        self.shape = tuple(low.shape)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def items(self):
        return self.spaces.items()
