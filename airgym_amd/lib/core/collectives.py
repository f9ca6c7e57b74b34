"""Every data-path collective of the PPO job goes through these two functions, which COUNT what they issue - so that
bench.py's `rccl` object can state, per epoch, the collectives the job really ran (gradient all-reduces, normaliser-moment
all-reduces, ...) instead of deriving the number (reference collectives: lib/agent/a2c_base.py:293-309,348-352,
lib/agent/a2c_continuous.py:111-123,188-192)."""
import collections

import torch.distributed as dist

COUNTS = collections.Counter()       # tag -> calls issued by this process since the last reset()
BYTES = collections.Counter()        # tag -> payload bytes


def reset():
    COUNTS.clear()
    BYTES.clear()


def snapshot():
    return {k: {"calls": int(v), "bytes": int(BYTES[k])} for k, v in sorted(COUNTS.items())}


def count(tag, nbytes, calls=1):
    """Account for collectives that were issued without passing through all_reduce(): the replay of a hipGraph that CONTAINS
    a captured all-reduce issues it on the device each time - the caller counts it here, once per replay."""
    COUNTS[tag] += calls
    BYTES[tag] += nbytes * calls


def all_reduce(tensor, tag, op=None, group=None, counted=True):
    """counted=False: inside a graph capture (the capture pass itself moves no data; replays are counted with count())."""
    if counted:
        COUNTS[tag] += 1
        BYTES[tag] += tensor.numel() * tensor.element_size()
    return dist.all_reduce(tensor, op=op if op is not None else dist.ReduceOp.SUM, group=group)


def broadcast(tensor, src, tag, group=None):
    COUNTS[tag] += 1
    BYTES[tag] += tensor.numel() * tensor.element_size()
    return dist.broadcast(tensor, src, group=group)
