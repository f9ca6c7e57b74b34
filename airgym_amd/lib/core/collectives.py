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


def all_reduce(tensor, tag, op=None, group=None):
    COUNTS[tag] += 1
    BYTES[tag] += tensor.numel() * tensor.element_size()
    return dist.all_reduce(tensor, op=op if op is not None else dist.ReduceOp.SUM, group=group)


def broadcast(tensor, src, tag, group=None):
    COUNTS[tag] += 1
    BYTES[tag] += tensor.numel() * tensor.element_size()
    return dist.broadcast(tensor, src, group=group)
