"""Scalar logger with the tensorboardX `add_scalar` surface (reference: SummaryWriter in
lib/agent/a2c_base.py:263-267).  tensorboardX is not a dependency of this build: use
torch.utils.tensorboard when the `tensorboard` package is importable, otherwise append JSON lines."""
import json
import os


class JsonlWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "scalars.jsonl")
        self.f = open(self.path, "a")

    def add_scalar(self, tag, value, step):
        self.f.write(json.dumps({"tag": tag, "value": float(value), "step": float(step)}) + "\n")

    def flush(self):
        self.f.flush()

    def close(self):
        self.f.close()


def make_writer(logdir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(logdir)
    except Exception:
        return JsonlWriter(logdir)
