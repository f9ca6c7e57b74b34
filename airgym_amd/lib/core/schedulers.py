"""LR schedulers (reference: lib/core/schedulers.py).  AdaptiveScheduler additionally has a
device-side form so the KL-driven LR update needs no host sync per minibatch."""
import torch


class RLScheduler:
    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        return current_lr, entropy_coef


class IdentityScheduler(RLScheduler):
    pass


class AdaptiveScheduler(RLScheduler):
    def __init__(self, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2):
        # bounds: the reference hard-codes [1e-6, 1e-2] (lib/core/schedulers.py:19-23); here they are also YAML keys
        # (`min_lr`, `max_lr` beside `kl_threshold`), defaults unchanged
        self.min_lr = float(min_lr)
        self.max_lr = float(max_lr)
        self.kl_threshold = kl_threshold

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        lr = current_lr
        if kl_dist > (2.0 * self.kl_threshold):
            lr = max(current_lr / 1.5, self.min_lr)
        if kl_dist < (0.5 * self.kl_threshold):
            lr = min(current_lr * 1.5, self.max_lr)
        return lr, entropy_coef

    def update_tensor_(self, lr, kl):
        """Same rule on 0-dim device tensors, in place on `lr` (float64 like the host arithmetic)."""
        down = torch.clamp(lr / 1.5, min=self.min_lr)
        up = torch.clamp(lr * 1.5, max=self.max_lr)
        new = torch.where(kl > 2.0 * self.kl_threshold, down, lr)
        new = torch.where(kl < 0.5 * self.kl_threshold, torch.where(kl > 2.0 * self.kl_threshold, torch.clamp(down * 1.5, max=self.max_lr), up), new)
        lr.copy_(new)
        return lr


class LinearScheduler(RLScheduler):
    def __init__(self, start_lr, min_lr=1e-6, max_steps=1000000, use_epochs=True, apply_to_entropy=False, **kwargs):
        self.start_lr = start_lr
        self.min_lr = min_lr
        self.max_steps = max_steps
        self.use_epochs = use_epochs
        self.apply_to_entropy = apply_to_entropy
        if apply_to_entropy:
            self.start_entropy_coef = kwargs.pop("start_entropy_coef", 0.01)
            self.min_entropy_coef = kwargs.pop("min_entropy_coef", 0.0001)

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        steps = epoch if self.use_epochs else frames
        mul = max(0, self.max_steps - steps) / self.max_steps
        lr = self.min_lr + (self.start_lr - self.min_lr) * mul
        if self.apply_to_entropy:
            entropy_coef = self.min_entropy_coef + (self.start_entropy_coef - self.min_entropy_coef) * mul
        return lr, entropy_coef
