"""Restatement of the reference's PPO numerics (oracle; test infrastructure).

Follows /root/reference/lib:
  actor_loss            lib/core/common_losses.py:39-48
  critic_loss           lib/core/common_losses.py:10-20
  bound_loss            lib/agent/a2c_continuous.py:382-390
  policy_kl             lib/core/torch_ext.py:27-36
  neglogp               lib/model/a2c_continuous_logstd_model.py:195-198
  RunningMeanStd        lib/core/running_mean_std.py:8-81  (float64 statistics, clamp +-5)
  AdaptiveScheduler     lib/core/schedulers.py:19-32
  GAE                   lib/agent/a2c_base.py:463-478
  MLP forward           lib/network/mlp.py:36-39 (activation after every layer)
Pinned by tests/golden/ppo.npz and gae.npz (outputs of the reference's own code).
"""
import math

import torch


def actor_loss(old_neglogp, neglogp, advantage, e_clip):
    ratio = torch.exp(old_neglogp - neglogp)
    surr1 = advantage * ratio
    surr2 = advantage * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    return torch.max(-surr1, -surr2)


def critic_loss(value_preds, values, e_clip, returns, clip_value):
    if clip_value:
        value_pred_clipped = value_preds + (values - value_preds).clamp(-e_clip, e_clip)
        return torch.max((values - returns) ** 2, (value_pred_clipped - returns) ** 2)
    return (returns - values) ** 2


def bound_loss(mu, soft_bound=1.1):
    mu_loss_high = torch.clamp_min(mu - soft_bound, 0.0) ** 2
    mu_loss_low = torch.clamp_max(mu + soft_bound, 0.0) ** 2
    return (mu_loss_low + mu_loss_high).sum(axis=-1)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    kl = (c1 + c2 - 0.5).sum(dim=-1)
    return kl.mean() if reduce else kl


def neglogp(x, mean, std, logstd):
    return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) \
        + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)


class RunningMeanStdRef:
    def __init__(self, insize, epsilon=1e-05):
        self.epsilon = epsilon
        self.running_mean = torch.zeros(insize, dtype=torch.float64)
        self.running_var = torch.ones(insize, dtype=torch.float64)
        self.count = torch.ones((), dtype=torch.float64)

    def update(self, x):
        mean = x.mean(0)
        var = x.var(0)
        batch_count = x.size()[0]
        delta = mean - self.running_mean
        tot = self.count + batch_count
        new_mean = self.running_mean + delta * batch_count / tot
        m2 = self.running_var * self.count + var * batch_count + delta ** 2 * self.count * batch_count / tot
        self.running_mean, self.running_var, self.count = new_mean, m2 / tot, tot

    def normalize(self, x):
        y = (x - self.running_mean.float()) / torch.sqrt(self.running_var.float() + self.epsilon)
        return torch.clamp(y, min=-5.0, max=5.0)

    def denormalize(self, x):
        y = torch.clamp(x, min=-5.0, max=5.0)
        return torch.sqrt(self.running_var.float() + self.epsilon) * y + self.running_mean.float()


def adaptive_lr(current_lr, kl_dist, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2):
    lr = current_lr
    if kl_dist > (2.0 * kl_threshold):
        lr = max(current_lr / 1.5, min_lr)
    if kl_dist < (0.5 * kl_threshold):
        lr = min(current_lr * 1.5, max_lr)
    return lr


def gae(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma, tau):
    horizon = mb_rewards.shape[0]
    lastgaelam = 0
    mb_advs = torch.zeros_like(mb_rewards)
    for t in reversed(range(horizon)):
        if t == horizon - 1:
            nextnonterminal = 1.0 - fdones
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - mb_fdones[t + 1]
            nextvalues = mb_values[t + 1]
        nextnonterminal = nextnonterminal.unsqueeze(1)
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        mb_advs[t] = lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam
    return mb_advs


def mlp_forward(x, weights, biases, activation=torch.nn.functional.elu):
    for w, b in zip(weights, biases):
        x = activation(torch.nn.functional.linear(x, w, b))
    return x
