"""PPO losses (reference: lib/core/common_losses.py:10-48, lib/agent/a2c_continuous.py:372-390)."""
import torch


def critic_loss(value_preds_batch, values, curr_e_clip, return_batch, clip_value):
    if clip_value:
        value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-curr_e_clip, curr_e_clip)
        return torch.max((values - return_batch) ** 2, (value_pred_clipped - return_batch) ** 2)
    return (return_batch - values) ** 2


def actor_loss(old_action_neglog_probs_batch, action_neglog_probs, advantage, is_ppo, curr_e_clip):
    if is_ppo:
        ratio = torch.exp(old_action_neglog_probs_batch - action_neglog_probs)
        surr1 = advantage * ratio
        surr2 = advantage * torch.clamp(ratio, 1.0 - curr_e_clip, 1.0 + curr_e_clip)
        return torch.max(-surr1, -surr2)
    return action_neglog_probs * advantage


def bound_loss(mu, soft_bound=1.1):
    mu_loss_high = torch.clamp_min(mu - soft_bound, 0.0) ** 2
    mu_loss_low = torch.clamp_max(mu + soft_bound, 0.0) ** 2
    return (mu_loss_low + mu_loss_high).sum(axis=-1)


def reg_loss(mu):
    return (mu * mu).sum(axis=-1)
