"""Autograd wrapper of the fused PPO-loss kernel (`ag_ppo_loss`, airgym_amd/csrc/ppo_kernels.hip).

forward: one launch computes every per-row term of calc_gradients (a2c_continuous.py:299-369) plus
d loss / d heads; the per-block partial sums are reduced with one deterministic `sum(0)`.
backward: returns the stored gradients (scaled by the incoming grad), so autograd continues into the
head GEMM and the MLP trunk exactly as with the composed torch ops.
"""
import ctypes
import math

import torch

from airgym_amd import _native as N

_HALF_LOG_2PI_P_HALF = 0.5 + 0.5 * math.log(2.0 * math.pi)
BOUND_TYPES = {None: 0, "none": 0, "bound": 1, "regularisation": 2}


class _FusedPPOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heads, logstd, actions, old_neglogp, advantages, returns, old_values, old_mu, old_sigma,
                e_clip, critic_coef, entropy_coef, bounds_loss_coef, clip_value, bound_type, write_back):
        lib = N.load()
        M, A1 = heads.shape
        A = A1 - 1
        for t in (heads, actions, old_neglogp, advantages, returns, old_values, old_mu, old_sigma):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "fused PPO loss needs contiguous f32 CUDA tensors"
        assert actions.shape == (M, A) and old_mu.shape == (M, A) and old_sigma.shape == (M, A)
        assert old_neglogp.numel() == M and advantages.numel() == M and returns.numel() == M and old_values.numel() == M
        logstd_c = logstd.detach().contiguous()
        d_heads = torch.empty_like(heads)
        nsums = lib.ag_ppo_loss_num_sums()
        partials = torch.empty(lib.ag_ppo_loss_max_blocks(), nsums, dtype=torch.float32, device=heads.device)
        nb = ctypes.c_int(0)
        stream = ctypes.c_void_p(torch.cuda.current_stream(heads.device).cuda_stream)
        wb_mu = old_mu.data_ptr() if write_back else None
        wb_sigma = old_sigma.data_ptr() if write_back else None
        rc = lib.ag_ppo_loss(heads.data_ptr(), logstd_c.data_ptr(), actions.data_ptr(), old_neglogp.data_ptr(),
                             advantages.data_ptr(), returns.data_ptr(), old_values.data_ptr(), old_mu.data_ptr(),
                             old_sigma.data_ptr(), M, A, float(e_clip), float(critic_coef),
                             float(bounds_loss_coef or 0.0), int(bool(clip_value)), int(bound_type),
                             d_heads.data_ptr(), wb_mu, wb_sigma, partials.data_ptr(), ctypes.byref(nb), stream)
        N.check(rc, "ag_ppo_loss")
        sums = partials[:nb.value].sum(0) / float(M)
        a_loss, c_loss, b_loss, kl = sums[0], sums[1], sums[2], sums[3]
        entropy = (_HALF_LOG_2PI_P_HALF + logstd_c).sum()
        d_logstd = sums[4:4 + A] - float(entropy_coef)
        loss = a_loss + 0.5 * c_loss * critic_coef - entropy * entropy_coef + b_loss * float(bounds_loss_coef or 0.0)
        ctx.save_for_backward(d_heads, d_logstd)
        # sums[-1] = rows whose probability ratio left [1 - e_clip, 1 + e_clip] / M = PpoDiagnostics' clip fraction
        # (lib/core/dignostics.py:49-59; torch_ext.policy_clip_fraction :168-178)
        stats = torch.stack((a_loss, c_loss, entropy, b_loss, kl, sums[nsums - 1]))
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, g_stats):
        d_heads, d_logstd = ctx.saved_tensors
        return (g_loss * d_heads, g_loss * d_logstd) + (None,) * 14


def fused_ppo_loss(heads, logstd, actions, old_neglogp, advantages, returns, old_values, old_mu, old_sigma, *,
                   e_clip, critic_coef, entropy_coef, bounds_loss_coef, clip_value, bound_loss_type, write_back=True):
    """-> (loss, stats[a_loss, c_loss, entropy, b_loss, kl, clip_frac]).  With write_back the rows of old_mu / old_sigma are
    overwritten by the current policy after being read (PPODataset.update_mu_sigma)."""
    bt = BOUND_TYPES[bound_loss_type] if bounds_loss_coef is not None else 0
    return _FusedPPOLossFn.apply(heads, logstd, actions, old_neglogp, advantages, returns, old_values, old_mu,
                                 old_sigma, e_clip, critic_coef, entropy_coef, bounds_loss_coef, clip_value, bt,
                                 write_back)
