// ppo_loss_math.hpp - the per-row arithmetic of the fused PPO loss (lib/agent/a2c_continuous.py:299-369 calc_gradients;
// lib/core/common_losses.py:10-20,39-48; a2c_continuous.py:382-390 bound_loss; lib/model/a2c_continuous_logstd_model.py:195-198
// neglogp; lib/core/torch_ext.py:27-36 policy_kl, :168-178 policy_clip_fraction), shared by ppo_kernels.hip (ag_ppo_loss) and
// split_gemm.hip (ag_split_gemm_loss_heads_bwd: the same rows inside the last hidden layer's GEMM epilogue) so that both
// evaluate the SAME expressions.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/airgym_hip.h"

namespace agloss {

constexpr int kDLogstd0 = 4;                              // a, c, b, kl, dlogstd[<=5], dbias_heads[<=6]
constexpr int kDBias0 = 4 + AG_MAX_ACTIONS;
constexpr int kClipCount = 4 + AG_MAX_ACTIONS + AG_MAX_ACTIONS + 1;   // rows whose ratio left [1 - e_clip, 1 + e_clip]
constexpr int kNumSums = kClipCount + 1;

struct LossParams {
    float e_clip, critic_coef, bounds_loss_coef, inv_m;
    int clip_value, bound_type;  // bound_type: 0 none, 1 'bound' (soft limit 1.1), 2 'regularisation'
};

// per-launch constants of the state-independent sigma
template <int A>
struct LossConsts {
    float sig[A], inv_sig[A], logstd_sum, half_log_2pi_a, log_lo, log_hi;
};

template <int A>
__device__ __forceinline__ void loss_consts(const float* __restrict__ logstd, float e_clip, LossConsts<A>& c) {
    c.logstd_sum = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float ls = logstd[a];
        c.sig[a] = expf(ls);
        c.inv_sig[a] = 1.0f / c.sig[a];
        c.logstd_sum += ls;
    }
    c.half_log_2pi_a = 0.5f * 1.8378770664093453f * (float)A;
    // policy_clip_fraction (lib/core/torch_ext.py:168-178): logratio outside [log(1 - e), log(1 + e)]
    c.log_lo = logf(1.0f - e_clip);
    c.log_hi = logf(1.0f + e_clip);
}

// One minibatch row.  h = the row's head outputs (mu[0..A-1], value); act / old_mu / old_sigma = the row's A-vectors.
// dh[0..A] <- d(total loss)/d(head outputs) already scaled by 1/M; acc[kNumSums] += the row's contributions to the sums.
// SERIAL: the per-action parts are kept apart (scheduling fences): the compiler otherwise interleaves the five logf / expf
// expansions for instruction-level parallelism, which costs ~150 live registers - fine in a kernel of its own, not in a GEMM
// epilogue that holds 128 accumulator registers.  Same operations either way.
template <int A, bool SERIAL = false>
__device__ __forceinline__ void loss_row(const float* h, const float* act, float old_neglogp, float adv, float ret, float old_value,
                                         const float* old_mu, const float* old_sigma, const LossConsts<A>& c, const LossParams& k,
                                         float* dh, float* acc) {
    // no FMA contraction in here: the KL term is a difference of O(1) quantities that leaves O(1e-5); whether `a * a + b * b`
    // became an fma used to depend on the kernel the function was inlined into (0.6 % of the minibatch KL between the two).
    // Unfused, both evaluate what the reference's eager torch ops evaluate (torch_ext.py:27-36).
#pragma clang fp contract(off)
    float mu[A], z[A];
    float q = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        mu[a] = h[a];
        z[a] = (act[a] - mu[a]) * c.inv_sig[a];
        q += z[a] * z[a];
    }
    const float v = h[A];
    const float nlp = 0.5f * q + c.half_log_2pi_a + c.logstd_sum;
    const float logratio = old_neglogp - nlp;
    if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    const float ratio = expf(logratio);
    if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    acc[kClipCount] += (logratio < c.log_lo || logratio > c.log_hi) ? 1.0f : 0.0f;
    const float lo = 1.0f - k.e_clip, hi = 1.0f + k.e_clip;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float l1 = -adv * ratio, l2 = -adv * rc;
    const float a_loss = fmaxf(l1, l2);
    // d a / d ratio with torch.max's tie rule (equal -> half to each branch); clamp passes the
    // gradient on the closed interval [lo, hi]
    const float in_range = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float w1 = (l1 > l2) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
    const float w2 = (l2 > l1) ? 1.0f : ((l1 == l2) ? 0.5f : 0.0f);
    const float da_dratio = -adv * (w1 + w2 * in_range);
    const float da_dnlp = da_dratio * (-ratio);      // d ratio / d nlp = -ratio
    // value loss (common_losses.py:10-20)
    float c_loss, dc_dv;
    if (k.clip_value) {
        const float vp = old_value;
        const float dvc = fminf(fmaxf(v - vp, -k.e_clip), k.e_clip);
        const float vc = vp + dvc;
        const float u1 = (v - ret) * (v - ret), u2 = (vc - ret) * (vc - ret);
        c_loss = fmaxf(u1, u2);
        const float pass = ((v - vp) >= -k.e_clip && (v - vp) <= k.e_clip) ? 1.0f : 0.0f;
        const float g1 = 2.0f * (v - ret), g2 = 2.0f * (vc - ret) * pass;
        dc_dv = (u1 > u2) ? g1 : ((u1 == u2) ? 0.5f * (g1 + g2) : g2);
    } else {
        c_loss = (ret - v) * (ret - v);
        dc_dv = 2.0f * (v - ret);
    }
    float b_loss = 0.0f, kl = 0.0f;
    if (SERIAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < A; ++a) {
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        // d nlp / d mu_a = -z_a / sigma_a ;  d nlp / d logstd_a = 1 - z_a^2
        float dmu = da_dnlp * (-z[a] * c.inv_sig[a]);
        acc[kDLogstd0 + a] += da_dnlp * (1.0f - z[a] * z[a]);
        if (k.bound_type == 1) {
            const float hi_v = fmaxf(mu[a] - 1.1f, 0.0f), lo_v = fminf(mu[a] + 1.1f, 0.0f);
            b_loss += lo_v * lo_v + hi_v * hi_v;
            dmu += k.bounds_loss_coef * 2.0f * (hi_v + lo_v);
        } else if (k.bound_type == 2) {
            b_loss += mu[a] * mu[a];
            dmu += k.bounds_loss_coef * 2.0f * mu[a];
        }
        dh[a] = dmu * k.inv_m;
        acc[kDBias0 + a] += dmu * k.inv_m;
        // KL(p0 = new || p1 = old), torch_ext.py:27-36
        const float s1 = old_sigma[a], m1 = old_mu[a];
        const float dm = m1 - mu[a];
        kl += logf(s1 * c.inv_sig[a] + 1e-5f) + (c.sig[a] * c.sig[a] + dm * dm) / (2.0f * (s1 * s1 + 1e-5f)) - 0.5f;
    }
    dh[A] = 0.5f * k.critic_coef * dc_dv * k.inv_m;
    acc[kDBias0 + A] += dh[A];
    acc[0] += a_loss;
    acc[1] += c_loss;
    acc[2] += b_loss;
    acc[3] += kl;
}

}  // namespace agloss
