// ppo_kernels.hip - fused PPO minibatch loss for the continuous (Gaussian, state-independent sigma) policy.
//
// One pass over the minibatch computes what the reference builds from ~100 eager elementwise kernels
// (lib/agent/a2c_continuous.py:299-369 calc_gradients; lib/core/common_losses.py:10-20,39-48;
// a2c_continuous.py:382-390 bound_loss; lib/model/a2c_continuous_logstd_model.py:195-198 neglogp;
// lib/core/torch_ext.py:27-36 policy_kl):
//   per row i:  neglogp, ratio, clipped surrogate a_i, value loss c_i, bound loss b_i, KL(new || old)
//   d(total loss)/d(head outputs)  [M, A+1]  (mu columns then the value column), already scaled by 1/M
//   per-block partial sums of {a, c, b, kl, d logstd_0..A-1}  (reduced deterministically by the caller)
// and writes the new (mu, sigma) rows back to the dataset (PPODataset.update_mu_sigma, datasets.py:20-24).
// Memory-bound: ~(A+1 + 3A + 4 + A+1 + 2A) floats per row.
#include <hip/hip_runtime.h>

#include "../../include/airgym_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kNumSums = 4 + AG_MAX_ACTIONS;  // a, c, b, kl, dlogstd[<=5]

struct LossArgs {
    const float* heads;      // [M, A+1]
    const float* logstd;     // [A]
    const float* actions;    // [M, A]
    const float* old_neglogp;
    const float* advantages;
    const float* returns;
    const float* old_values;
    const float* old_mu;     // [M, A]
    const float* old_sigma;  // [M, A]
    float* d_heads;          // [M, A+1]
    float* new_mu;           // [M, A] or null
    float* new_sigma;        // [M, A] or null
    float* partials;         // [gridDim.x, kNumSums]
    int M;
    float e_clip, critic_coef, bounds_loss_coef, inv_m;
    int clip_value, bound_type;  // bound_type: 0 none, 1 'bound' (soft limit 1.1), 2 'regularisation'
};

template <int A>
__global__ __launch_bounds__(kBlock) void ppo_loss_kernel(const LossArgs k) {
    __shared__ float red[kBlock / 64][kNumSums];
    float acc[kNumSums];
#pragma unroll
    for (int j = 0; j < kNumSums; ++j) acc[j] = 0.0f;
    float ls[A], sig[A], inv_sig[A];
    float logstd_sum = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        ls[a] = k.logstd[a];
        sig[a] = expf(ls[a]);
        inv_sig[a] = 1.0f / sig[a];
        logstd_sum += ls[a];
    }
    const float half_log_2pi_a = 0.5f * 1.8378770664093453f * (float)A;

    for (int i = blockIdx.x * kBlock + threadIdx.x; i < k.M; i += gridDim.x * kBlock) {
        const float* h = k.heads + (size_t)i * (A + 1);
        float mu[A], z[A];
        float q = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            mu[a] = h[a];
            z[a] = (k.actions[(size_t)i * A + a] - mu[a]) * inv_sig[a];
            q += z[a] * z[a];
        }
        const float v = h[A];
        const float nlp = 0.5f * q + half_log_2pi_a + logstd_sum;
        const float adv = k.advantages[i];
        const float ratio = expf(k.old_neglogp[i] - nlp);
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
        const float ret = k.returns[i];
        float c_loss, dc_dv;
        if (k.clip_value) {
            const float vp = k.old_values[i];
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
        float* dh = k.d_heads + (size_t)i * (A + 1);
#pragma unroll
        for (int a = 0; a < A; ++a) {
            // d nlp / d mu_a = -z_a / sigma_a ;  d nlp / d logstd_a = 1 - z_a^2
            float dmu = da_dnlp * (-z[a] * inv_sig[a]);
            acc[4 + a] += da_dnlp * (1.0f - z[a] * z[a]);
            if (k.bound_type == 1) {
                const float hi_v = fmaxf(mu[a] - 1.1f, 0.0f), lo_v = fminf(mu[a] + 1.1f, 0.0f);
                b_loss += lo_v * lo_v + hi_v * hi_v;
                dmu += k.bounds_loss_coef * 2.0f * (hi_v + lo_v);
            } else if (k.bound_type == 2) {
                b_loss += mu[a] * mu[a];
                dmu += k.bounds_loss_coef * 2.0f * mu[a];
            }
            dh[a] = dmu * k.inv_m;
            // KL(p0 = new || p1 = old), torch_ext.py:27-36
            const float s1 = k.old_sigma[(size_t)i * A + a], m1 = k.old_mu[(size_t)i * A + a];
            const float dm = m1 - mu[a];
            kl += logf(s1 * inv_sig[a] + 1e-5f) + (sig[a] * sig[a] + dm * dm) / (2.0f * (s1 * s1 + 1e-5f)) - 0.5f;
            if (k.new_mu) {
                k.new_mu[(size_t)i * A + a] = mu[a];
                k.new_sigma[(size_t)i * A + a] = sig[a];
            }
        }
        dh[A] = 0.5f * k.critic_coef * dc_dv * k.inv_m;
        acc[0] += a_loss;
        acc[1] += c_loss;
        acc[2] += b_loss;
        acc[3] += kl;
    }
    // block reduction: wave shuffle, then LDS across the 4 waves
#pragma unroll
    for (int j = 0; j < kNumSums; ++j) {
        float x = acc[j];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        acc[j] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kNumSums; ++j) red[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < kNumSums) {
        float x = 0.0f;
        for (int w = 0; w < kBlock / 64; ++w) x += red[w][threadIdx.x];
        k.partials[(size_t)blockIdx.x * kNumSums + threadIdx.x] = x;
    }
}

}  // namespace

extern "C" int ag_ppo_loss_num_sums(void) { return kNumSums; }

extern "C" int ag_ppo_loss_max_blocks(void) { return 2048; }

extern "C" int ag_ppo_loss(const float* heads, const float* logstd, const float* actions, const float* old_neglogp,
                           const float* advantages, const float* returns, const float* old_values, const float* old_mu,
                           const float* old_sigma, int M, int A, float e_clip, float critic_coef, float bounds_loss_coef,
                           int clip_value, int bound_type, float* d_heads, float* new_mu, float* new_sigma,
                           float* partials, int* num_blocks_out, void* stream) {
    if (!heads || !logstd || !actions || !old_neglogp || !advantages || !returns || !old_values || !old_mu ||
        !old_sigma || !d_heads || !partials || !num_blocks_out || M <= 0)
        return AG_ERR_INVALID_ARG;
    if ((new_mu == nullptr) != (new_sigma == nullptr)) return AG_ERR_INVALID_ARG;
    LossArgs k{heads, logstd, actions, old_neglogp, advantages, returns, old_values, old_mu, old_sigma,
               d_heads, new_mu, new_sigma, partials, M, e_clip, critic_coef, bounds_loss_coef, 1.0f / (float)M,
               clip_value, bound_type};
    int grid = (M + kBlock - 1) / kBlock;
    if (grid > 2048) grid = 2048;
    *num_blocks_out = grid;
    if (A == 4)
        hipLaunchKernelGGL(ppo_loss_kernel<4>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, k);
    else if (A == 5)
        hipLaunchKernelGGL(ppo_loss_kernel<5>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, k);
    else
        return AG_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
