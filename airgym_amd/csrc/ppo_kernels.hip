// ppo_kernels.hip - fused PPO minibatch loss for the continuous (Gaussian, state-independent sigma) policy.
//
// One pass over the minibatch computes what the reference builds from ~100 eager elementwise kernels
// (lib/agent/a2c_continuous.py:299-369 calc_gradients; lib/core/common_losses.py:10-20,39-48;
// a2c_continuous.py:382-390 bound_loss; lib/model/a2c_continuous_logstd_model.py:195-198 neglogp;
// lib/core/torch_ext.py:27-36 policy_kl):
//   per row i:  neglogp, ratio, clipped surrogate a_i, value loss c_i, bound loss b_i, KL(new || old)
//   d(total loss)/d(head outputs)  [M, A+1]  (mu columns then the value column), already scaled by 1/M
//   per-block partial sums of {a, c, b, kl, d logstd_0..A-1, column sums of d heads (= the head-bias gradient)}
//   (reduced deterministically by the caller or by ag_ppo_loss_finalize)
// and writes the new (mu, sigma) rows back to the dataset (PPODataset.update_mu_sigma, datasets.py:20-24).
// Memory-bound: ~(A+1 + 3A + 4 + A+1 + 2A) floats per row.
#include <hip/hip_runtime.h>

#include "../../include/airgym_hip.h"
#include "ppo_loss_math.hpp"

namespace {

constexpr int kBlock = 256;
using agloss::kClipCount;
using agloss::kDBias0;
using agloss::kDLogstd0;
using agloss::kNumSums;

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
    agloss::LossConsts<A> lc;
    agloss::loss_consts<A>(k.logstd, k.e_clip, lc);
    const agloss::LossParams lp{k.e_clip, k.critic_coef, k.bounds_loss_coef, k.inv_m, k.clip_value, k.bound_type};

    for (int i = blockIdx.x * kBlock + threadIdx.x; i < k.M; i += gridDim.x * kBlock) {
        float h[A + 1], act[A], om[A], os[A], dhr[A + 1];
#pragma unroll
        for (int a = 0; a <= A; ++a) h[a] = k.heads[(size_t)i * (A + 1) + a];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            act[a] = k.actions[(size_t)i * A + a];
            om[a] = k.old_mu[(size_t)i * A + a];
            os[a] = k.old_sigma[(size_t)i * A + a];
        }
        agloss::loss_row<A>(h, act, k.old_neglogp[i], k.advantages[i], k.returns[i], k.old_values[i], om, os, lc, lp, dhr, acc);
        float* dh = k.d_heads + (size_t)i * (A + 1);
#pragma unroll
        for (int a = 0; a <= A; ++a) dh[a] = dhr[a];
        if (k.new_mu) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                k.new_mu[(size_t)i * A + a] = h[a];
                k.new_sigma[(size_t)i * A + a] = lc.sig[a];
            }
        }
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


// Reduces the per-block partials of ppo_loss_kernel (fixed order -> deterministic) and writes everything the optimizer
// step needs straight into the flat gradient buffer: d loss / d logstd, the fused-head bias gradient, the minibatch KL
// (appended gradient element) and the logged scalars.  One workgroup; replaces ~15 tiny launches.
struct FinalizeArgs {
    const float* partials; int num_blocks; int M; int A;
    const float* logstd; float entropy_coef, critic_coef, bounds_loss_coef;
    float* grad_logstd; float* grad_head_bias; float* kl_out; float* stats;   // stats[8]: a, c, entropy, b, kl, loss, clip_frac, -
};

__device__ __forceinline__ void ppo_loss_finalize_block(const FinalizeArgs& k) {
    __shared__ float red[4][kNumSums];
    __shared__ float tot[kNumSums];
    float acc[kNumSums];
#pragma unroll
    for (int j = 0; j < kNumSums; ++j) acc[j] = 0.0f;
    for (int b = threadIdx.x; b < k.num_blocks; b += 256) {
#pragma unroll
        for (int j = 0; j < kNumSums; ++j) acc[j] += k.partials[(size_t)b * kNumSums + j];
    }
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
    if (threadIdx.x < kNumSums) tot[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_m = 1.0f / (float)k.M;
        const float a_loss = tot[0] * inv_m, c_loss = tot[1] * inv_m, b_loss = tot[2] * inv_m, kl = tot[3] * inv_m;
        float entropy = 0.0f;
        for (int a = 0; a < k.A; ++a) entropy += 0.5f + 0.5f * 1.8378770664093453f + k.logstd[a];
        for (int a = 0; a < k.A; ++a) k.grad_logstd[a] = tot[kDLogstd0 + a] * inv_m - k.entropy_coef;
        for (int a = 0; a <= k.A; ++a) k.grad_head_bias[a] = tot[kDBias0 + a];      // already scaled by 1/M
        *k.kl_out = kl;
        k.stats[0] = a_loss; k.stats[1] = c_loss; k.stats[2] = entropy; k.stats[3] = b_loss; k.stats[4] = kl;
        k.stats[5] = a_loss + 0.5f * c_loss * k.critic_coef - entropy * k.entropy_coef + b_loss * k.bounds_loss_coef;
        k.stats[6] = tot[kClipCount] * inv_m;      // diagnostics/clip_frac (lib/core/dignostics.py:55-59)
        k.stats[7] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void ppo_loss_finalize_kernel(const FinalizeArgs k) { ppo_loss_finalize_block(k); }

// Running-mean/std input normalisation (lib/core/running_mean_std.py:64-79): out = clamp((x - mean) / sqrt(var + eps),
// -clip, clip) with the statistics read from the float64 running buffers.  One pass instead of seven eager kernels.
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, const double* __restrict__ mean,
                                                             const double* __restrict__ var, float* __restrict__ out,
                                                             size_t total, int D, float eps, float clip) {
    extern __shared__ float stat[];      // mean[D] | std[D]
    for (int c = threadIdx.x; c < D; c += 256) {
        stat[c] = (float)mean[c];
        stat[D + c] = sqrtf((float)var[c] + eps);
    }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)D);
        const float y = (x[i] - stat[c]) / stat[D + c];
        out[i] = fminf(fmaxf(y, -clip), clip);
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

// ---------------------------------------------------------------------------------------------------
// ELU backward fused with the bias gradient of the Linear that produced the pre-activation.
//   dz[m,c] = dh[m,c] * (h[m,c] > 0 ? 1 : h[m,c] + 1)      (ELU alpha = 1, h = ELU(z): ELU'(z) = h + 1 for z <= 0)
//   db_partials[block, c] = sum over the block's rows of dz[m,c]
// Replaces elu_backward (read 2, write 1) + a separate column-sum pass (read 1) of the [M, C] gradient
// (lib/network/mlp.py:36-39 under autograd).  One pass: read dh, h; write dz.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kEluRowsPerBlock = 128;

__global__ __launch_bounds__(256) void elu_bwd_bias_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                           float* __restrict__ dz, float* __restrict__ db_partials,
                                                           int M, int C) {
    // thread layout: C/4 threads cover one row with float4; 256 / (C/4) rows are processed per pass
    __shared__ float4 red[256];
    const int tpr = C >> 2;                     // threads per row
    const int rpp = 256 / tpr;                  // rows per pass
    const int col4 = threadIdx.x % tpr;
    const int rsub = threadIdx.x / tpr;
    const int row0 = blockIdx.x * kEluRowsPerBlock;
    const int row_end = min(row0 + kEluRowsPerBlock, M);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < rpp) {
        for (int r = row0 + rsub; r < row_end; r += rpp) {
            const size_t idx = (size_t)r * tpr + col4;
            const float4 g = reinterpret_cast<const float4*>(dh)[idx];
            const float4 y = reinterpret_cast<const float4*>(h)[idx];
            float4 o;
            o.x = g.x * (y.x > 0.f ? 1.f : y.x + 1.f);
            o.y = g.y * (y.y > 0.f ? 1.f : y.y + 1.f);
            o.z = g.z * (y.z > 0.f ? 1.f : y.z + 1.f);
            o.w = g.w * (y.w > 0.f ? 1.f : y.w + 1.f);
            reinterpret_cast<float4*>(dz)[idx] = o;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < tpr) {
        float4 s = red[threadIdx.x];
        for (int j = 1; j < rpp; ++j) {
            const float4 t = red[threadIdx.x + j * tpr];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        reinterpret_cast<float4*>(db_partials)[(size_t)blockIdx.x * tpr + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// Gradient clip-by-norm + Adam + KL-adaptive learning rate in two 64-workgroup launches over the flat
// parameter buffer (trancate_gradients_and_step, lib/agent/a2c_base.py:293-316; AdaptiveScheduler,
// lib/core/schedulers.py:19-32; torch.optim.Adam update rule, eps outside the bias-corrected sqrt).
// state_dev: double[2] = {lr, step};  kl is read from grad[n] (the scalar appended to the flat gradient).
// Replaces ~35 tiny launches per optimizer step (a single-workgroup version took 104 us; this one 14 us).
// ---------------------------------------------------------------------------------------------------
struct AdamArgs {
    float* p; float* g; float* m; float* v;
    double* state;   // unused by the kernels (kept for symmetry with the C entry point)
    int n;
    float beta1, beta2, eps, weight_decay, max_grad_norm;   // max_grad_norm <= 0: no clipping
    float kl_threshold, min_lr, max_lr;                     // kl_threshold <= 0: LR not adapted
};

// Phase 1 of the optimizer step: kAdamBlocks per-block partial sums of g^2.  The grid covers the buffer in ONE pass where it can (one
// element per thread for up to kAdamBlocks x kAdamThreads = 65 536 parameters): with 64 x 256 threads striding 4 - 5 times over the 72 k
// parameters both phases were chains of dependent memory round trips (4.6 + 9.2 us for 0.3 MB; round 6).
constexpr int kAdamBlocks = 64;
constexpr int kAdamThreads = 1024;

// Block 0 also snapshots {lr, step} into `snap`: phase 2 reads the snapshot and publishes the new pair straight into the caller's
// slot (no block of phase 2 reads what block 0 of phase 2 writes) - the 16-byte device-to-device copy that used to follow is gone.
__global__ __launch_bounds__(kAdamThreads) void adam_norm_kernel(const float* __restrict__ g, int n, float* __restrict__ partial,
                                                                  const double* __restrict__ state, double* __restrict__ snap) {
    __shared__ float red[kAdamThreads / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) { snap[0] = state[0]; snap[1] = state[1]; }
    float ss = 0.f;
    for (int i = blockIdx.x * kAdamThreads + threadIdx.x; i < n; i += kAdamBlocks * kAdamThreads) {
        const float x = g[i];
        ss += x * x;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kAdamThreads / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

// Phase 2: every block re-reduces the 64 partials (deterministic order), reads {lr, step} from state_in (phase 1's snapshot),
// updates its slice; block 0 publishes {new lr, step + 1} to state_out (the caller's slot: no block reads what block 0 writes).
// The bias corrections (two double-precision pow) are formed by one thread per block and shared through LDS.
__global__ __launch_bounds__(kAdamThreads) void adam_clip_step_kernel(const AdamArgs k, const float* __restrict__ partial,
                                                                       const double* __restrict__ state_in,
                                                                       double* __restrict__ state_out) {
    __shared__ float sh[3];
    const double lr = state_in[0];
    const double step = state_in[1] + 1.0;
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < kAdamBlocks; ++w) tot += partial[w];
        const float norm = sqrtf(tot);
        sh[0] = (k.max_grad_norm > 0.f) ? fminf(k.max_grad_norm / (norm + 1e-6f), 1.0f) : 1.0f;
        const double bc1 = 1.0 - pow((double)k.beta1, step);
        const double bc2 = 1.0 - pow((double)k.beta2, step);
        sh[1] = (float)(lr / bc1);
        sh[2] = (float)(1.0 / sqrt(bc2));
        if (blockIdx.x == 0) {
            double nlr = lr;
            if (k.kl_threshold > 0.f) {      // legacy schedule: evaluated every minibatch, applies to the NEXT step
                const double kl = (double)k.g[k.n];
                if (kl > 2.0 * k.kl_threshold) nlr = fmax(lr / 1.5, (double)k.min_lr);
                if (kl < 0.5 * k.kl_threshold) nlr = fmin(lr * 1.5, (double)k.max_lr);
            }
            state_out[0] = nlr;
            state_out[1] = step;
        }
    }
    __syncthreads();
    const float coef = sh[0], step_size = sh[1], bc2r = sh[2];
    for (int i = blockIdx.x * kAdamThreads + threadIdx.x; i < k.n; i += gridDim.x * kAdamThreads) {
        float g = k.g[i] * coef;
        k.g[i] = g;
        const float p = k.p[i];
        if (k.weight_decay != 0.f) g += k.weight_decay * p;
        const float m = k.beta1 * k.m[i] + (1.f - k.beta1) * g;
        const float v = k.beta2 * k.v[i] + (1.f - k.beta2) * g * g;
        k.m[i] = m;
        k.v[i] = v;
        const float denom = sqrtf(v) * bc2r + k.eps;
        k.p[i] = p - step_size * (m / denom);
    }
}

}  // namespace

extern "C" int ag_elu_bwd_bias_rows_per_block(void) { return kEluRowsPerBlock; }

extern "C" int ag_elu_bwd_bias(const float* dh, const float* h, float* dz, float* db_partials, int M, int C, void* stream) {
    if (!dh || !h || !dz || !db_partials || M <= 0) return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > 1024 || (C & 3) || (256 % (C >> 2)) != 0) return AG_ERR_UNSUPPORTED;
    const int grid = (M + kEluRowsPerBlock - 1) / kEluRowsPerBlock;
    hipLaunchKernelGGL(elu_bwd_bias_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dh, h, dz, db_partials, M, C);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_adam_clip_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, double* state, int n,
                                 float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                                 float kl_threshold, float min_lr, float max_lr, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || n <= 0) return AG_ERR_INVALID_ARG;
    AdamArgs k{param, grad, exp_avg, exp_avg_sq, state, n, beta1, beta2, eps, weight_decay, max_grad_norm,
               kl_threshold, min_lr, max_lr};
    // state_dev layout: double[2 + 2 + 64 floats]: {lr, step} | snapshot of {lr, step} taken by phase 1 | 64 float partials
    double* snap = state + 2;
    float* partial = reinterpret_cast<float*>(state + 4);
    hipLaunchKernelGGL(adam_norm_kernel, dim3(kAdamBlocks), dim3(kAdamThreads), 0, (hipStream_t)stream, grad, n, partial,
                       (const double*)state, snap);
    const int upd_blocks = (n + kAdamThreads - 1) / kAdamThreads;          // one element per thread (capped: then a strided loop)
    hipLaunchKernelGGL(adam_clip_step_kernel, dim3(upd_blocks < 1024 ? upd_blocks : 1024), dim3(kAdamThreads), 0, (hipStream_t)stream, k,
                       partial, (const double*)snap, state);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_adam_state_bytes(void) { return (int)(4 * sizeof(double) + kAdamBlocks * sizeof(float)); }


extern "C" int ag_ppo_loss_finalize(const float* partials, int num_blocks, int M, int A, const float* logstd,
                                    float entropy_coef, float critic_coef, float bounds_loss_coef, float* grad_logstd,
                                    float* grad_head_bias, float* kl_out, float* stats, void* stream) {
    if (!partials || !logstd || !grad_logstd || !grad_head_bias || !kl_out || !stats || num_blocks <= 0 || M <= 0)
        return AG_ERR_INVALID_ARG;
    if (A < 1 || A > AG_MAX_ACTIONS) return AG_ERR_UNSUPPORTED;
    FinalizeArgs k{partials, num_blocks, M, A, logstd, entropy_coef, critic_coef, bounds_loss_coef,
                   grad_logstd, grad_head_bias, kl_out, stats};
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_normalize_rows(const float* x, const double* mean, const double* var, float* out, long long rows, int D,
                                 float eps, float clip, void* stream) {
    if (!x || !mean || !var || !out || rows <= 0 || D <= 0) return AG_ERR_INVALID_ARG;
    if (D > 4096) return AG_ERR_UNSUPPORTED;
    const size_t total = (size_t)rows * (size_t)D;
    size_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)grid), dim3(256), 2 * D * sizeof(float), (hipStream_t)stream,
                       x, mean, var, out, total, D, eps, clip);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------
// Fused edges of the MLP trunk (lib/network/mlp.py:36-39; lib/model/a2c_continuous_logstd_model.py:126-146).
// The 256x256 GEMMs stay with hipBLASLt; these kernels remove the HBM round trips AROUND them:
//   input_layer : xn = clamp((obs-mean)/std), h1 = ELU(xn W1^T + b1)        one pass: read obs, write xn + h1
//   elu_heads   : h = ELU(z) in place, heads = h Wh^T + bh                  the [M,C]x[C,A+1] GEMM rides along
//   (the backward counterpart, heads_bwd_elu_wgrad, is further down with the folded weight gradients)
// All are HBM-bound streaming kernels: C/4 threads cover one row with float4 accesses.
// ---------------------------------------------------------------------------------------------------
namespace {

// ELU(alpha = 1) as torch evaluates it, exp(z) - 1 on the negative side, with the hardware exp2 (v_exp_f32, ~1 ulp):
// expm1f from the device library costs ~10x more VALU issue slots and made these streaming kernels compute-bound.
__device__ __forceinline__ float elu1(float z) {
    return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f;
}

constexpr int kInTileRows = 64;

// grid: ceil(M / 64); block 256; dynamic LDS: Wt[D][C] + xs[64][D]
__global__ __launch_bounds__(256) void input_layer_kernel(const float* __restrict__ obs, const double* __restrict__ mean,
                                                          const double* __restrict__ var, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ xn,
                                                          float* __restrict__ h, int M, int D, int C, float eps, float clip,
                                                          int normalize) {
    extern __shared__ float lds[];
    float* Wt = lds;                    // [D][C]
    float* xs = lds + (size_t)D * C;    // [64][D]
    for (int i = threadIdx.x; i < D * C; i += 256) {
        const int c = i / D, d = i - c * D;          // W is [C][D] row-major
        Wt[d * C + c] = W[i];
    }
    const int row0 = blockIdx.x * kInTileRows;
    const int rows = min(kInTileRows, M - row0);
    for (int i = threadIdx.x; i < rows * D; i += 256) {
        const int d = i % D;
        float v = obs[(size_t)row0 * D + i];
        if (normalize) {
            v = (v - (float)mean[d]) / sqrtf((float)var[d] + eps);
            v = fminf(fmaxf(v, -clip), clip);
            xn[(size_t)row0 * D + i] = v;
        }
        xs[i] = v;
    }
    __syncthreads();
    const int tpr = C >> 2;
    const int rgroups = 256 / tpr;                   // row groups working concurrently
    const int col4 = threadIdx.x % tpr, rg = threadIdx.x / tpr;
    if (rg >= rgroups) return;
    const float4 b4 = reinterpret_cast<const float4*>(bias)[col4];
    for (int r = rg * 4; r < rows; r += rgroups * 4) {
        float4 acc[4] = {b4, b4, b4, b4};
        const int nr = min(4, rows - r);
        for (int d = 0; d < D; ++d) {
            const float4 w = reinterpret_cast<const float4*>(Wt + (size_t)d * C)[col4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = xs[min(r + j, rows - 1) * D + d];
                acc[j].x = fmaf(x, w.x, acc[j].x); acc[j].y = fmaf(x, w.y, acc[j].y);
                acc[j].z = fmaf(x, w.z, acc[j].z); acc[j].w = fmaf(x, w.w, acc[j].w);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nr) {
                float4 o;
                o.x = elu1(acc[j].x); o.y = elu1(acc[j].y); o.z = elu1(acc[j].z); o.w = elu1(acc[j].w);
                reinterpret_cast<float4*>(h + (size_t)(row0 + r + j) * C)[col4] = o;
            }
        }
    }
}

// Same contract as input_layer_kernel with D known at compile time: each thread keeps its 4 x D slab of W in registers
// (its column group never changes) and streams rows: per row D broadcast LDS reads, 4*D FMAs, one float4 store.
constexpr int kInRegTileRows = 128;

template <int CPT> struct VecN;
template <> struct VecN<4> { typedef float4 type; };
template <> struct VecN<2> { typedef float2 type; };

// CPT = output columns per thread (4: float4 stores, D <= 20; 2: float2 stores, wider inputs such as Tracking's D = 48,
// so that the D x CPT slab of W still fits in registers).
template <int D, int CPT>
__global__ __launch_bounds__(256) void input_layer_reg_kernel(const float* __restrict__ obs, const double* __restrict__ mean,
                                                              const double* __restrict__ var, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ xn,
                                                              float* __restrict__ h, int M, int C, float eps, float clip,
                                                              int normalize) {
    typedef typename VecN<CPT>::type vec_t;
    __shared__ float xs[kInRegTileRows * D];
    const int row0 = blockIdx.x * kInRegTileRows;
    const int rows = min(kInRegTileRows, M - row0);
    for (int i = threadIdx.x; i < rows * D; i += 256) {
        const int d = i % D;
        float v = obs[(size_t)row0 * D + i];
        if (normalize) {
            v = (v - (float)mean[d]) / sqrtf((float)var[d] + eps);
            v = fminf(fmaxf(v, -clip), clip);
            xn[(size_t)row0 * D + i] = v;
        }
        xs[i] = v;
    }
    const int tpr = C / CPT;
    const int rgroups = 256 / tpr;
    const int colg = threadIdx.x % tpr, rg = threadIdx.x / tpr;
    // column PAIRS in two-wide vectors: the D x CPT FMAs of a row become D x CPT / 2 v_pk_fma_f32 (same IEEE fma per element, same
    // order - bit-identical results); with scalar FMAs the kernel sat at ~55 % of its vector-ALU ceiling while writing h
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int CP = CPT / 2;
    f32x2_t w[D][CP];
    f32x2_t b[CP];
    if (rg < rgroups) {
#pragma unroll
        for (int j = 0; j < CP; ++j) {
            b[j] = f32x2_t{bias[colg * CPT + 2 * j], bias[colg * CPT + 2 * j + 1]};
#pragma unroll
            for (int d = 0; d < D; ++d)
                w[d][j] = f32x2_t{W[(size_t)(colg * CPT + 2 * j) * D + d], W[(size_t)(colg * CPT + 2 * j + 1) * D + d]};
        }
    }
    __syncthreads();
    if (rg >= rgroups) return;
    for (int r = rg; r < rows; r += rgroups) {
        const float* xr = xs + r * D;
        f32x2_t acc[CP];
#pragma unroll
        for (int j = 0; j < CP; ++j) acc[j] = b[j];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float x = xr[d];
#pragma unroll
            for (int j = 0; j < CP; ++j) acc[j] = __builtin_elementwise_fma(f32x2_t{x, x}, w[d][j], acc[j]);
        }
        float o[CPT];
#pragma unroll
        for (int j = 0; j < CP; ++j) {
            o[2 * j] = elu1(acc[j].x);
            o[2 * j + 1] = elu1(acc[j].y);
        }
        reinterpret_cast<vec_t*>(h + (size_t)(row0 + r) * C)[colg] = *reinterpret_cast<const vec_t*>(o);
    }
}

// Sum over groups of `width` consecutive lanes (width = 16, 32 or 64) with DPP row operations - pure VALU, no LDS round
// trips (six dependent ds_bpermute per value made elu_heads latency-bound).  The group total is valid in the LAST lane of
// each group (lane % width == width - 1).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

__device__ __forceinline__ float group_sum_last_lane(float v, int width) {
    v = dpp_add<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);       // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);       // row_mirror: every lane of a 16-lane row holds the row sum
    if (width >= 32) v = dpp_add<0x142, 0xA>(v);   // row_bcast15: rows 1,3 += lane 15 of rows 0,2
    if (width >= 64) v = dpp_add<0x143, 0xC>(v);   // row_bcast31: rows 2,3 += lane 31
    return v;
}

// h = ELU(z) in place; heads[m, a] = sum_c h[m,c] Wh[a,c] + bh[a].  tpr = C/4 threads per row (power of two <= 64).
template <int A1, bool WRITE_BACK>
__global__ __launch_bounds__(256) void elu_heads_kernel(float* __restrict__ zh, const float* __restrict__ Wh,
                                                        const float* __restrict__ bh, float* __restrict__ heads, int M, int C,
                                                        int rows_per_block, const float* __restrict__ zbias) {
    const int tpr = C >> 2;
    const int rpp = 256 / tpr;
    const int col4 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    // zbias: the producing Linear's bias when the GEMM ran without its bias epilogue (zh then holds x W^T only)
    const float4 zb = zbias ? reinterpret_cast<const float4*>(zbias)[col4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w[A1];
#pragma unroll
    for (int a = 0; a < A1; ++a) w[a] = reinterpret_cast<const float4*>(Wh + (size_t)a * C)[col4];
    const int row0 = blockIdx.x * rows_per_block;
    const int row_end = min(row0 + rows_per_block, M);
    for (int r = row0 + rsub; r < row_end; r += 2 * rpp) {
        // two rows in flight per thread: both loads are issued before either row is finished
        const int r2 = r + rpp;
        const bool has2 = r2 < row_end;
        float4* p0 = reinterpret_cast<float4*>(zh + (size_t)r * C) + col4;
        float4* p1 = reinterpret_cast<float4*>(zh + (size_t)(has2 ? r2 : r) * C) + col4;
        float4 z0 = *p0;
        float4 z1 = *p1;
        z0.x = elu1(z0.x + zb.x); z0.y = elu1(z0.y + zb.y); z0.z = elu1(z0.z + zb.z); z0.w = elu1(z0.w + zb.w);
        z1.x = elu1(z1.x + zb.x); z1.y = elu1(z1.y + zb.y); z1.z = elu1(z1.z + zb.z); z1.w = elu1(z1.w + zb.w);
        if (WRITE_BACK) {      // without it the buffer keeps the pre-activation z (HBM writes cost ~2x reads on this part)
            *p0 = z0;
            if (has2) *p1 = z1;
        }
        float s0[A1], s1[A1];
#pragma unroll
        for (int a = 0; a < A1; ++a) {
            s0[a] = (z0.x * w[a].x + z0.y * w[a].y) + (z0.z * w[a].z + z0.w * w[a].w);
            s1[a] = (z1.x * w[a].x + z1.y * w[a].y) + (z1.z * w[a].z + z1.w * w[a].w);
        }
#pragma unroll
        for (int a = 0; a < A1; ++a) {
            s0[a] = group_sum_last_lane(s0[a], tpr);
            s1[a] = group_sum_last_lane(s1[a], tpr);
        }
        if (col4 == tpr - 1) {
#pragma unroll
            for (int a = 0; a < A1; ++a) {
                heads[(size_t)r * A1 + a] = s0[a] + bh[a];
                if (has2) heads[(size_t)r2 * A1 + a] = s1[a] + bh[a];
            }
        }
    }
}

bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

}  // namespace

extern "C" int ag_mlp_input_layer(const float* obs, const double* mean, const double* var, const float* W, const float* bias,
                                  float* xn, float* h, int M, int D, int C, float eps, float clip, void* stream) {
    if (!obs || !W || !bias || !h || M <= 0 || D <= 0) return AG_ERR_INVALID_ARG;
    const int normalize = (mean && var && xn) ? 1 : 0;
    if (!normalize && (mean || var || xn)) return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > 1024 || (C & 3) || (256 % (C >> 2)) != 0) return AG_ERR_UNSUPPORTED;
#define AG_IN(DV, CPTV)                                                                                                    \
    hipLaunchKernelGGL((input_layer_reg_kernel<DV, CPTV>), dim3((M + kInRegTileRows - 1) / kInRegTileRows), dim3(256), 0,     \
                       (hipStream_t)stream, obs, mean, var, W, bias, xn, h, M, C, eps, clip, normalize)
    if (D == 16 || D == 18 || D == 20) {
        if (D == 16) AG_IN(16, 4); else if (D == 18) AG_IN(18, 4); else AG_IN(20, 4);
        return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
    }
    if (D == 48 && C <= 512 && (C & 1) == 0 && 256 % (C / 2) == 0) {      // Tracking: 48 observations
        AG_IN(48, 2);
        return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
    }
#undef AG_IN
    const size_t lds = ((size_t)D * C + (size_t)kInTileRows * D) * sizeof(float);
    if (lds > 64 * 1024) return AG_ERR_UNSUPPORTED;
    const int grid = (M + kInTileRows - 1) / kInTileRows;
    hipLaunchKernelGGL(input_layer_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, obs, mean, var, W, bias, xn, h,
                       M, D, C, eps, clip, normalize);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_elu_heads(float* zh, const float* Wh, const float* bh, float* heads, int M, int C, int A1, int write_back,
                            const float* zbias, void* stream) {
    if (!zh || !Wh || !bh || !heads || M <= 0) return AG_ERR_INVALID_ARG;
    if (C < 64 || C > 256 || !pow2(C)) return AG_ERR_UNSUPPORTED;      // C/4 lanes per row: one, two or four DPP rows
    const int rows_per_block = 64;
    const int grid = (M + rows_per_block - 1) / rows_per_block;
#define AG_EH(A1V, WB) hipLaunchKernelGGL((elu_heads_kernel<A1V, WB>), dim3(grid), dim3(256), 0, (hipStream_t)stream, zh, Wh, bh, \
                                         heads, M, C, rows_per_block, zbias)
    if (A1 == 5) { if (write_back) AG_EH(5, true); else AG_EH(5, false); }
    else if (A1 == 6) { if (write_back) AG_EH(6, true); else AG_EH(6, false); }
    else return AG_ERR_UNSUPPORTED;
#undef AG_EH
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------
// Backward edges with the small weight gradients folded in (no [M,C] intermediate is written just to be re-read):
//   heads_bwd_elu_wgrad : as heads_bwd_elu, plus per-block partials of dWh[a,c] = sum_m d_heads[m,a] h[m,c]
//   elu_bwd_input_wgrad : FIRST layer.  dz = dh * ELU'(h) is consumed on the fly: per-block partials of
//                         dW[c,d] = sum_m dz[m,c] x[m,d]  (D <= 24) and db[c] = sum_m dz[m,c]; dz itself is never stored
//                         (nothing upstream of the first layer needs it).
// Partials are [blocks, ...]; the caller reduces them with one sum over dim 0 (deterministic).
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kWgRows = 128;      // rows per block, head kernel (measured: 82 us at 128, 109 us at 256)
constexpr int kInWgRows = 256;    // rows per block, first-layer kernel (measured: 106 us at 128, 95 us at 256)

template <int A1, bool PREACT>
__global__ __launch_bounds__(256) void heads_bwd_elu_wgrad_kernel(const float* __restrict__ d_heads, const float* __restrict__ Wh,
                                                                  const float* __restrict__ h, float* __restrict__ dz,
                                                                  float* __restrict__ db_partials, float* __restrict__ dwh_partials,
                                                                  int M, int C, const float* __restrict__ zbias) {
    __shared__ float4 red[256];
    const int tpr = C >> 2;
    const int rpp = 256 / tpr;
    const int col4 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const float4 zb = (PREACT && zbias) ? reinterpret_cast<const float4*>(zbias)[col4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w[A1], gw[A1];
#pragma unroll
    for (int a = 0; a < A1; ++a) {
        w[a] = reinterpret_cast<const float4*>(Wh + (size_t)a * C)[col4];
        gw[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row0 = blockIdx.x * kWgRows;
    const int row_end = min(row0 + kWgRows, M);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < rpp) {
        for (int r = row0 + rsub; r < row_end; r += rpp) {
            const size_t idx = (size_t)r * tpr + col4;
            float4 y = reinterpret_cast<const float4*>(h)[idx];
            if (PREACT) {        // the buffer holds z, not ELU(z): rebuild h here (ELU'(z) = h + 1 on the negative side)
                y.x = elu1(y.x + zb.x); y.y = elu1(y.y + zb.y); y.z = elu1(y.z + zb.z); y.w = elu1(y.w + zb.w);
            }
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < A1; ++a) {
                const float d = d_heads[(size_t)r * A1 + a];
                g.x = fmaf(d, w[a].x, g.x); g.y = fmaf(d, w[a].y, g.y); g.z = fmaf(d, w[a].z, g.z); g.w = fmaf(d, w[a].w, g.w);
                gw[a].x = fmaf(d, y.x, gw[a].x); gw[a].y = fmaf(d, y.y, gw[a].y);
                gw[a].z = fmaf(d, y.z, gw[a].z); gw[a].w = fmaf(d, y.w, gw[a].w);
            }
            float4 o;
            o.x = g.x * (y.x > 0.f ? 1.f : y.x + 1.f);
            o.y = g.y * (y.y > 0.f ? 1.f : y.y + 1.f);
            o.z = g.z * (y.z > 0.f ? 1.f : y.z + 1.f);
            o.w = g.w * (y.w > 0.f ? 1.f : y.w + 1.f);
            reinterpret_cast<float4*>(dz)[idx] = o;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
    }
    // reduce the rpp row groups: bias sums, then each head row
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < tpr) {
        float4 s = red[threadIdx.x];
        for (int j = 1; j < rpp; ++j) {
            const float4 t = red[threadIdx.x + j * tpr];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        reinterpret_cast<float4*>(db_partials)[(size_t)blockIdx.x * tpr + threadIdx.x] = s;
    }
#pragma unroll
    for (int a = 0; a < A1; ++a) {
        __syncthreads();
        red[threadIdx.x] = gw[a];
        __syncthreads();
        if (threadIdx.x < tpr) {
            float4 s = red[threadIdx.x];
            for (int j = 1; j < rpp; ++j) {
                const float4 t = red[threadIdx.x + j * tpr];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            reinterpret_cast<float4*>(dwh_partials)[((size_t)blockIdx.x * A1 + a) * tpr + threadIdx.x] = s;
        }
    }
}

// D is a template parameter so the [CPT][D] accumulator tile lives in registers (CPT columns per thread).  Two rows are in flight per thread (loads issued before use); the row groups then add their
// tiles into one [C][D+1] LDS buffer in turn.
template <int D, int CPT, int ROWS>
__global__ __launch_bounds__(256) void elu_bwd_input_wgrad_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                                  const float* __restrict__ x, float* __restrict__ dw_partials,
                                                                  float* __restrict__ db_partials, int M, int C) {
    typedef typename VecN<CPT>::type vec_t;
    extern __shared__ float lds[];          // xs[ROWS][D] | red[C][D+1]
    float* xs = lds;
    float* red = lds + ROWS * D;
    const int tpr = C / CPT;
    const int rpp = 256 / tpr;
    const int colg = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int row0 = blockIdx.x * ROWS;
    const int rows = min(ROWS, M - row0);
    for (int i = threadIdx.x; i < rows * D; i += 256) xs[i] = x[(size_t)row0 * D + i];
    __syncthreads();
    float acc[CPT][D];
    float bsum[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        bsum[j] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[j][d] = 0.f;
    }
    if (rsub < rpp) {
        for (int r = rsub; r < rows; r += 2 * rpp) {
            const int r2 = r + rpp;
            const bool has2 = r2 < rows;
            const size_t i0 = (size_t)(row0 + r) * tpr + colg;
            const size_t i1 = (size_t)(row0 + (has2 ? r2 : r)) * tpr + colg;
            const vec_t g0v = reinterpret_cast<const vec_t*>(dh)[i0];
            const vec_t y0v = reinterpret_cast<const vec_t*>(h)[i0];
            const vec_t g1v = reinterpret_cast<const vec_t*>(dh)[i1];
            const vec_t y1v = reinterpret_cast<const vec_t*>(h)[i1];
            const float* g0 = reinterpret_cast<const float*>(&g0v);
            const float* y0 = reinterpret_cast<const float*>(&y0v);
            const float* g1 = reinterpret_cast<const float*>(&g1v);
            const float* y1 = reinterpret_cast<const float*>(&y1v);
            const float m2 = has2 ? 1.f : 0.f;
            float o[CPT], q[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                o[j] = g0[j] * (y0[j] > 0.f ? 1.f : y0[j] + 1.f);
                q[j] = m2 * g1[j] * (y1[j] > 0.f ? 1.f : y1[j] + 1.f);
            }
            const float* xr0 = xs + r * D;
            const float* xr1 = xs + (has2 ? r2 : r) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xv0 = xr0[d], xv1 = xr1[d];
#pragma unroll
                for (int j = 0; j < CPT; ++j) acc[j][d] = fmaf(q[j], xv1, fmaf(o[j], xv0, acc[j][d]));
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) bsum[j] += o[j] + q[j];
        }
    }
    // the rpp row groups add their tiles into red[C][D+1] one after the other (fixed order -> deterministic)
    for (int g2 = 0; g2 < rpp; ++g2) {
        if (rsub == g2) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                float* dst = red + (size_t)(colg * CPT + j) * (D + 1);
                if (g2 == 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) dst[d] = acc[j][d];
                    dst[D] = bsum[j];
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) dst[d] += acc[j][d];
                    dst[D] += bsum[j];
                }
            }
        }
        __syncthreads();
    }
    const int per = C * (D + 1);
    for (int i = threadIdx.x; i < per; i += 256) {
        const int c = i / (D + 1), d = i - c * (D + 1);
        if (d < D) dw_partials[((size_t)blockIdx.x * C + c) * D + d] = red[i];
        else db_partials[(size_t)blockIdx.x * C + c] = red[i];
    }
}

}  // namespace

extern "C" int ag_wgrad_rows_per_block(int which) { return which == 0 ? kWgRows : kInWgRows; }

// D = 48 (Tracking) was tried with 2 columns per thread: 194 us vs 168 us for ag_elu_bwd_bias + the split-K bmm, so wide
// inputs keep the unfused path.
extern "C" int ag_input_wgrad_rows(int D) { return (D == 16 || D == 18 || D == 20) ? kInWgRows : 0; }

extern "C" int ag_heads_bwd_elu_wgrad(const float* d_heads, const float* Wh, const float* h, float* dz, float* db_partials,
                                      float* dwh_partials, int M, int C, int A1, int h_is_preactivation, const float* zbias,
                                      void* stream) {
    if (!d_heads || !Wh || !h || !dz || !db_partials || !dwh_partials || M <= 0) return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > 1024 || (C & 3) || (256 % (C >> 2)) != 0) return AG_ERR_UNSUPPORTED;
    const int grid = (M + kWgRows - 1) / kWgRows;
#define AG_HB(A1V, PA) hipLaunchKernelGGL((heads_bwd_elu_wgrad_kernel<A1V, PA>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                         d_heads, Wh, h, dz, db_partials, dwh_partials, M, C, zbias)
    if (A1 == 5) { if (h_is_preactivation) AG_HB(5, true); else AG_HB(5, false); }
    else if (A1 == 6) { if (h_is_preactivation) AG_HB(6, true); else AG_HB(6, false); }
    else return AG_ERR_UNSUPPORTED;
#undef AG_HB
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_elu_bwd_input_wgrad(const float* dh, const float* h, const float* x, float* dw_partials, float* db_partials,
                                      int M, int C, int D, void* stream) {
    if (!dh || !h || !x || !dw_partials || !db_partials || M <= 0) return AG_ERR_INVALID_ARG;
    const int cpt = 4;
    const int rows = ag_input_wgrad_rows(D);
    if (rows == 0 || C <= 0 || C > 256 * cpt || (C % cpt) != 0 || (256 % (C / cpt)) != 0) return AG_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)rows * D + (size_t)C * (D + 1));
    if (lds > 160 * 1024) return AG_ERR_UNSUPPORTED;
    const int grid = (M + rows - 1) / rows;
#define AG_LAUNCH_D(DV, CPTV, ROWSV)                                                                                          \
    case DV: {                                                                                                                \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)elu_bwd_input_wgrad_kernel<DV, CPTV, ROWSV>,                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)      \
            return AG_ERR_HIP;                                                                                                \
        hipLaunchKernelGGL((elu_bwd_input_wgrad_kernel<DV, CPTV, ROWSV>), dim3(grid), dim3(256), lds, (hipStream_t)stream,    \
                           dh, h, x, dw_partials, db_partials, M, C);                                                         \
        break;                                                                                                                \
    }
    switch (D) {
        AG_LAUNCH_D(16, 4, kInWgRows)
        AG_LAUNCH_D(18, 4, kInWgRows)
        AG_LAUNCH_D(20, 4, kInWgRows)
        default: return AG_ERR_UNSUPPORTED;
    }
#undef AG_LAUNCH_D
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------
// All partial-sum reductions of one minibatch in TWO launches (instead of a fill + reduce pair per gradient):
// stage 1 sums each job's S partial rows in up to 64 groups (a flat grid over all jobs' column-blocks x groups), stage 2 sums the groups
// into the gradient slice.  Fixed summation order -> deterministic.  n % 4 == 0 and 16-byte aligned partials; the
// destination only needs 4-byte alignment.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kSumMaxGroups = 64;        // partial rows are first summed in up to 64 groups per job (>= 8 rows per group)
constexpr int kMaxSumJobs = AG_MAX_SUM_JOBS;

struct SumJobs {
    const float* in[kMaxSumJobs];
    float* out[kMaxSumJobs];
    int S[kMaxSumJobs];
    int n4[kMaxSumJobs];
    int groups[kMaxSumJobs];
    int block0_s1[kMaxSumJobs + 1];       // first flat block of each job in stage 1 (blocks = ceil(n4/64) * groups)
    int block0_s2[kMaxSumJobs + 1];       // ... and in stage 2 (blocks = ceil(n4/64))
    long long scratch_off[kMaxSumJobs];   // in floats, into scratch [sum over jobs of groups * n]
    float* scratch;
    int njobs;
};

__device__ __forceinline__ int find_job(const int* block0, int njobs, int b) {
    int j = 0;
    while (j + 1 < njobs && b >= block0[j + 1]) ++j;
    return j;
}

// block = 64 float4 columns x 4 row lanes.  FIN: one more workgroup behind the last job's blocks runs ppo_loss_finalize's body (its
// inputs - the loss partials of the forward launch - are long complete, its outputs are other slots of the flat gradient than the
// jobs'): a launch less per optimizer step
template <bool FIN>
__global__ __launch_bounds__(256) void sum_rows_stage1_kernel(const SumJobs k, const FinalizeArgs fin) {
    if constexpr (FIN) {
        if ((int)blockIdx.x == k.block0_s1[k.njobs]) {
            ppo_loss_finalize_block(fin);
            return;
        }
    }
    __shared__ float4 red[256];
    const int j = find_job(k.block0_s1, k.njobs, blockIdx.x);
    const int local = blockIdx.x - k.block0_s1[j];
    const int n4 = k.n4[j], G = k.groups[j], S = k.S[j];
    const int g = local % G, bx = local / G;
    const int col = bx * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x >> 6;
    const int rpg = (S + G - 1) / G;
    const int s_end = min((g + 1) * rpg, S);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4) {
        const float4* src = reinterpret_cast<const float4*>(k.in[j]);
#pragma unroll 4
        for (int s = g * rpg + lane; s < s_end; s += 4) {
            const float4 v = src[(size_t)s * n4 + col];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < n4) {
        float4 s0 = red[threadIdx.x];
        const float4 s1 = red[threadIdx.x + 64], s2 = red[threadIdx.x + 128], s3 = red[threadIdx.x + 192];
        s0.x = (s0.x + s1.x) + (s2.x + s3.x); s0.y = (s0.y + s1.y) + (s2.y + s3.y);
        s0.z = (s0.z + s1.z) + (s2.z + s3.z); s0.w = (s0.w + s1.w) + (s2.w + s3.w);
        reinterpret_cast<float4*>(k.scratch + k.scratch_off[j])[(size_t)g * n4 + col] = s0;
    }
}

__global__ __launch_bounds__(256) void sum_rows_stage2_kernel(const SumJobs k) {
    __shared__ float4 red[256];
    const int j = find_job(k.block0_s2, k.njobs, blockIdx.x);
    const int bx = blockIdx.x - k.block0_s2[j];
    const int n4 = k.n4[j], G = k.groups[j];
    const int col = bx * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4) {
        const float4* src = reinterpret_cast<const float4*>(k.scratch + k.scratch_off[j]);
#pragma unroll 4
        for (int g = lane; g < G; g += 4) {
            const float4 v = src[(size_t)g * n4 + col];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < n4) {
        const float4 s0 = red[threadIdx.x], s1 = red[threadIdx.x + 64], s2 = red[threadIdx.x + 128], s3 = red[threadIdx.x + 192];
        float* dst = k.out[j] + (size_t)col * 4;
        dst[0] = (s0.x + s1.x) + (s2.x + s3.x); dst[1] = (s0.y + s1.y) + (s2.y + s3.y);
        dst[2] = (s0.z + s1.z) + (s2.z + s3.z); dst[3] = (s0.w + s1.w) + (s2.w + s3.w);
    }
}

}  // namespace

extern "C" int ag_sum_rows_groups(void) { return kSumMaxGroups; }

static int sum_rows_multi_launch(const ag_sum_job* jobs, int njobs, float* scratch, long long scratch_floats, const FinalizeArgs* fin,
                                 void* stream) {
    if (!jobs || !scratch || njobs <= 0) return AG_ERR_INVALID_ARG;
    if (njobs > kMaxSumJobs) return AG_ERR_UNSUPPORTED;
    SumJobs k{};
    k.scratch = scratch;
    k.njobs = njobs;
    long long off = 0;
    int b1 = 0, b2 = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!jobs[j].partials_dev || !jobs[j].out_dev || jobs[j].rows <= 0 || jobs[j].n <= 0) return AG_ERR_INVALID_ARG;
        if (jobs[j].n % 4 != 0 || (reinterpret_cast<uintptr_t>(jobs[j].partials_dev) & 15)) return AG_ERR_UNSUPPORTED;
        k.in[j] = jobs[j].partials_dev;
        k.out[j] = jobs[j].out_dev;
        k.S[j] = jobs[j].rows;
        k.n4[j] = jobs[j].n / 4;
        int G = jobs[j].rows / 8;
        G = G < 1 ? 1 : (G > kSumMaxGroups ? kSumMaxGroups : G);
        k.groups[j] = G;
        k.scratch_off[j] = off;
        off += (long long)G * jobs[j].n;
        const int nbx = (k.n4[j] + 63) / 64;
        k.block0_s1[j] = b1;
        k.block0_s2[j] = b2;
        b1 += nbx * G;
        b2 += nbx;
    }
    k.block0_s1[njobs] = b1;
    k.block0_s2[njobs] = b2;
    if (off > scratch_floats || (reinterpret_cast<uintptr_t>(scratch) & 15)) return AG_ERR_INVALID_ARG;
    if (fin)
        hipLaunchKernelGGL(sum_rows_stage1_kernel<true>, dim3(b1 + 1), dim3(256), 0, (hipStream_t)stream, k, *fin);
    else
        hipLaunchKernelGGL(sum_rows_stage1_kernel<false>, dim3(b1), dim3(256), 0, (hipStream_t)stream, k, FinalizeArgs{});
    hipLaunchKernelGGL(sum_rows_stage2_kernel, dim3(b2), dim3(256), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_sum_rows_multi(const ag_sum_job* jobs, int njobs, float* scratch, long long scratch_floats, void* stream) {
    return sum_rows_multi_launch(jobs, njobs, scratch, scratch_floats, nullptr, stream);
}

extern "C" int ag_sum_rows_multi_finalize(const ag_sum_job* jobs, int njobs, float* scratch, long long scratch_floats,
                                          const float* loss_partials, int num_blocks, int M, int A, const float* logstd,
                                          float entropy_coef, float critic_coef, float bounds_loss_coef, float* grad_logstd,
                                          float* grad_head_bias, float* kl_out, float* stats, void* stream) {
    if (!loss_partials || !logstd || !grad_logstd || !grad_head_bias || !kl_out || !stats || num_blocks <= 0 || M <= 0)
        return AG_ERR_INVALID_ARG;
    if (A < 1 || A > AG_MAX_ACTIONS) return AG_ERR_UNSUPPORTED;
    const FinalizeArgs fin{loss_partials, num_blocks, M, A, logstd, entropy_coef, critic_coef, bounds_loss_coef,
                           grad_logstd, grad_head_bias, kl_out, stats};
    return sum_rows_multi_launch(jobs, njobs, scratch, scratch_floats, &fin, stream);
}

// ---------------------------------------------------------------------------------------------------
// RunningMeanStd.update (lib/core/running_mean_std.py:31-62) for a [rows, D] batch in two launches: per-block column
// sums / sums of squares in float64, then one workgroup forms the batch mean and unbiased variance and merges them into
// the running statistics with the reference's parallel-variance formula.  Replaces ~20 eager kernels per call.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kRmsBlocks = 256;

__global__ __launch_bounds__(256) void rms_moments_kernel(const float* __restrict__ x, long long rows, int D,
                                                          double* __restrict__ partial) {   // [kRmsBlocks][2][D]
    extern __shared__ double sh[];      // [groups][2][D]
    const int groups = 256 / D;         // row groups per block (D <= 256)
    const int c = threadIdx.x % D, g = threadIdx.x / D;
    double s = 0.0, q = 0.0;
    if (g < groups) {
        // four rows in flight per thread (one dependent load per trip made this launch latency-bound: 23 us for 14 MB)
        const long long stride = (long long)kRmsBlocks * groups;
        long long r = (long long)blockIdx.x * groups + g;
        double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
        for (; r + 3 * stride < rows; r += 4 * stride) {
            const float a0 = x[r * D + c], a1 = x[(r + stride) * D + c], a2 = x[(r + 2 * stride) * D + c], a3 = x[(r + 3 * stride) * D + c];
            const double v0 = (double)a0, v1 = (double)a1, v2 = (double)a2, v3 = (double)a3;
            s += v0; q += v0 * v0;
            s1 += v1; q1 += v1 * v1;
            s2 += v2; q2 += v2 * v2;
            s3 += v3; q3 += v3 * v3;
        }
        for (; r < rows; r += stride) {
            const double v = (double)x[r * D + c];
            s += v;
            q += v * v;
        }
        s = (s + s1) + (s2 + s3);
        q = (q + q1) + (q2 + q3);
        sh[(g * 2 + 0) * D + c] = s;
        sh[(g * 2 + 1) * D + c] = q;
    }
    __syncthreads();
    if (threadIdx.x < D) {
        double ts = 0.0, tq = 0.0;
        for (int k = 0; k < groups; ++k) {
            ts += sh[(k * 2 + 0) * D + threadIdx.x];
            tq += sh[(k * 2 + 1) * D + threadIdx.x];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * D + threadIdx.x] = ts;
        partial[((size_t)blockIdx.x * 2 + 1) * D + threadIdx.x] = tq;
    }
}

__global__ __launch_bounds__(256) void rms_merge_kernel(const double* __restrict__ partial, long long rows, int D,
                                                        double* __restrict__ mean, double* __restrict__ var,
                                                        double* __restrict__ count) {
    __shared__ double shs[256], shq[256];
    const double cnt = *count;
    // the blocks' partials: 256 / D lanes per column (D <= 256), then a fixed-order sum over the lanes (D serial chains of
    // 2 x kRmsBlocks dependent loads made this launch 20 us)
    const int lanes = 256 / D;
    {
        const int c = (int)threadIdx.x % D, l = (int)threadIdx.x / D;
        double ps = 0.0, pq = 0.0;
        if (l < lanes) {
            for (int b = l; b < kRmsBlocks; b += lanes) {
                ps += partial[((size_t)b * 2 + 0) * D + c];
                pq += partial[((size_t)b * 2 + 1) * D + c];
            }
        }
        shs[threadIdx.x] = ps;
        shq[threadIdx.x] = pq;
    }
    __syncthreads();                     // (also: everyone has read the old count before thread 0 overwrites it)
    if ((int)threadIdx.x < D) {
        const int c = threadIdx.x;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < lanes; ++k) {
            s += shs[k * D + c];
            q += shq[k * D + c];
        }
        const double n = (double)rows;
        const double bmean = s / n;
        const double bvar = (q - s * bmean) / (n - 1.0);        // unbiased, as torch.var
        const double delta = bmean - mean[c];
        const double tot = cnt + n;
        const double m2 = var[c] * cnt + bvar * n + delta * delta * cnt * n / tot;
        mean[c] = mean[c] + delta * n / tot;
        var[c] = m2 / tot;
    }
    if (threadIdx.x == 0) *count = cnt + (double)rows;
}

}  // namespace

extern "C" long long ag_rms_scratch_doubles(int D) { return (long long)kRmsBlocks * 2 * D; }

extern "C" int ag_rms_update(const float* x, long long rows, int D, double* mean, double* var, double* count, double* scratch,
                             void* stream) {
    if (!x || !mean || !var || !count || !scratch || rows < 2) return AG_ERR_INVALID_ARG;
    if (D <= 0 || D > 256) return AG_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(256 / D) * 2 * D * sizeof(double);
    hipLaunchKernelGGL(rms_moments_kernel, dim3(kRmsBlocks), dim3(256), lds, (hipStream_t)stream, x, rows, D, scratch);
    hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, rows, D, mean, var, count);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
