// rollout_kernels.hip - the per-step bookkeeping of A2CBase.play_steps (lib/agent/a2c_base.py:651-695) and GAE
// (a2c_base.py:463-478) as three streaming kernels, so that one rollout step is
//     ag_mlp_input_layer -> GEMM -> ag_elu_heads -> ag_policy_sample -> ag_step_into -> ag_rollout_account
// instead of ~60 eager launches.
//   policy_sample   : a = mu + sigma * N(0,1) (Philox4x32-10, counter = (env, rollout*H + slot, stream 16, block)),
//                     neglogp, value de-normalisation, mu/sigma copies, clamped env action
//                     (a2c_continuous_logstd_model.py:159-167, base_model.py:29-35, a2c_base.py:229-236)
//   rollout_account : reward shaping (tr_helpers.py:16-42), optional time-out bootstrap (a2c_base.py:672-673),
//                     running episode reward/length and the sums over the episodes that ended (a2c_base.py:678-695)
//   gae             : backward recursion per env over the horizon (a2c_base.py:463-478), returns = adv + value
#include <hip/hip_runtime.h>

#include "../../include/airgym_hip.h"
#include "rollout_math.hpp"

namespace {

struct SampleArgs {
    const float* heads; const float* logstd; const double* vmean; const double* vvar; float veps;
    uint32_t key0, key1; const long long* counter; int horizon, slot; long long id_offset;
    float* actions; float* neglogp; float* values; float* mus; float* sigmas; float* env_actions;
    int n;
};

template <int A>
__global__ __launch_bounds__(256) void policy_sample_kernel(const SampleArgs k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k.n) return;
    const uint32_t tick = (uint32_t)(*k.counter) * (uint32_t)k.horizon + (uint32_t)k.slot;
    const uint32_t env = (uint32_t)(k.id_offset + i);
    float z[6];
    ag::policy_normals<A>(env, tick, k.key0, k.key1, z);
    float h[A + 1], ls[A], act[A], mu[A], sigma[A], ea[A], nlp, v;
#pragma unroll
    for (int a = 0; a <= A; ++a) h[a] = k.heads[(size_t)i * (A + 1) + a];
#pragma unroll
    for (int a = 0; a < A; ++a) ls[a] = k.logstd[a];
    const bool denorm = k.vmean != nullptr;
    ag::policy_sample_row<A>(h, ls, z, denorm, denorm ? (float)k.vmean[0] : 0.f, denorm ? (float)k.vvar[0] : 1.f, k.veps, act, mu,
                             sigma, ea, nlp, v);
#pragma unroll
    for (int a = 0; a < A; ++a) {
        k.actions[(size_t)i * A + a] = act[a];
        k.mus[(size_t)i * A + a] = mu[a];
        k.sigmas[(size_t)i * A + a] = sigma[a];
        if (k.env_actions) k.env_actions[(size_t)i * A + a] = ea[a];
    }
    k.neglogp[i] = nlp;
    k.values[i] = v;
}

struct AccountArgs {
    const float* raw_reward; const unsigned char* dones; const unsigned char* timeouts; const float* values;
    float scale, shift, min_val, max_val; int log_val; float gamma;
    float* shaped; float* cur_rew; float* cur_shaped; float* cur_len; double* partials; int n;
};

__global__ __launch_bounds__(256) void rollout_account_kernel(const AccountArgs k) {
    __shared__ double red[4][4];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k.n) {
        const float r = k.raw_reward[i];
        float sh = ag::shape_reward(r, ag::ShapeParams{k.scale, k.shift, k.min_val, k.max_val, k.log_val, k.gamma});
        if (k.timeouts && k.timeouts[i]) sh += k.gamma * k.values[i];
        k.shaped[i] = sh;
        float cr = k.cur_rew[i] + r, cs = k.cur_shaped[i] + sh, cl = k.cur_len[i] + 1.0f;
        if (k.dones[i] != 0) {
            s[0] = 1.0; s[1] = cr; s[2] = cs; s[3] = cl;
            cr = cs = cl = 0.0f;
        }
        k.cur_rew[i] = cr; k.cur_shaped[i] = cs; k.cur_len[i] = cl;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < 4)
        k.partials[(size_t)blockIdx.x * 4 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                  const unsigned char* __restrict__ dones, const float* __restrict__ last_values,
                                                  float gamma, float tau, float* __restrict__ advs, float* __restrict__ returns,
                                                  int H, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lastgaelam = 0.f;
    float nextvalues = last_values[i];
    for (int t = H - 1; t >= 0; --t) {
        const size_t idx = (size_t)t * n + i;
        const float nnt = 1.0f - (float)dones[(size_t)(t + 1) * n + i];
        const float v = values[idx];
        const float delta = rewards[idx] + gamma * nextvalues * nnt - v;
        lastgaelam = delta + gamma * tau * nnt * lastgaelam;
        advs[idx] = lastgaelam;
        returns[idx] = lastgaelam + v;
        nextvalues = v;
    }
}

}  // namespace

extern "C" int ag_policy_sample(const float* heads, const float* logstd, const double* vmean, const double* vvar, float veps,
                                unsigned long long seed, const long long* counter, int horizon, int slot, long long id_offset,
                                float* actions, float* neglogp, float* values, float* mus, float* sigmas, float* env_actions,
                                int n, int A, void* stream) {
    if (!heads || !logstd || !counter || !actions || !neglogp || !values || !mus || !sigmas || n <= 0 || horizon <= 0)
        return AG_ERR_INVALID_ARG;
    if ((vmean == nullptr) != (vvar == nullptr)) return AG_ERR_INVALID_ARG;
    SampleArgs k{heads, logstd, vmean, vvar, veps, (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32), counter, horizon,
                 slot, id_offset, actions, neglogp, values, mus, sigmas, env_actions, n};
    const int grid = (n + 255) / 256;
    if (A == 4)
        hipLaunchKernelGGL(policy_sample_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    else if (A == 5)
        hipLaunchKernelGGL(policy_sample_kernel<5>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    else
        return AG_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_rollout_account_blocks(int n) { return (n + 255) / 256; }

extern "C" int ag_rollout_account(const float* raw_reward, const unsigned char* dones, const unsigned char* timeouts,
                                  const float* values, float scale, float shift, float min_val, float max_val, int log_val,
                                  float gamma, float* shaped, float* cur_rew, float* cur_shaped, float* cur_len,
                                  double* partials, int n, void* stream) {
    if (!raw_reward || !dones || !shaped || !cur_rew || !cur_shaped || !cur_len || !partials || n <= 0)
        return AG_ERR_INVALID_ARG;
    if (timeouts && !values) return AG_ERR_INVALID_ARG;
    AccountArgs k{raw_reward, dones, timeouts, values, scale, shift, min_val, max_val, log_val, gamma,
                  shaped, cur_rew, cur_shaped, cur_len, partials, n};
    hipLaunchKernelGGL(rollout_account_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_gae(const float* rewards, const float* values, const unsigned char* dones, const float* last_values, float gamma,
                      float tau, float* advs, float* returns, int H, int n, void* stream) {
    if (!rewards || !values || !dones || !last_values || !advs || !returns || H <= 0 || n <= 0) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gae_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, values, dones,
                       last_values, gamma, tau, advs, returns, H, n);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
