// rollout_math.hpp - per-env arithmetic of the rollout bookkeeping around one env step
// (A2CBase.play_steps, lib/agent/a2c_base.py:651-695), shared by the stand-alone kernels of rollout_kernels.hip
// (ag_policy_sample, ag_rollout_account) and by the fused rollout form of the env-step kernel (step_kernel.hip,
// ag_step_rollout_fused) so that both produce the same bits.
//   policy sampling : a = mu + sigma * N(0,1), neglogp, value de-normalisation, clamped env action
//                     (lib/model/a2c_continuous_logstd_model.py:159-167,195-198; lib/model/base_model.py:29-35;
//                      a2c_base.py:229-236 preprocess_actions)
//   reward shaping  : DefaultRewardsShaper (lib/utils/tr_helpers.py:16-42) + the time-out bootstrap (a2c_base.py:672-673)
#pragma once

#include "env_math.hpp"

namespace ag {

constexpr uint32_t kStreamPolicy = 16;   // Philox stream id of the action noise (env streams are 0..5)

// A standard normals for env `env` at rollout tick `tick` (tick = rollout counter * horizon + slot)
template <int A>
AG_HD void policy_normals(uint32_t env, uint32_t tick, uint32_t key0, uint32_t key1, float* z) {
    const U4 r = philox4x32_10(env, tick, kStreamPolicy, 0u, key0, key1);
    box_muller(r.x, r.y, z[0], z[1]);
    box_muller(r.z, r.w, z[2], z[3]);
    if (A > 4) {
        const U4 r2 = philox4x32_10(env, tick, kStreamPolicy, 1u, key0, key1);
        float unused;
        box_muller(r2.x, r2.y, z[4], unused);
    }
}

// One row of the policy head: h = (mu[0..A), normalised value).  Returns what the rollout buffer stores.
template <int A>
AG_HD void policy_sample_row(const float* h, const float* logstd, const float* z, bool denorm, float vmean, float vvar,
                             float veps, float* act, float* mu, float* sigma, float* env_act, float& neglogp, float& value) {
    float q = 0.f, ls_sum = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float ls = logstd[a];
        const float sg = expf(ls);
        const float m = h[a];
        const float x = m + sg * z[a];
        const float zz = (x - m) / sg;      // what the update recomputes from the stored action
        q += zz * zz;
        ls_sum += ls;
        act[a] = x;
        mu[a] = m;
        sigma[a] = sg;
        env_act[a] = fminf(fmaxf(x, -1.0f), 1.0f);
    }
    neglogp = 0.5f * q + 0.5f * 1.8378770664093453f * (float)A + ls_sum;
    float v = h[A];
    if (denorm) v = sqrtf(vvar + veps) * fminf(fmaxf(v, -5.0f), 5.0f) + vmean;
    value = v;
}

struct ShapeParams {
    float scale, shift, min_val, max_val;
    int log_val;
    float gamma;
};

AG_HD float shape_reward(float r, const ShapeParams& p) {
    float sh = (r + p.shift) * p.scale;
    sh = fminf(fmaxf(sh, p.min_val), p.max_val);
    if (p.log_val) sh = logf(sh);
    return sh;
}

}  // namespace ag
