// step_kernel.hip - the fused env-step kernels for gfx950 (MI355X).
// Compiled once per (task, ctl_mode):  hipcc -DAG_TASK=<0|1> -DAG_CTL=<0..4> -c step_kernel.hip
//
// One env per lane.  Per lane:
//   16-byte coalesced loads of the SoA state (kernel_args.hpp)  -> registers
//   env_step_physics / env_observe_reward / env_reset_done  (env_math.hpp: Hovering.step, hovering.py:286-308)
//   16-byte coalesced stores of the new state
//   observation rows ([n, NOBS] row-major, the layout the reference API exposes, base_task.py:73) staged through LDS
//   so that the tile leaves as full 16-byte-per-lane lines instead of 72-byte strided rows
//   one __ballot per wavefront publishes the done mask (hovering.py:300 nonzero) and lets a wavefront with no done
//   lane skip the reset path entirely.
//
// Kernels that live here:
//   step_kernel_ws2<TASK, CTL, false>  ag_step / ag_step_into / ag_step_rollout    (the env step alone, one step per launch)
//   step_kernel_multi<TASK, CTL>       ag_step_multi: K steps per launch, state in registers between them
//   step_kernel_ws2<TASK, CTL, true>   ag_step_rollout_fused: the same step with the rollout's policy sampling in front
//                                      of it and its reward / episode accounting behind it, in the same launch
//   step_kernel_ext<TASK, CTL>         ag_step_with_inputs (parity mode: random numbers supplied by the caller)
//   eval_obs_reward_kernel<TASK, CTL>  ag_eval_obs_reward (parity / inspection: obs + reward of the current state)
#include <hip/hip_runtime.h>

#include "kernel_args.hpp"
#include "rollout_math.hpp"

#ifndef AG_TASK
#error "compile with -DAG_TASK=<0|1> -DAG_CTL=<0..4>"
#endif

namespace ag {

// ---------------------------------------------------------------------------------------------------
// Parity mode (ag_step_with_inputs): one wavefront per 64 envs, the whole env_step() per lane, observation noise and
// reset uniforms read from caller arrays instead of Philox.  This is also the kernel the wave-specialised one is
// cross-checked against on the GPU (tests/test_gpu_parity.py): same device functions, different schedule.
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL>
__global__ __launch_bounds__(64) void step_kernel_ext(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int LDS_STRIDE = NOBS + 1;  // odd stride: conflict-free row writes (19 / 49 dwords)
    __shared__ float tile[64 * LDS_STRIDE];

    const int tid = threadIdx.x;
    const int i = blockIdx.x * 64 + tid;
    const bool active = i < k.n;

    EnvState s;
    CtlState c;
    load_env(k, i, s);          // arena is padded to a multiple of 256 envs: always in bounds
    load_ctl<CTL>(k, i, c);
    float pre_a[A], raw_a[A];
    {
        const float4 pa = k.PA[i];
        pre_a[0] = pa.x; pre_a[1] = pa.y; pre_a[2] = pa.z; pre_a[3] = pa.w;
        if (A == 5) pre_a[A - 1] = k.PA4[i];
    }
    float ext_z[18], ext_u[12];
    if (active) {
#pragma unroll
        for (int j = 0; j < A; ++j) raw_a[j] = k.actions[(size_t)i * A + j];
#pragma unroll
        for (int j = 0; j < 18; ++j) ext_z[j] = k.ext_noise[(size_t)i * 18 + j];
#pragma unroll
        for (int j = 0; j < 12; ++j) ext_u[j] = k.ext_uniforms[(size_t)i * 12 + j];
    } else {
#pragma unroll
        for (int j = 0; j < A; ++j) raw_a[j] = 0.0f;
#pragma unroll
        for (int j = 0; j < 18; ++j) ext_z[j] = 0.0f;
#pragma unroll
        for (int j = 0; j < 12; ++j) ext_u[j] = 0.5f;
    }

    StepParams P = k.P;          // uniform: stays in SGPRs
    P.tick = *k.tick_in;
    if (blockIdx.x == 0 && tid == 0) *k.tick_out = P.tick + 1u;

    float obs[NOBS];
    StepOut o;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    env_step<TASK, CTL, true>(s, c, pre_a, raw_a, P, env_global, ext_z, ext_u, obs, o);

    store_env(k, i, s);
    store_ctl<CTL>(k, i, c);
    k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
    if (A == 5) k.PA4[i] = pre_a[A - 1];

    const unsigned long long ballot = __ballot(active && o.done);
    if (active) {
        k.rew[i] = o.rew;
        k.reset[i] = (long long)o.done;
        k.timeout[i] = (uint8_t)o.timeout;
        if (tid == 0) k.mask[i >> 6] = ballot;
        if (k.cmd != nullptr) {
            k.cmd[i] = make_float4(o.cmd[0], o.cmd[1], o.cmd[2], o.cmd[3]);
#pragma unroll
            for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
        }
    }

#pragma unroll
    for (int j = 0; j < NOBS; ++j) tile[tid * LDS_STRIDE + j] = obs[j];
    __syncthreads();
    const int block_env0 = blockIdx.x * 64;
    const int valid = min(64, k.n - block_env0) * NOBS;  // floats of this block that exist
    float* out = k.obs + (size_t)block_env0 * NOBS;
    constexpr int NV4 = 64 * NOBS / 4;
    constexpr int ITERS = (NV4 + 63) / 64;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int m = tid + it * 64;
        if (m >= NV4) break;
        const int e = 4 * m;
        if (e + 3 < valid) {
            float4 v;
            v.x = tile[(e + 0) / NOBS * LDS_STRIDE + (e + 0) % NOBS];
            v.y = tile[(e + 1) / NOBS * LDS_STRIDE + (e + 1) % NOBS];
            v.z = tile[(e + 2) / NOBS * LDS_STRIDE + (e + 2) % NOBS];
            v.w = tile[(e + 3) / NOBS * LDS_STRIDE + (e + 3) % NOBS];
            reinterpret_cast<float4*>(out)[m] = v;
        } else {
            for (int q = e; q < e + 4 && q < valid; ++q) out[q] = tile[q / NOBS * LDS_STRIDE + q % NOBS];
        }
    }
}

// Sum over the 64 lanes of a wave in ONE fixed order, shared by every step kernel (so that the per-tile reward-term sums - the
// Episode/<term> logs - are bit-identical whichever kernel produced them).  row_bcast15 / row_bcast31 are gfx9 DPP controls.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "sm_wave_sum uses the gfx9 row_bcast DPP controls"
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sm_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
// sum over the 64 lanes in a fixed order; valid in lane 63
__device__ __forceinline__ float sm_wave_sum(float v) {
    v = sm_dpp_add<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v = sm_dpp_add<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v = sm_dpp_add<0x141, 0xF>(v);       // row_half_mirror
    v = sm_dpp_add<0x140, 0xF>(v);       // row_mirror
    v = sm_dpp_add<0x142, 0xA>(v);       // row_bcast15: rows 1, 3 += lane 15 of rows 0, 2
    v = sm_dpp_add<0x143, 0xC>(v);       // row_bcast31: rows 2, 3 += lane 31
    return v;
}

// ---------------------------------------------------------------------------------------------------
// The shipped kernel.  128-thread workgroup = two wavefronts for the same 64 envs:
//   physics wave: state I/O, controller, RK4, reward / termination, reset; after barrier 1 it forms
//                 (clean + sigma*z) - target in registers (target lives in SGPRs) and writes FINAL observation rows into
//                 an LDS tile laid out exactly like the HBM rows;
//   noise wave  : Philox4x32-10 + Box-Muller for the 18 noisy columns -> sigma*z rows in LDS; between barriers 1 and 2
//                 (while the physics wave adds the noise) the per-tile sums of the nine reward terms;
//   both        : copy the tile out as ds_read_b128 -> global_store_dwordx4 (no per-element index arithmetic).
// Why two waves: at 65 536 envs a one-env-per-lane launch is 1 024 waves on 1 024 SIMDs; a lone wave issues about one
// VALU instruction per 4 cycles (the SIMD-32 needs a second instruction stream to reach one per 2), so the launch is
// bound by ONE wave's instruction count, a quarter of which is observation noise that does not depend on the physics.
// Rollout form (k.reset_u8): done flags as u8 and per-tile sums of the reward terms instead of nine f32[N] arrays.
//
// FUSED (ag_step_rollout_fused) = one step of A2CBase.play_steps (a2c_base.py:651-695) behind the policy GEMMs:
//   noise wave, before barrier 0: a = mu + sigma * N(0,1), neglogp, value de-normalisation (ag_policy_sample's arithmetic,
//                 rollout_math.hpp) -> actions / mus / sigmas / neglogp / values to the rollout slot, the clamped env
//                 action to LDS; the physics wave's state loads are in flight meanwhile;
//   noise wave, between barriers 1 and 2: reward shaping (+ time-out bootstrap), running episode reward / length, the
//                 per-tile sums over the episodes that ended (ag_rollout_account's arithmetic).
// Two launches and their two dependent-launch boundaries per rollout step disappear, the action never goes through HBM.
// Arithmetic and its order are those of env_step(): results are bit-identical to step_kernel_ext on the same inputs.
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL, bool FUSED>
__global__ __launch_bounds__(128) void step_kernel_ws2(const KArgs k, const TailArgs ta) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int SB = 19;                       // odd stride: conflict-free row access of the noise tile
    constexpr int ST = 9;
    constexpr int SA = (A == 4) ? 4 : 5;         // env-action rows: one ds_write_b128 per lane (A = 4) or odd stride 5
    __shared__ __attribute__((aligned(16))) float tileO[64 * NOBS];   // final observation rows, HBM order
    __shared__ float tileB[64 * SB];
    __shared__ float tileT[64 * ST];
    __shared__ __attribute__((aligned(16))) float tileA[FUSED ? 64 * SA : 4];
    __shared__ float tileR[FUSED ? 64 : 1];
    __shared__ int tileD[FUSED ? 64 : 1];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const bool active = i < k.n;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    const bool want_terms = (k.term_sums != nullptr);

    if (wave == 0) {
        EnvState s;
        CtlState c;
        load_env(k, i, s);
        load_ctl<CTL>(k, i, c);
        float pre_a[A], raw_a[A], a[A];
        {
            const float4 pa = k.PA[i];
            pre_a[0] = pa.x; pre_a[1] = pa.y; pre_a[2] = pa.z; pre_a[3] = pa.w;
            if (A == 5) pre_a[A - 1] = k.PA4[i];
        }
        if (FUSED) {
            __syncthreads();   // barrier 0: this step's clamped actions are in tileA (zeros for padding lanes)
            if (A == 4) {
                const float4 av = reinterpret_cast<const float4*>(tileA)[lane];
                raw_a[0] = av.x; raw_a[1] = av.y; raw_a[2] = av.z; raw_a[3] = av.w;
            } else {
#pragma unroll
                for (int j = 0; j < A; ++j) raw_a[j] = tileA[lane * SA + j];
            }
        } else if (active) {
            if (A == 4) {
                const float4 av = reinterpret_cast<const float4*>(k.actions)[i];
                raw_a[0] = av.x; raw_a[1] = av.y; raw_a[2] = av.z; raw_a[3] = av.w;
            } else {
#pragma unroll
                for (int j = 0; j < A; ++j) raw_a[j] = k.actions[(size_t)i * A + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) raw_a[j] = 0.0f;
        }
        StepOut o;
        env_step_physics<TASK, CTL>(s, c, raw_a, P, a, o.cmd);
        float obs[NOBS];
        env_observe_reward<TASK, CTL, false, true>(s, a, pre_a, o.cmd, P, env_global, nullptr, obs, o);
#pragma unroll
        for (int j = 0; j < A; ++j) pre_a[j] = a[j];
        const int progress_end = s.progress;      // progress_buf after the increment, before reset_idx zeroes it
        s.was_reset = o.done;
        if (o.done) env_reset_done<CTL, false>(s, c, pre_a, P, env_global, nullptr);
        o.timeout = step_timeout(progress_end, s.progress, P);
        store_env(k, i, s);
        store_ctl<CTL>(k, i, c);
        k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
        if (A == 5) k.PA4[i] = pre_a[A - 1];
        const unsigned long long ballot = __ballot(active && o.done);
        if (active) {
            k.rew[i] = o.rew;
            if (k.reset_u8 != nullptr) k.reset_u8[i] = (uint8_t)o.done;
            else k.reset[i] = (long long)o.done;
            k.timeout[i] = (uint8_t)o.timeout;
            if (k.timeout_steps != nullptr) k.timeout_steps[i] = (uint8_t)o.timeout;
            if (lane == 0) k.mask[i >> 6] = ballot;
            if (k.cmd != nullptr) {
                k.cmd[i] = make_float4(o.cmd[0], o.cmd[1], o.cmd[2], o.cmd[3]);
#pragma unroll
                for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
            }
        }
        if (want_terms) {
#pragma unroll
            for (int t = 0; t < 9; ++t) tileT[lane * ST + t] = active ? o.terms[t] : 0.0f;
        }
        if (FUSED) {
            tileR[lane] = o.rew;
            tileD[lane] = (o.done ? 1 : 0) | (o.timeout ? 2 : 0);
        }
        __syncthreads();   // barrier 1: sigma*z rows are in tileB; reward / done / terms of this tile are in LDS
        // obs = (clean + sigma*z) - target for the 18 noisy columns (hovering.py:343-345), clean elsewhere
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            float v = __fadd_rn(obs[j], tileB[lane * SB + j]);      // an add of its own (never contracted with the product that
            if (TASK == TASK_HOVERING) v -= P.target[j];            // formed obs[j]): the K-step kernel adds it in another wave
            obs[j] = v;
        }
        if (NOBS % 4 == 0) {
#pragma unroll
            for (int j = 0; j < NOBS / 4; ++j)
                reinterpret_cast<float4*>(tileO)[lane * (NOBS / 4) + j] = make_float4(obs[4 * j], obs[4 * j + 1], obs[4 * j + 2], obs[4 * j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < NOBS / 2; ++j)
                reinterpret_cast<float2*>(tileO)[lane * (NOBS / 2) + j] = make_float2(obs[2 * j], obs[2 * j + 1]);
        }
    } else {
        // ---- rollout head: sample this step's action from the policy heads (FUSED only)
        float value = 0.0f, cur_r = 0.0f, cur_s = 0.0f, cur_l = 0.0f;
        if (FUSED) {
            float ea[A];
#pragma unroll
            for (int j = 0; j < A; ++j) ea[j] = 0.0f;
            if (active) {
                cur_r = ta.cur_rew[i]; cur_s = ta.cur_shaped[i]; cur_l = ta.cur_len[i];   // consumed after barrier 1
                float h[A + 1], ls[A], z[6], act[A], mu[A], sigma[A], nlp;
#pragma unroll
                for (int j = 0; j <= A; ++j) h[j] = ta.heads[(size_t)i * (A + 1) + j];
#pragma unroll
                for (int j = 0; j < A; ++j) ls[j] = ta.logstd[j];
                const uint32_t ptick = (uint32_t)(*ta.counter) * (uint32_t)ta.horizon + (uint32_t)ta.slot;
                policy_normals<A>((uint32_t)(ta.id_offset + i), ptick, ta.key0, ta.key1, z);
                const bool denorm = ta.vmean != nullptr;
                policy_sample_row<A>(h, ls, z, denorm, denorm ? (float)ta.vmean[0] : 0.f, denorm ? (float)ta.vvar[0] : 1.f,
                                     ta.veps, act, mu, sigma, ea, nlp, value);
                if (A == 4) {
                    reinterpret_cast<float4*>(ta.actions)[i] = make_float4(act[0], act[1], act[2], act[3]);
                    reinterpret_cast<float4*>(ta.mus)[i] = make_float4(mu[0], mu[1], mu[2], mu[3]);
                    reinterpret_cast<float4*>(ta.sigmas)[i] = make_float4(sigma[0], sigma[1], sigma[2], sigma[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < A; ++j) {
                        ta.actions[(size_t)i * A + j] = act[j];
                        ta.mus[(size_t)i * A + j] = mu[j];
                        ta.sigmas[(size_t)i * A + j] = sigma[j];
                    }
                }
                ta.neglogp[i] = nlp;
                ta.values[i] = value;
            }
            if (A == 4) {
                reinterpret_cast<float4*>(tileA)[lane] = make_float4(ea[0], ea[1], ea[2], ea[3]);
            } else {
#pragma unroll
                for (int j = 0; j < A; ++j) tileA[lane * SA + j] = ea[j];
            }
            __syncthreads();   // barrier 0
        }
        // ---- observation noise
        float z[18];
        if (!P.noise_off) {
            obs_noise_normals(P, env_global, z);
        } else {
#pragma unroll
            for (int j = 0; j < 18; ++j) z[j] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 18; ++j) tileB[lane * SB + j] = noise_sigma(j) * z[j];
        __syncthreads();   // barrier 1
        // ---- between the barriers (the physics wave is adding the noise): reductions over the tile
        if (want_terms) {  // per-tile sums of the nine reward terms: nine wave sums, fixed order (as step_kernel_multi)
            float tsum[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) tsum[t] = sm_wave_sum(tileT[lane * ST + t]);
            if (lane == 63) {
#pragma unroll
                for (int t = 0; t < 9; ++t) k.term_sums[(size_t)blockIdx.x * 12 + t] = tsum[t];
            }
        }
        if (FUSED) {       // rollout tail: reward shaping + episode accounting (a2c_base.py:668-695)
            double sums[4] = {0.0, 0.0, 0.0, 0.0};
            if (active) {
                const float r = tileR[lane];
                const int flags = tileD[lane];
                const ShapeParams sp{ta.scale, ta.shift, ta.min_val, ta.max_val, ta.log_val, ta.gamma};
                float sh = shape_reward(r, sp);
                if (ta.bootstrap && (flags & 2)) sh += ta.gamma * value;
                ta.shaped[i] = sh;
                float cr = cur_r + r, cs = cur_s + sh, cl = cur_l + 1.0f;
                if (flags & 1) {
                    sums[0] = 1.0; sums[1] = cr; sums[2] = cs; sums[3] = cl;
                    cr = cs = cl = 0.0f;
                }
                ta.cur_rew[i] = cr; ta.cur_shaped[i] = cs; ta.cur_len[i] = cl;
            }
            // episodes that ended in this tile: skip the 24 double shuffles when no lane has one (the usual case)
            if (__ballot(sums[0] != 0.0) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    for (int off = 32; off > 0; off >>= 1) sums[j] += __shfl_down(sums[j], off, 64);
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ta.partials[(size_t)blockIdx.x * 4 + j] = sums[j];
            }
        }
    }
    __syncthreads();       // barrier 2: final rows are in tileO

    const int block_env0 = blockIdx.x * 64;
    const int valid = min(64, k.n - block_env0) * NOBS;    // floats of this tile that exist
    float* out = k.obs + (size_t)block_env0 * NOBS;
    constexpr int NV4 = 64 * NOBS / 4;
    constexpr int ITERS = (NV4 + 127) / 128;
    const int t2 = wave * 64 + lane;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int m = t2 + it * 128;
        if (m < NV4) {
            const int e = 4 * m;
            if (e + 3 < valid) {
                reinterpret_cast<float4*>(out)[m] = reinterpret_cast<const float4*>(tileO)[m];
            } else {
                for (int q = e; q < e + 4 && q < valid; ++q) out[q] = tileO[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// step_kernel_multi: K env steps per launch (ag_step_multi; K = 1: ag_step / ag_step_into / ag_step_rollout).  Same two
// wavefronts per 64 envs and the same device functions in the same order as step_kernel_ws2, but organised around ONE
// workgroup barrier per step so that nothing except the arithmetic of the step is on the physics wave's path:
//   physics wave: state / controller memory / previous action loaded once, kept in registers for the K steps, stored once.
//                 Per step: action from LDS -> env_step_physics -> env_observe_reward -> in-place reset -> reward / done
//                 stores -> CLEAN observation row + the nine reward terms to LDS -> barrier -> next step.  It issues no global
//                 load inside the loop (loads and stores share vmcnt on gfx950: a wave that consumes a load behind stores
//                 waits for the stores too), so its stores never stall it.
//   noise wave  : loads the actions one step ahead and hands them over through LDS; Philox + Box-Muller for step kk BEFORE
//                 the barrier; behind it (while the physics wave already runs step kk + 1) it forms (clean + sigma z) - target
//                 rows (hovering.py:343-345), lays them out in HBM order, copies the tile out as ds_read_b128 ->
//                 global_store_dwordx4 and reduces the reward terms of the tile (nine wave sums by DPP, fixed order).
// LDS tiles the physics wave writes (clean rows, terms, and the action buffer it reads) are double-buffered by step parity: the
// noise wave works on step kk's while the physics wave fills step kk + 1's, and reaches barrier kk + 1 only when it is done.
// Results: bit-identical for any split of a step sequence into launches (one instantiation serves every K).
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL>
__global__ __launch_bounds__(128) void step_kernel_multi(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int SC = (NOBS & 1) ? NOBS : NOBS + 1;     // odd row stride of the clean-observation tile: conflict-free rows
    constexpr int ST = 9;
    constexpr int SA = (A == 4) ? 4 : 5;
    __shared__ __attribute__((aligned(16))) float tileO[64 * NOBS];        // final rows, HBM order (noise wave only)
    __shared__ float tileC[2][64 * SC];                                    // clean observation rows of step parity 0 / 1
    __shared__ float tileT[2][64 * ST];                                    // reward terms
    __shared__ __attribute__((aligned(16))) float tileA[2][64 * SA];       // env actions

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const bool active = i < k.n;
    StepParams P = k.P;
    const int K = k.num_steps;
    const uint32_t tick0 = *k.tick_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *k.tick_out = tick0 + (uint32_t)K;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    const bool want_terms = (k.term_sums != nullptr);
    const size_t nsz = (size_t)k.n;
    const int ntiles = (k.n + 63) >> 6;

    if (wave == 0) {
        // =============================================================== physics wave
        EnvState s;
        CtlState c;
        float pre_a[A];
        load_env(k, i, s);
        load_ctl<CTL>(k, i, c);
        {
            const float4 pa = k.PA[i];
            pre_a[0] = pa.x; pre_a[1] = pa.y; pre_a[2] = pa.z; pre_a[3] = pa.w;
            if (A == 5) pre_a[A - 1] = k.PA4[i];
        }
        __syncthreads();                         // step 0's action is in tileA[0]
        // everything requested so far has landed BEFORE the loop: left to the compiler the wait sits at the first use inside
        // the loop body, where from the second step on it would wait for the previous step's stores
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#pragma unroll 1
        for (int kk = 0; kk < K; ++kk) {
            P.tick = tick0 + (uint32_t)kk;
            const bool last = (kk == K - 1);
            const int par = kk & 1;
            float raw_a[A], a[A];
            if (A == 4) {
                const float4 av = reinterpret_cast<const float4*>(tileA[par])[lane];
                raw_a[0] = av.x; raw_a[1] = av.y; raw_a[2] = av.z; raw_a[3] = av.w;
            } else {
#pragma unroll
                for (int j = 0; j < A; ++j) raw_a[j] = tileA[par][lane * SA + j];
            }
            StepOut o;
            env_step_physics<TASK, CTL>(s, c, raw_a, P, a, o.cmd);
            float obs[NOBS];
            env_observe_reward<TASK, CTL, false, true>(s, a, pre_a, o.cmd, P, env_global, nullptr, obs, o);
#pragma unroll
            for (int j = 0; j < A; ++j) pre_a[j] = a[j];
            const int progress_end = s.progress;      // progress_buf after the increment, before reset_idx zeroes it
            s.was_reset = o.done;
            if (o.done) env_reset_done<CTL, false>(s, c, pre_a, P, env_global, nullptr);
            o.timeout = step_timeout(progress_end, s.progress, P);
            if (last) {            // the state goes back to HBM once per launch
                store_env(k, i, s);
                store_ctl<CTL>(k, i, c);
                k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
                if (A == 5) k.PA4[i] = pre_a[A - 1];
            }
            const unsigned long long ballot = __ballot(active && o.done);
            if (active) {
                k.rew[(size_t)kk * nsz + i] = o.rew;
                if (k.reset_u8 != nullptr) k.reset_u8[(size_t)kk * nsz + i] = (uint8_t)o.done;
                else k.reset[i] = (long long)o.done;
                if (k.timeout_steps != nullptr) k.timeout_steps[(size_t)kk * nsz + i] = (uint8_t)o.timeout;
                if (last) {
                    k.timeout[i] = (uint8_t)o.timeout;
                    if (lane == 0) k.mask[i >> 6] = ballot;
                }
                if (k.cmd != nullptr) {
                    k.cmd[i] = make_float4(o.cmd[0], o.cmd[1], o.cmd[2], o.cmd[3]);
#pragma unroll
                    for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
                }
            }
#pragma unroll
            for (int j = 0; j < NOBS; ++j) tileC[par][lane * SC + j] = obs[j];
            if (want_terms) {
#pragma unroll
                for (int t = 0; t < 9; ++t) tileT[par][lane * ST + t] = active ? o.terms[t] : 0.0f;
            }
            __syncthreads();       // barrier kk: clean rows / terms of step kk are in LDS; step kk + 1's action is in tileA[par ^ 1]
        }
    } else {
        // =============================================================== noise wave
        float raw_next[A];
#define AG_LOAD_ACTION(step_)                                                                          \
        do {                                                                                           \
            if (active) {                                                                              \
                const float* an_ = k.actions + (size_t)(step_) * nsz * A;                              \
                if (A == 4) {                                                                          \
                    const float4 av_ = reinterpret_cast<const float4*>(an_)[i];                        \
                    raw_next[0] = av_.x; raw_next[1] = av_.y; raw_next[2] = av_.z; raw_next[3] = av_.w; \
                } else {                                                                               \
                    _Pragma("unroll") for (int j = 0; j < A; ++j) raw_next[j] = an_[(size_t)i * A + j]; \
                }                                                                                      \
            } else {                                                                                   \
                _Pragma("unroll") for (int j = 0; j < A; ++j) raw_next[j] = 0.0f;                      \
            }                                                                                          \
        } while (0)
#define AG_PUT_ACTION(buf_)                                                                            \
        do {                                                                                           \
            if (A == 4) {                                                                              \
                reinterpret_cast<float4*>(tileA[buf_])[lane] = make_float4(raw_next[0], raw_next[1], raw_next[2], raw_next[3]); \
            } else {                                                                                   \
                _Pragma("unroll") for (int j = 0; j < A; ++j) tileA[buf_][lane * SA + j] = raw_next[j]; \
            }                                                                                          \
        } while (0)
        AG_LOAD_ACTION(0);
        AG_PUT_ACTION(0);                  // step 0's action
        if (K > 1) AG_LOAD_ACTION(1);      // step 1's is requested now and handed over during step 0
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < K; ++kk) {
            P.tick = tick0 + (uint32_t)kk;
            const int par = kk & 1;
            float* obs_out = k.obs + (size_t)kk * nsz * NOBS;
            if (kk + 1 < K) {
                // step kk + 1's action (requested a whole step ago) goes to the other buffer - the physics wave reads it behind
                // barrier kk - and step kk + 2's is requested.  Consume first, then issue: the wait must not cover the new load.
                AG_PUT_ACTION(par ^ 1);
                if (kk + 2 < K) AG_LOAD_ACTION(kk + 2);
            }
            float z[18];
            if (!P.noise_off) {
                obs_noise_normals(P, env_global, z);
            } else {
#pragma unroll
                for (int j = 0; j < 18; ++j) z[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 18; ++j) z[j] = noise_sigma(j) * z[j];      // a product of its own, as in step_kernel_ws2 (no FMA)
            __syncthreads();       // barrier kk
            // obs = (clean + sigma*z) - target for the 18 noisy columns (hovering.py:343-345), clean elsewhere
            float obs[NOBS];
#pragma unroll
            for (int j = 0; j < NOBS; ++j) obs[j] = tileC[par][lane * SC + j];
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                float v = __fadd_rn(obs[j], z[j]);
                if (TASK == TASK_HOVERING) v -= P.target[j];
                obs[j] = v;
            }
            if (NOBS % 4 == 0) {
#pragma unroll
                for (int j = 0; j < NOBS / 4; ++j)
                    reinterpret_cast<float4*>(tileO)[lane * (NOBS / 4) + j] = make_float4(obs[4 * j], obs[4 * j + 1], obs[4 * j + 2], obs[4 * j + 3]);
            } else {
#pragma unroll
                for (int j = 0; j < NOBS / 2; ++j)
                    reinterpret_cast<float2*>(tileO)[lane * (NOBS / 2) + j] = make_float2(obs[2 * j], obs[2 * j + 1]);
            }
            __builtin_amdgcn_wave_barrier();       // one wave: its own LDS writes are ordered before its reads below
            const int block_env0 = blockIdx.x * 64;
            const int valid = min(64, k.n - block_env0) * NOBS;    // floats of this tile that exist
            float* out = obs_out + (size_t)block_env0 * NOBS;
            constexpr int NV4 = 64 * NOBS / 4;
            constexpr int ITERS = (NV4 + 63) / 64;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int m = lane + it * 64;
                if (m < NV4) {
                    const int e = 4 * m;
                    if (e + 3 < valid) {
                        reinterpret_cast<float4*>(out)[m] = reinterpret_cast<const float4*>(tileO)[m];
                    } else {
                        for (int q = e; q < e + 4 && q < valid; ++q) out[q] = tileO[q];
                    }
                }
            }
            if (want_terms) {      // per-tile sums of the nine reward terms: nine wave sums, fixed order
                float* term_out = k.term_sums + ((size_t)kk * ntiles + blockIdx.x) * 12;
                float tsum[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) tsum[t] = sm_wave_sum(tileT[par][lane * ST + t]);
                if (lane == 63) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) term_out[t] = tsum[t];
                }
            }
            __builtin_amdgcn_wave_barrier();       // tileO is rewritten next step only after the reads above were issued
        }
#undef AG_LOAD_ACTION
#undef AG_PUT_ACTION
    }
}

// ---------------------------------------------------------------------------------------------------
// Parity / inspection kernel (ag_eval_obs_reward): compute_observations + compute_quadcopter_reward of the reference
// (hovering.py:337-459, tracking.py:202-296) on the handle's CURRENT state, with the processed actions and the
// controller output supplied by the caller - the exact inputs the golden fixtures recorded from the reference's own
// methods.  Same device functions (and therefore the same v_rcp / v_rsq / v_sqrt / v_exp code) as the step kernel.
// Writes obs / reward / done / reward terms; the state is not modified.
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL>
__global__ __launch_bounds__(64) void eval_obs_reward_kernel(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= k.n) return;
    EnvState s;
    load_env(k, i, s);
    float pre_a[A], a[A], cmd[4], z[18];
    {
        const float4 pa = k.PA[i];
        pre_a[0] = pa.x; pre_a[1] = pa.y; pre_a[2] = pa.z; pre_a[3] = pa.w;
        if (A == 5) pre_a[A - 1] = k.PA4[i];
    }
#pragma unroll
    for (int j = 0; j < A; ++j) a[j] = k.eval_actions[(size_t)i * A + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) cmd[j] = k.eval_cmd[(size_t)i * 4 + j];
#pragma unroll
    for (int j = 0; j < 18; ++j) z[j] = (k.ext_noise != nullptr) ? k.ext_noise[(size_t)i * 18 + j] : 0.0f;
    StepParams P = k.P;
    P.noise_off = 0;
    float obs[NOBS];
    StepOut o;
    env_observe_reward<TASK, CTL, true, false>(s, a, pre_a, cmd, P, 0u, z, obs, o);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) k.obs[(size_t)i * NOBS + j] = obs[j];
    k.rew[i] = o.rew;
    k.reset[i] = (long long)o.done;
    if (k.cmd != nullptr) {
        k.cmd[i] = make_float4(cmd[0], cmd[1], cmd[2], cmd[3]);
#pragma unroll
        for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
    }
}

#define AG_ECAT_(a, b, c) launch_eval_##a##_##b
#define AG_ECAT(a, b) AG_ECAT_(a, b, 0)
hipError_t AG_ECAT(AG_TASK, AG_CTL)(const KArgs& k, hipStream_t stream) {
    hipLaunchKernelGGL((eval_obs_reward_kernel<AG_TASK, AG_CTL>), dim3((k.n + 63) / 64), dim3(64), 0, stream, k);
    return hipGetLastError();
}

#define AG_CAT_(a, b, c) launch_step_##a##_##b
#define AG_CAT(a, b) AG_CAT_(a, b, 0)
hipError_t AG_CAT(AG_TASK, AG_CTL)(const KArgs& k, const TailArgs* tail, hipStream_t stream) {
    const dim3 g64((k.n + 63) / 64);
    if (k.ext_noise != nullptr) {   // parity mode
        if (tail != nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL((step_kernel_ext<AG_TASK, AG_CTL>), g64, dim3(64), 0, stream, k);
    } else if (tail != nullptr) {
        hipLaunchKernelGGL((step_kernel_ws2<AG_TASK, AG_CTL, true>), g64, dim3(128), 0, stream, k, *tail);
    } else if (k.num_steps == 1 && !k.force_multi) {
        // one step per launch (ag_step / ag_step_into / ag_step_rollout): the dedicated two-wave kernel - the physics wave reads
        // its action itself (no hand-over barrier in front of the physics, no double-buffered tiles); bit-identical to
        // step_kernel_multi with K = 1 (tests/test_gpu_step_multi.py), 8 % faster per launch
        hipLaunchKernelGGL((step_kernel_ws2<AG_TASK, AG_CTL, false>), g64, dim3(128), 0, stream, k, TailArgs{});
    } else {
        hipLaunchKernelGGL((step_kernel_multi<AG_TASK, AG_CTL>), g64, dim3(128), 0, stream, k);
    }
    return hipGetLastError();
}

}  // namespace ag
