// step_kernel.hip - the fused env-step kernel for gfx950 (MI355X).
// Compiled once per (task, ctl_mode):  hipcc -DAG_TASK=<0|1> -DAG_CTL=<0..4> -c step_kernel.hip
//
// One env per lane, 64 envs per wavefront.  Per lane:
//   16-byte coalesced loads of the SoA state (kernel_args.hpp)  -> registers
//   env_step<TASK,CTL>()  (env_math.hpp: Hovering.step, hovering.py:286-308)
//   16-byte coalesced stores of the new state
//   observation rows ([n, NOBS] row-major, the layout the reference API exposes,
//   base_task.py:73) staged through LDS so that the block writes its NOBS*BLOCK
//   contiguous floats as full 16-byte-per-lane lines instead of 72-byte strided rows
//   one __ballot per wavefront publishes the done mask (hovering.py:300 nonzero) and lets a
//   wavefront with no done lane skip the reset path entirely.
#include <hip/hip_runtime.h>

#include "kernel_args.hpp"

#ifndef AG_TASK
#error "compile with -DAG_TASK=<0|1> -DAG_CTL=<0..4>"
#endif

namespace ag {

template <int TASK, int CTL, int BLOCK, bool OBS_LDS, bool EXT>
__global__ __launch_bounds__(BLOCK) void step_kernel(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int LDS_STRIDE = NOBS + 1;  // odd stride: conflict-free row writes (19 / 49 dwords)
    __shared__ float tile[OBS_LDS ? BLOCK * LDS_STRIDE : 1];

    const int tid = threadIdx.x;
    const int i = blockIdx.x * BLOCK + tid;
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
    if (active) {
        if (A == 4) {
            const float4 a = reinterpret_cast<const float4*>(k.actions)[i];
            raw_a[0] = a.x; raw_a[1] = a.y; raw_a[2] = a.z; raw_a[3] = a.w;
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) raw_a[j] = k.actions[(size_t)i * A + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < A; ++j) raw_a[j] = 0.0f;
    }

    // parity mode: random numbers supplied by the caller
    float ext_z[EXT ? 18 : 1], ext_u[EXT ? 12 : 1];
    if (EXT) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 18; ++j) ext_z[j] = k.ext_noise[(size_t)i * 18 + j];
#pragma unroll
            for (int j = 0; j < 12; ++j) ext_u[j] = k.ext_uniforms[(size_t)i * 12 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 18; ++j) ext_z[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < 12; ++j) ext_u[j] = 0.5f;
        }
    }

    StepParams P = k.P;          // uniform: stays in SGPRs
    P.tick = *k.tick_in;
    if (blockIdx.x == 0 && tid == 0) *k.tick_out = P.tick + 1u;

    float obs[NOBS];
    StepOut o;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    env_step<TASK, CTL, EXT>(s, c, pre_a, raw_a, P, env_global, ext_z, ext_u, obs, o);

    // ---- state back to HBM
    store_env(k, i, s);
    store_ctl<CTL>(k, i, c);
    k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
    if (A == 5) k.PA4[i] = pre_a[A - 1];

    // ---- per-env outputs
    const unsigned long long ballot = __ballot(active && o.done);
    if (active) {
        k.rew[i] = o.rew;
        k.reset[i] = (long long)o.done;
        k.timeout[i] = (uint8_t)o.timeout;
        if ((tid & 63) == 0) k.mask[i >> 6] = ballot;
        if (k.cmd != nullptr) {
            k.cmd[i] = make_float4(o.cmd[0], o.cmd[1], o.cmd[2], o.cmd[3]);
#pragma unroll
            for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
        }
    }

    // ---- observations
    if (OBS_LDS) {
#pragma unroll
        for (int j = 0; j < NOBS; ++j) tile[tid * LDS_STRIDE + j] = obs[j];
        __syncthreads();
        const int block_env0 = blockIdx.x * BLOCK;
        const int valid = min(BLOCK, k.n - block_env0) * NOBS;  // floats of this block that exist
        float* out = k.obs + (size_t)block_env0 * NOBS;
        constexpr int NV4 = BLOCK * NOBS / 4;
        constexpr int ITERS = (NV4 + BLOCK - 1) / BLOCK;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int m = tid + it * BLOCK;
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
    } else if (active) {
        float* out = k.obs + (size_t)i * NOBS;
        if (NOBS % 4 == 0) {
#pragma unroll
            for (int j = 0; j < NOBS / 4; ++j)
                reinterpret_cast<float4*>(out)[j] = make_float4(obs[4 * j], obs[4 * j + 1], obs[4 * j + 2], obs[4 * j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < NOBS / 2; ++j) reinterpret_cast<float2*>(out)[j] = make_float2(obs[2 * j], obs[2 * j + 1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised variant: 128-thread workgroup = 2 wavefronts for the same 64 envs.
//   wave 0 ("physics"): state I/O, controller, RK4, reward/termination, reset  -> clean obs row to LDS tile A
//   wave 1 ("noise")  : Philox4x32-10 + Box-Muller for the 18 noisy columns    -> sigma*z row to LDS tile B
//   barrier, then all 128 lanes write (A + B) - target as 16-byte-per-lane contiguous lines.
// Why: at 65 536 envs a one-env-per-lane launch is 1 024 waves on 1 024 SIMDs; a lone wave issues about one
// VALU instruction per 4 cycles (the SIMD-32 needs a second wave to reach 2), so the kernel is bound by ONE
// wave's instruction count, ~25 % of which is observation noise that does not depend on the physics.  Splitting
// it off halves the critical path and puts two waves on every SIMD (profiles/r01_env_kernel_pmc.md, N-scaling).
// Results are bit-identical to step_kernel<...> (same expressions, same order).
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL>
__global__ __launch_bounds__(128) void step_kernel_ws(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int SA = NOBS + 1;   // odd strides: conflict-free row writes
    constexpr int SB = 19;
    __shared__ float tileA[64 * SA];
    __shared__ float tileB[64 * SB];
    __shared__ float tgt[18];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const bool active = i < k.n;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;

    if (wave == 0) {
        EnvState s;
        CtlState c;
        load_env(k, i, s);
        load_ctl<CTL>(k, i, c);
        float pre_a[A], raw_a[A];
        {
            const float4 pa = k.PA[i];
            pre_a[0] = pa.x; pre_a[1] = pa.y; pre_a[2] = pa.z; pre_a[3] = pa.w;
            if (A == 5) pre_a[A - 1] = k.PA4[i];
        }
        if (active) {
            if (A == 4) {
                const float4 a = reinterpret_cast<const float4*>(k.actions)[i];
                raw_a[0] = a.x; raw_a[1] = a.y; raw_a[2] = a.z; raw_a[3] = a.w;
            } else {
#pragma unroll
                for (int j = 0; j < A; ++j) raw_a[j] = k.actions[(size_t)i * A + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) raw_a[j] = 0.0f;
        }
        float obs[NOBS];
        StepOut o;
        env_step<TASK, CTL, false, true>(s, c, pre_a, raw_a, P, env_global, nullptr, nullptr, obs, o);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) tileA[lane * SA + j] = obs[j];
        store_env(k, i, s);
        store_ctl<CTL>(k, i, c);
        k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
        if (A == 5) k.PA4[i] = pre_a[A - 1];
        const unsigned long long ballot = __ballot(active && o.done);
        if (active) {
            k.rew[i] = o.rew;
            k.reset[i] = (long long)o.done;
            k.timeout[i] = (uint8_t)o.timeout;
            if (lane == 0) k.mask[i >> 6] = ballot;
            if (k.cmd != nullptr) {
                k.cmd[i] = make_float4(o.cmd[0], o.cmd[1], o.cmd[2], o.cmd[3]);
#pragma unroll
                for (int t = 0; t < 9; ++t) k.terms[t][i] = o.terms[t];
            }
        }
    } else {
        float z[18];
        if (!P.noise_off) {
            obs_noise_normals(P, env_global, z);
        } else {
#pragma unroll
            for (int j = 0; j < 18; ++j) z[j] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 18; ++j) tileB[lane * SB + j] = noise_sigma(j) * z[j];
        if (lane < 18) tgt[lane] = (TASK == TASK_HOVERING) ? k.P.target[lane] : 0.0f;
    }
    __syncthreads();

    // obs = (clean + sigma*z) - target for the 18 noisy columns (hovering.py:343-345), clean elsewhere
    const int block_env0 = blockIdx.x * 64;
    const int valid = min(64, k.n - block_env0) * NOBS;
    float* out = k.obs + (size_t)block_env0 * NOBS;
    constexpr int NV4 = 64 * NOBS / 4;
    constexpr int ITERS = (NV4 + 127) / 128;
    auto elem = [&](int e) -> float {
        const int row = e / NOBS, col = e - row * NOBS;
        float v = tileA[row * SA + col];
        if (col < 18) v = (v + tileB[row * SB + col]) - tgt[col];
        return v;
    };
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int m = (int)threadIdx.x + it * 128;
        if (m >= NV4) break;
        const int e = 4 * m;
        if (e + 3 < valid) {
            reinterpret_cast<float4*>(out)[m] = make_float4(elem(e), elem(e + 1), elem(e + 2), elem(e + 3));
        } else {
            for (int q = e; q < e + 4 && q < valid; ++q) out[q] = elem(q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised kernel, second form (the default).  Same 2-wave split per 64 envs; differences to step_kernel_ws:
//   * the physics wave STORES THE STATE RIGHT AFTER THE RK4 (it is final unless the env terminates; a terminating lane
//     stores again after its reset), so the 112 B/env of state writes drain under the reward / observation arithmetic
//     instead of behind it;
//   * after the noise is ready (barrier 1) the physics wave forms (clean + sigma*z) - target in registers (target lives in
//     SGPRs) and writes FINAL rows into an LDS tile laid out exactly like the HBM rows, so the copy-out (after barrier 2)
//     is ds_read_b128 -> global_store_dwordx4 with no per-element index arithmetic (the first form spent ~450
//     instructions per wave on row/column div-mod and three LDS reads per element);
//   * rollout form: done flags as u8 and per-tile sums of the nine reward terms (summed by the noise wave, which has
//     slack) instead of nine f32[N] arrays + cmd -> 59 B/env less HBM traffic;
//   * FLIP: workgroups alternate which hardware wave plays which role so that a SIMD is not handed two physics waves.
// Arithmetic and its order are those of env_step(): results are bit-identical to step_kernel<...>.
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL, bool EARLY_STORE, bool FLIP>
__global__ __launch_bounds__(128) void step_kernel_ws2(const KArgs k) {
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int SB = 19;                       // odd stride: conflict-free row access of the noise tile
    constexpr int ST = 9;
    __shared__ __attribute__((aligned(16))) float tileO[64 * NOBS];   // final observation rows, HBM order
    __shared__ float tileB[64 * SB];
    __shared__ float tileT[64 * ST];

    if (k.stagger > 0) {
        // de-phasing experiment: the workgroups of a launch run in lockstep (all load, all compute, all store); every second
        // workgroup of a CU (by its LDS allocation slot) starts late so that its memory phases meet the others' arithmetic
        const unsigned base = __builtin_amdgcn_s_getreg((31 << 11) | 6) & 0xFF;      // HW_REG_LDS_ALLOC.LDS_BASE, 256-byte units
        if (((base + 4) / 46) & 1) {
#pragma unroll 1
            for (int q = 0; q < k.stagger; ++q) __builtin_amdgcn_s_sleep(16);
        }
    }
    const int hw_wave = threadIdx.x >> 6;
    const int wave = FLIP ? (hw_wave ^ ((blockIdx.x >> 1) & 1)) : hw_wave;
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
        if (active) {
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
        if (EARLY_STORE) {          // final for every lane that does not terminate this step
            store_env(k, i, s);
            store_ctl<CTL>(k, i, c);
            k.PA[i] = make_float4(a[0], a[1], a[2], a[3]);
            if (A == 5) k.PA4[i] = a[A - 1];
        }
        float obs[NOBS];
        env_observe_reward<TASK, CTL, false, true>(s, a, pre_a, o.cmd, P, env_global, nullptr, obs, o);
#pragma unroll
        for (int j = 0; j < A; ++j) pre_a[j] = a[j];
        s.was_reset = o.done;
        if (o.done) {
            env_reset_done<CTL, false>(s, c, pre_a, P, env_global, nullptr);
            if (EARLY_STORE) {
                store_env(k, i, s);
                store_ctl<CTL>(k, i, c);
                k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
                if (A == 5) k.PA4[i] = pre_a[A - 1];
            }
        }
        o.timeout = (s.progress > P.max_episode_length) ? 1 : 0;
        if (!EARLY_STORE) {
            store_env(k, i, s);
            store_ctl<CTL>(k, i, c);
            k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
            if (A == 5) k.PA4[i] = pre_a[A - 1];
        }
        const unsigned long long ballot = __ballot(active && o.done);
        if (active) {
            k.rew[i] = o.rew;
            if (k.reset_u8 != nullptr) k.reset_u8[i] = (uint8_t)o.done;
            else k.reset[i] = (long long)o.done;
            k.timeout[i] = (uint8_t)o.timeout;
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
        __syncthreads();   // barrier 1: sigma*z rows are in tileB
        // obs = (clean + sigma*z) - target for the 18 noisy columns (hovering.py:343-345), clean elsewhere
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            float v = obs[j] + tileB[lane * SB + j];
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
    } else {
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
    }
    __syncthreads();       // barrier 2: final rows are in tileO

    const int block_env0 = blockIdx.x * 64;
    const int valid = min(64, k.n - block_env0) * NOBS;    // floats of this tile that exist
    float* out = k.obs + (size_t)block_env0 * NOBS;
    constexpr int NV4 = 64 * NOBS / 4;
    constexpr int ITERS = (NV4 + 127) / 128;
    const int t2 = wave * 64 + lane;     // role-relative thread index (any bijection onto 0..127 works)
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
    if (want_terms && wave == 1) {     // per-tile sums of the reward terms, fixed order: 7 row groups x 9 terms, then 7 -> 1
        const int term = lane % 9, part = lane / 9;          // lanes 0..62
        const int r0 = (part == 0) ? 0 : 9 * part + 1, r1 = 9 * part + 10;   // row groups of 10,9,9,9,9,9,9
        float acc = 0.0f;
        if (lane < 63) {
            for (int r = r0; r < r1; ++r) acc += tileT[r * ST + term];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 63) tileB[lane] = acc;                    // tileB is free after barrier 2 (same wave wrote it)
        __builtin_amdgcn_wave_barrier();
        if (lane < 9) {
            float tot = 0.0f;
#pragma unroll
            for (int p = 0; p < 7; ++p) tot += tileB[p * 9 + lane];
            k.term_sums[(size_t)blockIdx.x * 12 + lane] = tot;
        }
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

template <int TASK, int CTL>
static hipError_t launch_step(const KArgs& k, int block, int obs_via_lds, hipStream_t stream) {
    const int n = k.n;
#define AG_LAUNCH(B, L)                                                                          \
    do {                                                                                         \
        const int grid = (n + (B)-1) / (B);                                                      \
        hipLaunchKernelGGL((step_kernel<TASK, CTL, B, L, false>), dim3(grid), dim3(B), 0, stream, k); \
        return hipGetLastError();                                                                \
    } while (0)
    if (k.ext_noise != nullptr) {  // parity mode: one geometry only
        hipLaunchKernelGGL((step_kernel<TASK, CTL, 64, true, true>), dim3((n + 63) / 64), dim3(64), 0, stream, k);
        return hipGetLastError();
    }
    const dim3 g64((n + 63) / 64);
    if (block == 0) {   // wave-specialised geometry, second form (default: state stored after the reward, measured faster)
        KArgs ks = k;
        ks.stagger = obs_via_lds > 1 ? obs_via_lds - 1 : 0;
        hipLaunchKernelGGL((step_kernel_ws2<TASK, CTL, false, false>), g64, dim3(128), 0, stream, ks);
        return hipGetLastError();
    }
    // A/B variants of the wave-specialised kernel (ag_set_launch_params block_size 1..4; tools/sweep_env_kernel.py)
    if (block == 2) { hipLaunchKernelGGL((step_kernel_ws2<TASK, CTL, true, false>), g64, dim3(128), 0, stream, k); return hipGetLastError(); }
    if (block == 3) { hipLaunchKernelGGL((step_kernel_ws2<TASK, CTL, true, true>), g64, dim3(128), 0, stream, k); return hipGetLastError(); }
    if (block == 4) { hipLaunchKernelGGL((step_kernel_ws2<TASK, CTL, false, true>), g64, dim3(128), 0, stream, k); return hipGetLastError(); }
    if (k.reset_u8 != nullptr || k.term_sums != nullptr) return hipErrorInvalidValue;   // rollout form: ws2 only
    if (block == 1) {   // first wave-specialised form, kept for A/B measurements
        hipLaunchKernelGGL((step_kernel_ws<TASK, CTL>), g64, dim3(128), 0, stream, k);
        return hipGetLastError();
    }
    if (block == 64) { if (obs_via_lds) AG_LAUNCH(64, true); else AG_LAUNCH(64, false); }
    if (block == 128) { if (obs_via_lds) AG_LAUNCH(128, true); else AG_LAUNCH(128, false); }
    if (block == 256) { if (obs_via_lds) AG_LAUNCH(256, true); else AG_LAUNCH(256, false); }
#undef AG_LAUNCH
    return hipErrorInvalidValue;
}

#define AG_ECAT_(a, b, c) launch_eval_##a##_##b
#define AG_ECAT(a, b) AG_ECAT_(a, b, 0)
hipError_t AG_ECAT(AG_TASK, AG_CTL)(const KArgs& k, hipStream_t stream) {
    hipLaunchKernelGGL((eval_obs_reward_kernel<AG_TASK, AG_CTL>), dim3((k.n + 63) / 64), dim3(64), 0, stream, k);
    return hipGetLastError();
}

#define AG_CAT_(a, b, c) launch_step_##a##_##b
#define AG_CAT(a, b) AG_CAT_(a, b, 0)
hipError_t AG_CAT(AG_TASK, AG_CTL)(const KArgs& k, int block, int obs_via_lds, hipStream_t stream) {
    return launch_step<AG_TASK, AG_CTL>(k, block, obs_via_lds, stream);
}

}  // namespace ag
