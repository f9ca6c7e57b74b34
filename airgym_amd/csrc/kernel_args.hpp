// kernel_args.hpp - device-side view of a handle's arena, shared by the step kernels
// (step_kernel.hip, one translation unit per task x ctl_mode) and the C-ABI (airgym_hip.hip).
//
// HBM layout (SoA, one env per lane, every array padded to a multiple of 256 envs):
//   S0 = (pos.xyz, progress as int bits)   S1 = quat xyzw
//   S2 = (linvel.xyz, was_reset as int bits)   S3 = (angvel.xyz, 0)
//   C0 = rate integrator, C1 = previous body rate, C2 = velocity integrator, C3 = previous velocity
//   PA = previous processed action (xyzw), PA4 = its 5th component (atti mode only)
// Each is a float4 array: one 16-byte load per lane, 1 KiB per wave-instruction, fully coalesced.
// This replaces the reference's AoS root tensor view [N, actors, 13] (hovering.py:70-77).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env_math.hpp"

namespace ag {

struct KArgs {
    float4* S[4];
    float4* C[4];
    float4* PA;
    float* PA4;
    const float* actions;        // [n, A]
    float* obs;                  // [n, NOBS]
    float* rew;                  // [n]
    long long* reset;            // [n] int64
    uint8_t* timeout;            // [n]
    unsigned long long* mask;    // [ceil(n/64)]
    float* terms[9];             // each [n] or null
    float4* cmd;                 // [n] or null
    const float* ext_noise;      // [n,18] or null
    const float* ext_uniforms;   // [n,12] or null
    // rollout form (ag_step_rollout): done flags as u8 (the width the rollout buffer stores, experience.py:329) and
    // per-64-env-tile sums of the reward terms instead of nine per-env arrays (Episode/<term> logging only needs means)
    uint8_t* reset_u8;           // [n] or null; when set, `reset` (int64) is not written
    // K env steps per launch (ag_step_multi): actions / obs / rew / reset_u8 / term_sums / timeout_steps are [K, ...] arrays,
    // step kk at offset kk * n * width; the state is loaded before step 0 and stored after step K - 1
    int num_steps;               // >= 1
    int force_multi;             // ag_step_multi: always the K-step kernel (num_steps == 1 otherwise takes the one-step kernel)
    uint8_t* timeout_steps;      // [K, n] or null: per-step time-out flags (`timeout` [n] keeps the LAST step's)
    float* term_sums;            // [ceil(n/64), 12] or null: sums over the tile's envs of terms[0..8]
    // parity / inspection mode (ag_eval_obs_reward): processed actions + controller output supplied by the caller
    const float* eval_actions;   // [n, A]
    const float* eval_cmd;       // [n, 4]
    // RNG tick lives in device memory so that a captured hipGraph of env steps replays with fresh
    // counters: each launch reads tick_in and thread 0 publishes tick+1 to tick_out (the two slots
    // alternate launch to launch; stream order between launches makes the hand-off race-free).
    const uint32_t* tick_in;
    uint32_t* tick_out;
    int n;
    StepParams P;
};

// Rollout head + tail fused into the env-step launch (ag_step_rollout_fused; step_kernel_ws2<.., true>): what
// ag_policy_sample reads / writes in front of the step and what ag_rollout_account reads / writes behind it
// (lib/agent/a2c_base.py:651-695).
struct TailArgs {
    const float* heads;          // [n, A+1] mu | normalised value
    const float* logstd;         // [A]
    const double* vmean;         // [1] or null (value de-normalisation, base_model.py:29-35)
    const double* vvar;
    float veps;
    uint32_t key0, key1;         // Philox key of the action noise
    const long long* counter;    // [1] rollout counter (device): tick = counter * horizon + slot
    int horizon, slot;
    long long id_offset;         // global id of local env 0 for the action noise
    float* actions; float* neglogp; float* values; float* mus; float* sigmas;   // rollout slot t
    float scale, shift, min_val, max_val; int log_val; float gamma; int bootstrap;   // reward shaper, time-out bootstrap
    float* shaped;               // [n] shaped reward of slot t
    float* cur_rew; float* cur_shaped; float* cur_len;   // [n] running episode sums (read-modify-write)
    double* partials;            // [ceil(n/64), 4] per-tile {episodes ended, sum reward, sum shaped, sum length}
};

// Planning-task extras (airgym_amd/csrc/planning_kernel.hip)
struct PlanArgs {
    float4* OB;                  // planning: [40][n_pad] obstacle root pose (x, y, yaw, variant index as int bits);
                                 // avoid: [1][n_pad] velocity of the thrown cube
    float4* GOAL;                // [n_pad] planning: (goal xyz, prev_related_dist); balloon: balloon xyz; avoid: cube xyz
    float4* PRP;                 // [n_pad] (pre_root_positions xyz, esdf_dist = min pixel of the last image)
    float* image;                // [n, 212*120]  == full_camera_array [n, 1, 212, 120]
    float* collisions;           // [n]
    float* terms[11];            // item_reward_info arrays or null (planning 11, avoid 8, balloon 6)
    const float* table;          // [100, 8] obstacle variants (centre3, axis3, radius, half length)
    const float* ext_uniforms;   // [n, 121] or null (parity mode)
    const float* ext_collisions; // [n] or null (parity mode: collision flags supplied instead of the geometric test)
    int n_pad;
    int debug_skip;              // diagnostics only: bit0 skip ray-cast, bit1 skip noise passes, bit2 skip the 5x5 pass
};

hipError_t launch_planning_step(const KArgs& k, const PlanArgs& pa, int ctl, int phase, hipStream_t st);
hipError_t launch_planning_render(const KArgs& k, const PlanArgs& pa, hipStream_t st);
hipError_t launch_planning_reset_all(const KArgs& k, const PlanArgs& pa, int num_actions, hipStream_t st);
// Balloon / Avoid (task = 3 / 4): same phase convention as Planning; Avoid renders with launch_avoid_render
hipError_t launch_custom_step(const KArgs& k, const PlanArgs& pa, int task, int ctl, int phase, hipStream_t st);
hipError_t launch_custom_reset_all(const KArgs& k, const PlanArgs& pa, int task, int num_actions, hipStream_t st);
hipError_t launch_avoid_render(const KArgs& k, const PlanArgs& pa, hipStream_t st);
hipError_t launch_custom_reset_ids(const KArgs& k, const PlanArgs& pa, int task, int num_actions, const int* ids, int count,
                                   hipStream_t st);

typedef hipError_t (*StepLauncher)(const KArgs& k, const TailArgs* tail, hipStream_t stream);
typedef hipError_t (*EvalLauncher)(const KArgs& k, hipStream_t stream);

// defined in step_kernel.hip compiled with -DAG_TASK=<t> -DAG_CTL=<c>
#define AG_DECL_LAUNCHER(t, c) hipError_t launch_step_##t##_##c(const KArgs& k, const TailArgs* tail, hipStream_t stream);
AG_DECL_LAUNCHER(0, 0) AG_DECL_LAUNCHER(0, 1) AG_DECL_LAUNCHER(0, 2) AG_DECL_LAUNCHER(0, 3) AG_DECL_LAUNCHER(0, 4)
AG_DECL_LAUNCHER(1, 0) AG_DECL_LAUNCHER(1, 1) AG_DECL_LAUNCHER(1, 2) AG_DECL_LAUNCHER(1, 3) AG_DECL_LAUNCHER(1, 4)
#undef AG_DECL_LAUNCHER
#define AG_DECL_EVAL(t, c) hipError_t launch_eval_##t##_##c(const KArgs& k, hipStream_t stream);
AG_DECL_EVAL(0, 0) AG_DECL_EVAL(0, 1) AG_DECL_EVAL(0, 2) AG_DECL_EVAL(0, 3) AG_DECL_EVAL(0, 4)
AG_DECL_EVAL(1, 0) AG_DECL_EVAL(1, 1) AG_DECL_EVAL(1, 2) AG_DECL_EVAL(1, 3) AG_DECL_EVAL(1, 4)
#undef AG_DECL_EVAL

__device__ __forceinline__ void load_env(const KArgs& k, int i, EnvState& s) {
    const float4 a = k.S[0][i], b = k.S[1][i], c = k.S[2][i], d = k.S[3][i];
    s.p = V3{a.x, a.y, a.z};
    s.progress = __float_as_int(a.w);
    s.q = Q4{b.x, b.y, b.z, b.w};
    s.v = V3{c.x, c.y, c.z};
    s.was_reset = __float_as_int(c.w);
    s.w = V3{d.x, d.y, d.z};
}

__device__ __forceinline__ void store_env(const KArgs& k, int i, const EnvState& s) {
    k.S[0][i] = make_float4(s.p.x, s.p.y, s.p.z, __int_as_float(s.progress));
    k.S[1][i] = make_float4(s.q.x, s.q.y, s.q.z, s.q.w);
    k.S[2][i] = make_float4(s.v.x, s.v.y, s.v.z, __int_as_float(s.was_reset));
    k.S[3][i] = make_float4(s.w.x, s.w.y, s.w.z, 0.0f);
}

// controller memory each mode actually touches: rate/atti use C0,C1; vel/pos use C0..C3; prop none
template <int CTL>
__device__ __forceinline__ void load_ctl(const KArgs& k, int i, CtlState& c) {
    if (CTL != CTL_PROP) {
        const float4 a = k.C[0][i], b = k.C[1][i];
        c.rate_int[0] = a.x; c.rate_int[1] = a.y; c.rate_int[2] = a.z;
        c.prev_rate[0] = b.x; c.prev_rate[1] = b.y; c.prev_rate[2] = b.z;
    }
    if (CTL == CTL_VEL || CTL == CTL_POS) {
        const float4 a = k.C[2][i], b = k.C[3][i];
        c.vel_int[0] = a.x; c.vel_int[1] = a.y; c.vel_int[2] = a.z;
        c.prev_vel[0] = b.x; c.prev_vel[1] = b.y; c.prev_vel[2] = b.z;
    }
}

template <int CTL>
__device__ __forceinline__ void store_ctl(const KArgs& k, int i, const CtlState& c) {
    if (CTL != CTL_PROP) {
        k.C[0][i] = make_float4(c.rate_int[0], c.rate_int[1], c.rate_int[2], 0.0f);
        k.C[1][i] = make_float4(c.prev_rate[0], c.prev_rate[1], c.prev_rate[2], 0.0f);
    }
    if (CTL == CTL_VEL || CTL == CTL_POS) {
        k.C[2][i] = make_float4(c.vel_int[0], c.vel_int[1], c.vel_int[2], 0.0f);
        k.C[3][i] = make_float4(c.prev_vel[0], c.prev_vel[1], c.prev_vel[2], 0.0f);
    }
}

}  // namespace ag
