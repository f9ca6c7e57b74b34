// env_math.hpp - per-environment arithmetic of one AirGym env step, float32.
//
// One call of env_step<TASK, CTL>() advances ONE environment (one GPU lane):
//   action map -> control cascade -> rotor wrench -> RK4 -> progress++ -> obs -> reward/done -> reset
// It is pure register arithmetic; all memory traffic lives in step_kernel.hip.
//
// Every function cites the reference code it replaces (emNavi/AirGym, paths
// relative to the reference root) or, for the two pieces that are external
// binaries in the reference (PhysX integrator, rlPx4Controller cascades), the
// build's written spec in oracle/rigid_body.py and oracle/px4_cascade.py.  The
// evaluation order of every expression is the oracle's, so that the only
// differences are FMA contraction and libm-vs-OCML ulps.
//
// The functions are `__host__ __device__` so that tests/host_harness can run the
// SAME source under g++ on the GPU-less build box (a test aid: the product
// library airgym_amd/csrc/airgym_hip.hip has no CPU execution path).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AG_HD __host__ __device__ __forceinline__
#else
#define AG_HD inline
#endif

namespace ag {

enum : int { TASK_HOVERING = 0, TASK_TRACKING = 1 };
enum : int { CTL_POS = 0, CTL_VEL = 1, CTL_ATTI = 2, CTL_RATE = 3, CTL_PROP = 4 };

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };  // xyzw, body -> world (IsaacGym root-state order, hovering.py:75)

// ---------------------------------------------------------------------------
// Vehicle + force model constants.
// X152b composite body from airgym/assets/robots/X152b/model.urdf:19-24,36-39,86-105;
// thrust / reaction-torque scales hovering.py:256,270; speed caps airgym/assets/__init__.py:30-35.
// ---------------------------------------------------------------------------
constexpr double kMassD = 0.601;
constexpr double kIxxD = 0.0400591785, kIyyD = 0.0400591785, kIzzD = 0.0400964156;
constexpr float kInvMass = (float)(1.0 / kMassD);
constexpr float kIxx = (float)kIxxD, kIyy = (float)kIyyD, kIzz = (float)kIzzD;
constexpr float kInvIxx = (float)(1.0 / kIxxD), kInvIyy = (float)(1.0 / kIyyD), kInvIzz = (float)(1.0 / kIzzD);
constexpr float kGravityZ = -9.81f;
constexpr float kMaxLinVel = 100.0f, kMaxAngVel = 100.0f;
constexpr float kThrustPerCmd = 9.59f;
constexpr float kRotorArm = 0.05374f;
constexpr float kYawTorquePerCmd = 0.2f;

// Control cascade gains: oracle/px4_cascade.py (PX4 multicopter defaults; build's own spec).
constexpr float kCtlDt = 0.01f, kInvCtlDt = 100.0f;
constexpr float kRateKp[3] = {0.15f, 0.15f, 0.2f};
constexpr float kRateKi[3] = {0.2f, 0.2f, 0.1f};
constexpr float kRateKd[3] = {0.003f, 0.003f, 0.0f};
constexpr float kRateIntLim = 0.3f;
constexpr float kRateIAttenInv = (float)(1.0 / (400.0 * 3.14159265358979323846 / 180.0));
constexpr float kMixRP = 0.70710678f, kMixYaw = 1.0f;
constexpr float kAttGain[3] = {6.5f, 6.5f, (float)(2.8 / 0.4)};
constexpr float kAttYawW = 0.4f;
constexpr float kAttRateLim[3] = {(float)(220.0 * 3.14159265358979323846 / 180.0),
                                  (float)(220.0 * 3.14159265358979323846 / 180.0),
                                  (float)(200.0 * 3.14159265358979323846 / 180.0)};
constexpr float kVelKp[3] = {1.8f, 1.8f, 4.0f};
constexpr float kVelKi[3] = {0.4f, 0.4f, 2.0f};
constexpr float kVelKd[3] = {0.2f, 0.2f, 0.0f};
constexpr float kVelIntLim = 9.81f;
constexpr float kGrav = 9.81f;
constexpr float kHoverOverG = (float)(0.1537 / 9.81);
constexpr float kThrMin = 0.03f, kThrMax = 1.0f;
constexpr float kCosTiltMax = 0.70710678118654757f;  // cos(45 deg)
constexpr float kSinTiltMax = 0.70710678118654746f;  // sin(45 deg)
constexpr float kPosKp[3] = {0.95f, 0.95f, 1.0f};
constexpr float kPosVelXYMax = 12.0f, kPosVelUpMax = 3.0f, kPosVelDnMax = 1.5f;

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.283185307179586f;
constexpr float kInv2p24 = 1.0f / 16777216.0f;

// observation noise sigmas, hovering.py:350-353
constexpr float kSigMat = 1e-3f, kSigPos = 5e-3f, kSigVel = 2e-2f, kSigAng = 4e-1f;

// ---------------------------------------------------------------------------
// Uniform (per launch) parameters.
// ---------------------------------------------------------------------------
struct StepParams {
    float dt, half_dt, dt_over_6;
    int max_episode_length;
    float target[18];      // cfg.env.target_state, hovering_config.py:12
    float target_yaw;      // matrix_to_euler_angles(target, 'XYZ')[2] = atan2(-T01, T00), hovering.py:401
    uint32_t key0, key1;   // Philox key = seed
    uint32_t tick;         // env-step index of the handle
    uint32_t env_id_offset;
    uint32_t noise_off;
    uint32_t fix_time_outs; // AG_FLAG_FIX_TIME_OUTS (opt-in): see step_timeout()
    uint32_t stagger_phase; // AG_FLAG_STAGGER_PHASE (opt-in): see stagger_progress()
    // reset distribution (hovering.py:316-329 / tracking.py:166-179)
    float reset_pos_scale[3], reset_pos_offset[3], reset_euler_scale[3];
    float reset_linvel_scale, reset_angvel_scale;
};

// Host-side construction of the per-launch parameter block (reset distributions:
// hovering.py:316-329 / tracking.py:166-179).  The oracle multiplies f32 tensors by python
// doubles, i.e. the scalar is rounded to f32 once: dt, dt/2 and dt/6 are formed in double first.
inline StepParams make_step_params(int task, double dt, int max_episode_length, const float* target18, uint64_t seed,
                                   uint32_t env_id_offset, bool noise_off, bool fix_time_outs = false,
                                   bool stagger_phase = false) {
    StepParams P;
    P.dt = (float)dt;
    P.half_dt = (float)(0.5 * dt);
    P.dt_over_6 = (float)(dt / 6.0);
    P.max_episode_length = max_episode_length;
    for (int i = 0; i < 18; ++i) P.target[i] = target18[i];
    P.target_yaw = atan2f(-P.target[1], P.target[0]);
    P.key0 = (uint32_t)(seed & 0xFFFFFFFFull);
    P.key1 = (uint32_t)(seed >> 32);
    P.tick = 0;
    P.env_id_offset = env_id_offset;
    P.noise_off = noise_off ? 1u : 0u;
    P.fix_time_outs = fix_time_outs ? 1u : 0u;
    P.stagger_phase = stagger_phase ? 1u : 0u;
    if (task == 1) {
        P.reset_pos_scale[0] = P.reset_pos_scale[1] = P.reset_pos_scale[2] = 0.1f;
        P.reset_pos_offset[0] = P.reset_pos_offset[1] = 0.0f;
        P.reset_pos_offset[2] = 1.0f;
        P.reset_euler_scale[0] = P.reset_euler_scale[1] = 0.1f;
        P.reset_euler_scale[2] = 0.2f;
    } else {
        P.reset_pos_scale[0] = P.reset_pos_scale[1] = P.reset_pos_scale[2] = 1.0f;
        P.reset_pos_offset[0] = P.reset_pos_offset[1] = P.reset_pos_offset[2] = 0.0f;
        P.reset_euler_scale[0] = P.reset_euler_scale[1] = 0.01f;
        P.reset_euler_scale[2] = 0.05f;
    }
    P.reset_linvel_scale = 0.5f;
    P.reset_angvel_scale = 0.2f;
    return P;
}

struct EnvState {
    V3 p;
    Q4 q;
    V3 v;   // world
    V3 w;   // world (IsaacGym convention)
    int progress;
    int was_reset;  // reset_buf value left by the previous step (hovering.py:209,268)
};

struct CtlState {
    float rate_int[3], prev_rate[3], vel_int[3], prev_vel[3];
};

template <int TASK>
struct TaskTraits;
template <>
struct TaskTraits<TASK_HOVERING> { static constexpr int kNumObs = 18; };
template <>
struct TaskTraits<TASK_TRACKING> { static constexpr int kNumObs = 48; };

template <int CTL>
struct CtlTraits { static constexpr int kNumActions = (CTL == CTL_ATTI) ? 5 : 4; };

// Action limits, hovering.py:93-121 (Tracking: pos limits +-6, tracking.py:95-99).
template <int TASK, int CTL>
AG_HD void action_limits(float* lo, float* hi) {
    if (CTL == CTL_POS) {
        const float l = (TASK == TASK_TRACKING) ? 6.0f : 3.0f;
        lo[0] = -l; lo[1] = -l; lo[2] = -l; lo[3] = -6.0f;
        hi[0] = l; hi[1] = l; hi[2] = l; hi[3] = 6.0f;
    } else if (CTL == CTL_VEL) {
        for (int i = 0; i < 4; ++i) { lo[i] = -6.0f; hi[i] = 6.0f; }
    } else if (CTL == CTL_ATTI) {
        for (int i = 0; i < 4; ++i) { lo[i] = -1.0f; hi[i] = 1.0f; }
        lo[4] = 0.0f; hi[4] = 1.0f;
    } else if (CTL == CTL_RATE) {
        for (int i = 0; i < 3; ++i) { lo[i] = -6.0f; hi[i] = 6.0f; }
        lo[3] = 0.0f; hi[3] = 1.0f;
    } else {
        for (int i = 0; i < 4; ++i) { lo[i] = 0.0f; hi[i] = 1.0f; }
    }
}

AG_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---------------------------------------------------------------------------
// Transcendental units.  On gfx950 these map to single quarter-rate VALU instructions
// (v_rcp/v_rsq/v_sqrt/v_log/v_exp/v_sin/v_cos, <= 1 ulp; v_sin/v_cos take REVOLUTIONS, so
// sin(2*pi*u) needs no multiply and no range reduction).  OCML's correctly-rounded sinf/cosf carry
// a Payne-Hanek path that made up ~40 % of the kernel's instructions.  The host build (test harness)
// uses libm; both stay inside the 1e-5 parity budget.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
AG_HD float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
AG_HD float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
AG_HD float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
AG_HD float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }
AG_HD float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504089f); }
AG_HD float sin_2pi(float rev) { return __builtin_amdgcn_sinf(rev); }
AG_HD float cos_2pi(float rev) { return __builtin_amdgcn_cosf(rev); }
#else
AG_HD float fast_rcp(float x) { return 1.0f / x; }
AG_HD float fast_rsq(float x) { return 1.0f / sqrtf(x); }
AG_HD float fast_sqrt(float x) { return sqrtf(x); }
AG_HD float fast_ln(float x) { return logf(x); }
AG_HD float fast_exp(float x) { return expf(x); }
AG_HD float sin_2pi(float rev) { return sinf(6.283185307179586f * rev); }
AG_HD float cos_2pi(float rev) { return cosf(6.283185307179586f * rev); }
#endif

// ---------------------------------------------------------------------------
// Philox4x32-10 (oracle/philox.py; Salmon et al. SC'11)
// ---------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

AG_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

AG_HD float u32_to_unit(uint32_t x) { return (float)(x >> 8) * kInv2p24; }
AG_HD float u32_to_open_unit(uint32_t x) { return ((float)(x >> 8) + 1.0f) * kInv2p24; }

enum : uint32_t { STREAM_RESET = 0, STREAM_OBS_NOISE = 1, STREAM_PHASE = 2 };

// AG_FLAG_STAGGER_PHASE: the progress a FULL reset gives env `env_global`, uniform on {0 .. max_episode_length - 2}
// (oracle/hovering_ref.py HoveringRef._reset_all)
AG_HD int stagger_progress(const StepParams& P, uint32_t env_global) {
    const U4 r = philox4x32_10(env_global, P.tick, STREAM_PHASE, 0u, P.key0, P.key1);
    const uint32_t span = (uint32_t)(P.max_episode_length > 1 ? P.max_episode_length - 1 : 1);
    return (int)(r.x % span);
}

// 12 U[0,1): pos3 euler3 linvel3 angvel3
AG_HD void reset_uniforms(const StepParams& P, uint32_t env_global, float* u) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_RESET, (uint32_t)b, P.key0, P.key1);
        u[4 * b + 0] = u32_to_unit(r.x);
        u[4 * b + 1] = u32_to_unit(r.y);
        u[4 * b + 2] = u32_to_unit(r.z);
        u[4 * b + 3] = u32_to_unit(r.w);
    }
}

AG_HD void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = u32_to_open_unit(a);
    const float u2 = u32_to_unit(b);
    const float r = fast_sqrt(-2.0f * fast_ln(u1));
    z0 = r * cos_2pi(u2);
    z1 = r * sin_2pi(u2);
}

// 18 N(0,1) for add_noise (hovering.py:349-358)
AG_HD void obs_noise_normals(const StepParams& P, uint32_t env_global, float* z) {
    uint32_t raw[20];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_OBS_NOISE, (uint32_t)b, P.key0, P.key1);
        raw[4 * b + 0] = r.x; raw[4 * b + 1] = r.y; raw[4 * b + 2] = r.z; raw[4 * b + 3] = r.w;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) box_muller(raw[2 * k], raw[2 * k + 1], z[2 * k], z[2 * k + 1]);
}

// ---------------------------------------------------------------------------
// quaternion / vector helpers (xyzw)
// ---------------------------------------------------------------------------
AG_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

AG_HD V3 quat_rotate(Q4 q, V3 v) {  // oracle/rigid_body.py quat_rotate_xyzw
    const V3 qv{q.x, q.y, q.z};
    V3 t = cross(qv, v);
    t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
    const V3 c = cross(qv, t);
    return V3{v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

AG_HD V3 quat_rotate_inverse(Q4 q, V3 v) {
    const V3 qv{q.x, q.y, q.z};
    V3 t = cross(qv, v);
    t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
    const V3 c = cross(qv, t);
    return V3{v.x - q.w * t.x + c.x, v.y - q.w * t.y + c.y, v.z - q.w * t.z + c.z};
}

AG_HD Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
AG_HD Q4 qconj(Q4 a) { return Q4{-a.x, -a.y, -a.z, a.w}; }
AG_HD V3 q_body_z(Q4 q) {
    return V3{2.0f * (q.x * q.z + q.w * q.y), 2.0f * (q.y * q.z - q.w * q.x), 1.0f - 2.0f * (q.x * q.x + q.y * q.y)};
}

// ---------------------------------------------------------------------------
// Control cascade (oracle/px4_cascade.py; replaces rlPx4Controller, hovering.py:235-250)
// ---------------------------------------------------------------------------
AG_HD void ctl_reset(CtlState& c, const EnvState& s) {
    const V3 wb = quat_rotate_inverse(s.q, s.w);
    c.rate_int[0] = c.rate_int[1] = c.rate_int[2] = 0.0f;
    c.vel_int[0] = c.vel_int[1] = c.vel_int[2] = 0.0f;
    c.prev_rate[0] = wb.x; c.prev_rate[1] = wb.y; c.prev_rate[2] = wb.z;
    c.prev_vel[0] = s.v.x; c.prev_vel[1] = s.v.y; c.prev_vel[2] = s.v.z;
}

AG_HD void rate_control(CtlState& c, const float* rate_sp, const float* wb, float* u) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float err = rate_sp[i] - wb[i];
        const float wdot = (wb[i] - c.prev_rate[i]) * kInvCtlDt;
        u[i] = kRateKp[i] * err + c.rate_int[i] - kRateKd[i] * wdot;
        const float a = err * kRateIAttenInv;
        const float i_factor = fmaxf(1.0f - a * a, 0.0f);
        const float ri = c.rate_int[i] + i_factor * kRateKi[i] * err * kCtlDt;
        c.rate_int[i] = clampf(ri, -kRateIntLim, kRateIntLim);
        c.prev_rate[i] = wb[i];
    }
}

// Quad-X control allocation with PX4's sequential desaturation, airmode disabled (PX4 v1.14
// ControlAllocationSequentialDesaturation::mixAirmodeDisabled / mixYaw / desaturateActuators; oracle/px4_cascade.py
// mix_quad_x): roll+pitch+thrust are mixed first; a saturated output is relieved by REDUCING thrust only, then by scaling
// back the roll, then the pitch demand; yaw is mixed last against limits widened by 15 % and gives way first; a final
// thrust-reduce pass, then the [0,1] clip.  (A clip-only mixer turns every saturated torque demand into extra collective
// thrust; the reference's own trained Planning policy cannot hold altitude on such a vehicle - DESIGN.md section 2.)
// gain k such that o + k * vec relieves the worst violation on either side: k_i = (clamp(o_i) - o_i) / vec_i is zero for
// an output inside [lo, hi], so min / max over all four equal PX4's loop over the violating ones (computeDesaturationGain).
// `inv` holds 1 / vec_i (the mixer columns are +-0.70710678 or +-1: reciprocals are compile-time constants).
AG_HD float desat_gain(const float* o, const float* inv, float lo, float hi) {
    const float k0 = (clampf(o[0], lo, hi) - o[0]) * inv[0];
    const float k1 = (clampf(o[1], lo, hi) - o[1]) * inv[1];
    const float k2 = (clampf(o[2], lo, hi) - o[2]) * inv[2];
    const float k3 = (clampf(o[3], lo, hi) - o[3]) * inv[3];
    const float kmin = fminf(fminf(fminf(k0, k1), fminf(k2, k3)), 0.0f);
    const float kmax = fmaxf(fmaxf(fmaxf(k0, k1), fmaxf(k2, k3)), 0.0f);
    return kmin + kmax;
}

AG_HD void desaturate(float* o, const float* vec, const float* inv, float lo, float hi, bool reduce_only) {
    float k1 = desat_gain(o, inv, lo, hi);
    const bool active = !(reduce_only && k1 > 0.0f);
    k1 = active ? k1 : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(k1, vec[i], o[i]);
    float k2 = 0.5f * desat_gain(o, inv, lo, hi);
    k2 = active ? k2 : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(k2, vec[i], o[i]);
}

AG_HD void mix_quad_x(float thrust, const float* u, float* cmd) {
    // rotor torque-sign pattern (roll, pitch, yaw): 1:(-,-,-) 2:(+,+,-) 3:(+,-,+) 4:(-,+,+)
    constexpr float kInvRP = 1.41421356f, kInvYaw = 1.0f;       // 1 / kMixRP, 1 / kMixYaw
    const float rv[4] = {-kMixRP, kMixRP, kMixRP, -kMixRP}, ri[4] = {-kInvRP, kInvRP, kInvRP, -kInvRP};
    const float pv[4] = {-kMixRP, kMixRP, -kMixRP, kMixRP}, pi[4] = {-kInvRP, kInvRP, -kInvRP, kInvRP};
    const float yv[4] = {-kMixYaw, -kMixYaw, kMixYaw, kMixYaw}, yi[4] = {-kInvYaw, -kInvYaw, kInvYaw, kInvYaw};
    const float tv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = thrust + rv[i] * u[0] + pv[i] * u[1];
    desaturate(o, tv, tv, 0.0f, 1.0f, true);
    desaturate(o, rv, ri, 0.0f, 1.0f, false);
    desaturate(o, pv, pi, 0.0f, 1.0f, false);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(yv[i], u[2], o[i]);
    desaturate(o, yv, yi, 0.0f, 1.15f, false);
    desaturate(o, tv, tv, 0.0f, 1.0f, true);
#pragma unroll
    for (int i = 0; i < 4; ++i) cmd[i] = clampf(o[i], 0.0f, 1.0f);
}

AG_HD void attitude_control(Q4 q, Q4 qd, float* rate_sp) {
    const float n2 = qd.x * qd.x + qd.y * qd.y + qd.z * qd.z + qd.w * qd.w;
    const bool bad = n2 < 1e-12f;
    const float inv = fast_rsq(bad ? 1.0f : n2);
    qd = Q4{bad ? 0.0f : qd.x * inv, bad ? 0.0f : qd.y * inv, bad ? 0.0f : qd.z * inv, bad ? 1.0f : qd.w * inv};
    const V3 ez = q_body_z(q);
    const V3 ezd = q_body_z(qd);
    const float cx = ez.y * ezd.z - ez.z * ezd.y;
    const float cy = ez.z * ezd.x - ez.x * ezd.z;
    const float cz = ez.x * ezd.y - ez.y * ezd.x;
    const float dot = ez.x * ezd.x + ez.y * ezd.y + ez.z * ezd.z;
    const float rw = dot + 1.0f;
    const bool singular = rw < 1e-5f;
    const float rn = fast_rsq(singular ? 1.0f : cx * cx + cy * cy + cz * cz + rw * rw);
    Q4 red = qmul(Q4{cx * rn, cy * rn, cz * rn, rw * rn}, q);
    if (singular) red = qd;
    const Q4 qmix = qmul(qconj(red), qd);
    const float sgn = qmix.w < 0.0f ? -1.0f : 1.0f;
    const float mw = clampf(qmix.w * sgn, -1.0f, 1.0f);
    const float mz = clampf(qmix.z * sgn, -1.0f, 1.0f);
    const Q4 yaw_q{0.0f, 0.0f, sinf(kAttYawW * asinf(mz)), cosf(kAttYawW * acosf(mw))};
    const Q4 qdd = qmul(red, yaw_q);
    const Q4 qe = qmul(qconj(q), qdd);
    const float s2 = qe.w < 0.0f ? -2.0f : 2.0f;
    const float e[3] = {qe.x, qe.y, qe.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) rate_sp[i] = clampf(kAttGain[i] * (s2 * e[i]), -kAttRateLim[i], kAttRateLim[i]);
}

AG_HD void velocity_control(CtlState& c, const float* vel_sp, const float* vel, float yaw_sp, Q4& q_sp, float& coll) {
    float err[3], acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        err[i] = vel_sp[i] - vel[i];
        const float vdot = (vel[i] - c.prev_vel[i]) * kInvCtlDt;
        acc[i] = kVelKp[i] * err[i] + c.vel_int[i] - kVelKd[i] * vdot;
    }
    const float bn = fast_rsq(acc[0] * acc[0] + acc[1] * acc[1] + kGrav * kGrav);
    float bx = acc[0] * bn, by = acc[1] * bn, bz = kGrav * bn;
    const bool over = bz < kCosTiltMax;
    const float hn = sqrtf(bx * bx + by * by);
    const float hs = kSinTiltMax / (over ? hn : 1.0f);
    bx = over ? bx * hs : bx;
    by = over ? by * hs : by;
    bz = over ? kCosTiltMax : bz;
    const float coll_raw = (acc[2] + kGrav) * kHoverOverG / bz;
    coll = clampf(coll_raw, kThrMin, kThrMax);
    const bool sat = ((coll_raw >= kThrMax) && (err[2] >= 0.0f)) || ((coll_raw <= kThrMin) && (err[2] <= 0.0f));
    err[2] = sat ? 0.0f : err[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float vi = c.vel_int[i] + kVelKi[i] * err[i] * kCtlDt;
        c.vel_int[i] = clampf(vi, -kVelIntLim, kVelIntLim);
        c.prev_vel[i] = vel[i];
    }
    const float tw = 1.0f + bz;
    const float tn = fast_rsq(bx * bx + by * by + tw * tw);
    const Q4 q_tilt{-by * tn, bx * tn, 0.0f, tw * tn};
    const float half = 0.5f * yaw_sp;
    const Q4 q_yaw{0.0f, 0.0f, sinf(half), cosf(half)};
    q_sp = qmul(q_tilt, q_yaw);
}

AG_HD void position_control(const float* pos_sp, const float* pos, float* vel_sp) {
    float vx = kPosKp[0] * (pos_sp[0] - pos[0]);
    float vy = kPosKp[1] * (pos_sp[1] - pos[1]);
    float vz = kPosKp[2] * (pos_sp[2] - pos[2]);
    const float n = sqrtf(vx * vx + vy * vy);
    const bool over = n > kPosVelXYMax;
    const float s = kPosVelXYMax / (over ? n : 1.0f);
    vel_sp[0] = over ? vx * s : vx;
    vel_sp[1] = over ? vy * s : vy;
    vel_sp[2] = clampf(vz, -kPosVelDnMax, kPosVelUpMax);
}

// hovering.py:234-254 dispatch.  a[] = pre-processed, clamped action.
template <int CTL>
AG_HD void controller_update(CtlState& c, const EnvState& s, const float* a, float* cmd) {
    if (CTL == CTL_PROP) {
        cmd[0] = a[0]; cmd[1] = a[1]; cmd[2] = a[2]; cmd[3] = a[3];
        return;
    }
#if defined(AG_EXP_CASCADE) && (AG_EXP_CASCADE & 1)      /* identification probe (tools/cascade_sweep.py): world-frame omega */
    const V3 wbv = s.w;
#else
    const V3 wbv = quat_rotate_inverse(s.q, s.w);
#endif
    const float wb[3] = {wbv.x, wbv.y, wbv.z};
    float rate_sp[3];
    float thrust;
    if (CTL == CTL_RATE) {
        rate_sp[0] = a[0]; rate_sp[1] = a[1]; rate_sp[2] = a[2];
        thrust = a[3];
    } else if (CTL == CTL_ATTI) {
        attitude_control(s.q, Q4{a[1], a[2], a[3], a[0]}, rate_sp);  // action = (qw,qx,qy,qz,thrust), hovering.py:105
        thrust = a[4];
    } else {
        const float vel[3] = {s.v.x, s.v.y, s.v.z};
        float vel_sp[3];
        if (CTL == CTL_POS) {
            const float pos[3] = {s.p.x, s.p.y, s.p.z};
            position_control(a, pos, vel_sp);
        } else {
            vel_sp[0] = a[0]; vel_sp[1] = a[1]; vel_sp[2] = a[2];
        }
        Q4 q_sp;
        velocity_control(c, vel_sp, vel, a[3], q_sp, thrust);
        attitude_control(s.q, q_sp, rate_sp);
    }
    float u[3];
    rate_control(c, rate_sp, wb, u);
    mix_quad_x(thrust, u, cmd);
}

// ---------------------------------------------------------------------------
// Wrench assembly (hovering.py:256-277 reduced to the composite body; oracle/rigid_body.py)
// ---------------------------------------------------------------------------
AG_HD void body_wrench_from_cmd(const float* cmd, float thrust_mask, float& fz, V3& tau) {
    const float t0 = cmd[0] * kThrustPerCmd * thrust_mask;
    const float t1 = cmd[1] * kThrustPerCmd * thrust_mask;
    const float t2 = cmd[2] * kThrustPerCmd * thrust_mask;
    const float t3 = cmd[3] * kThrustPerCmd * thrust_mask;
    fz = t0 + t1 + t2 + t3;
    tau.x = kRotorArm * (-t0 + t1 + t2 - t3);
    tau.y = kRotorArm * (-t0 + t1 - t2 + t3);
    tau.z = kYawTorquePerCmd * (-cmd[0] - cmd[1] + cmd[2] + cmd[3]);
}

// ---------------------------------------------------------------------------
// RK4 rigid body (oracle/rigid_body.py rk4_step; replaces gym.simulate, hovering.py:290)
// ---------------------------------------------------------------------------
struct Deriv { V3 acc; Q4 qd; V3 alpha; };

AG_HD Deriv rb_deriv(Q4 q, V3 wb, float fz, V3 tau) {
    Deriv d;
    const float zbx = 2.0f * (q.x * q.z + q.w * q.y);
    const float zby = 2.0f * (q.y * q.z - q.w * q.x);
    const float zbz = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    const float am = fz * kInvMass;
    d.acc = V3{am * zbx, am * zby, am * zbz + kGravityZ};
    d.alpha.x = (tau.x - (wb.y * (kIzz * wb.z) - wb.z * (kIyy * wb.y))) * kInvIxx;
    d.alpha.y = (tau.y - (wb.z * (kIxx * wb.x) - wb.x * (kIzz * wb.z))) * kInvIyy;
    d.alpha.z = (tau.z - (wb.x * (kIyy * wb.y) - wb.y * (kIxx * wb.x))) * kInvIzz;
    d.qd = Q4{0.5f * (q.w * wb.x + q.y * wb.z - q.z * wb.y),
              0.5f * (q.w * wb.y + q.z * wb.x - q.x * wb.z),
              0.5f * (q.w * wb.z + q.x * wb.y - q.y * wb.x),
              0.5f * (-q.x * wb.x - q.y * wb.y - q.z * wb.z)};
    return d;
}

AG_HD V3 clamp_norm(V3 v, float vmax) {
    const float n = fast_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (n > vmax) {
        const float s = vmax / n;
        v.x *= s; v.y *= s; v.z *= s;
    }
    return v;
}

#define AG_AXPY3(a, h, b) V3{(a).x + (h) * (b).x, (a).y + (h) * (b).y, (a).z + (h) * (b).z}
#define AG_AXPY4(a, h, b) Q4{(a).x + (h) * (b).x, (a).y + (h) * (b).y, (a).z + (h) * (b).z, (a).w + (h) * (b).w}
#define AG_RK_SUM(k1, k2, k3, k4) ((k1) + 2.0f * (k2) + 2.0f * (k3) + (k4))

AG_HD void rk4_step(EnvState& s, float fz, V3 tau, const StepParams& P) {
    const float h = P.half_dt, dt = P.dt, s6 = P.dt_over_6;
    const V3 wb = quat_rotate_inverse(s.q, s.w);
    const Deriv k1 = rb_deriv(s.q, wb, fz, tau);
    const V3 v2 = AG_AXPY3(s.v, h, k1.acc);
    const Deriv k2 = rb_deriv(AG_AXPY4(s.q, h, k1.qd), AG_AXPY3(wb, h, k1.alpha), fz, tau);
    const V3 v3 = AG_AXPY3(s.v, h, k2.acc);
    const Deriv k3 = rb_deriv(AG_AXPY4(s.q, h, k2.qd), AG_AXPY3(wb, h, k2.alpha), fz, tau);
    const V3 v4 = AG_AXPY3(s.v, dt, k3.acc);
    const Deriv k4 = rb_deriv(AG_AXPY4(s.q, dt, k3.qd), AG_AXPY3(wb, dt, k3.alpha), fz, tau);

    s.p = V3{s.p.x + s6 * AG_RK_SUM(s.v.x, v2.x, v3.x, v4.x),
             s.p.y + s6 * AG_RK_SUM(s.v.y, v2.y, v3.y, v4.y),
             s.p.z + s6 * AG_RK_SUM(s.v.z, v2.z, v3.z, v4.z)};
    V3 vn{s.v.x + s6 * AG_RK_SUM(k1.acc.x, k2.acc.x, k3.acc.x, k4.acc.x),
          s.v.y + s6 * AG_RK_SUM(k1.acc.y, k2.acc.y, k3.acc.y, k4.acc.y),
          s.v.z + s6 * AG_RK_SUM(k1.acc.z, k2.acc.z, k3.acc.z, k4.acc.z)};
    Q4 qn{s.q.x + s6 * AG_RK_SUM(k1.qd.x, k2.qd.x, k3.qd.x, k4.qd.x),
          s.q.y + s6 * AG_RK_SUM(k1.qd.y, k2.qd.y, k3.qd.y, k4.qd.y),
          s.q.z + s6 * AG_RK_SUM(k1.qd.z, k2.qd.z, k3.qd.z, k4.qd.z),
          s.q.w + s6 * AG_RK_SUM(k1.qd.w, k2.qd.w, k3.qd.w, k4.qd.w)};
    const V3 wbn{wb.x + s6 * AG_RK_SUM(k1.alpha.x, k2.alpha.x, k3.alpha.x, k4.alpha.x),
                 wb.y + s6 * AG_RK_SUM(k1.alpha.y, k2.alpha.y, k3.alpha.y, k4.alpha.y),
                 wb.z + s6 * AG_RK_SUM(k1.alpha.z, k2.alpha.z, k3.alpha.z, k4.alpha.z)};
    const float inv = fast_rsq(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    qn = Q4{qn.x * inv, qn.y * inv, qn.z * inv, qn.w * inv};
    s.q = qn;
    s.v = clamp_norm(vn, kMaxLinVel);
    s.w = clamp_norm(quat_rotate(qn, wbn), kMaxAngVel);
}

// ---------------------------------------------------------------------------
// Reset (hovering.py:310-335, tracking.py:159-192): u[12] in [0,1)
// ---------------------------------------------------------------------------
AG_HD void reset_state_from_uniforms(EnvState& s, const float* u, const StepParams& P) {
    // torch_rand_float(lo, hi) = (hi - lo) * u + lo   (airgym/utils/torch_utils.py:192-193)
    s.p.x = P.reset_pos_scale[0] * (2.0f * u[0] + -1.0f) + P.reset_pos_offset[0];
    s.p.y = P.reset_pos_scale[1] * (2.0f * u[1] + -1.0f) + P.reset_pos_offset[1];
    s.p.z = P.reset_pos_scale[2] * (2.0f * u[2] + -1.0f) + P.reset_pos_offset[2];
    const float a0 = P.reset_euler_scale[0] * (kTwoPi * u[3] + -kPi);
    const float a1 = P.reset_euler_scale[1] * (kTwoPi * u[4] + -kPi);
    const float a2 = P.reset_euler_scale[2] * (kTwoPi * u[5] + -kPi);
    // euler 'XYZ' (intrinsic) -> quaternion = qx(a0) * qy(a1) * qz(a2); equals
    // matrix_to_quaternion(euler_angles_to_matrix(.)) of hovering.py:323-324 (w > 0 for these small angles)
    const float cx = cosf(0.5f * a0), sx = sinf(0.5f * a0);
    const float cy = cosf(0.5f * a1), sy = sinf(0.5f * a1);
    const float cz = cosf(0.5f * a2), sz = sinf(0.5f * a2);
    s.q.w = cx * cy * cz - sx * sy * sz;
    s.q.x = sx * cy * cz + cx * sy * sz;
    s.q.y = cx * sy * cz - sx * cy * sz;
    s.q.z = cx * cy * sz + sx * sy * cz;
    s.v.x = P.reset_linvel_scale * (2.0f * u[6] + -1.0f);
    s.v.y = P.reset_linvel_scale * (2.0f * u[7] + -1.0f);
    s.v.z = P.reset_linvel_scale * (2.0f * u[8] + -1.0f);
    s.w.x = P.reset_angvel_scale * (2.0f * u[9] + -1.0f);
    s.w.y = P.reset_angvel_scale * (2.0f * u[10] + -1.0f);
    s.w.z = P.reset_angvel_scale * (2.0f * u[11] + -1.0f);
    s.progress = 0;
    s.was_reset = 1;
}

// ---------------------------------------------------------------------------
// Observation + reward
// ---------------------------------------------------------------------------
// quaternion_to_matrix (pytorch3d convention; oracle/rotations.py), row-major R[9]
AG_HD void quat_to_matrix(Q4 q, float* R) {
    const float r = q.w, i = q.x, j = q.y, k = q.z;
    const float two_s = 2.0f * fast_rcp(r * r + i * i + j * j + k * k);
    R[0] = 1.0f - two_s * (j * j + k * k);
    R[1] = two_s * (i * j - k * r);
    R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r);
    R[4] = 1.0f - two_s * (i * i + k * k);
    R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r);
    R[7] = two_s * (j * k + i * r);
    R[8] = 1.0f - two_s * (i * i + j * j);
}

// compute_yaw_diff, hovering.py:33-38
AG_HD float yaw_diff(float a, float b) {
    float d = b - a;
    d = d < -kPi ? d + kTwoPi : d;
    d = d > kPi ? d - kTwoPi : d;
    return d;
}

// tracking.py:194-200, reference point k (k = 0..9)
AG_HD V3 lemniscate_point(float st, float ct) {
    const float iden = fast_rcp(1.0f + ct * ct);
    return V3{3.0f * st * iden, 3.0f * st * ct * iden, 1.0f};
}
AG_HD V3 lemniscate_ref(int progress, int k, float dt) {
    const float t = (float)(progress + 5 * k) * dt * 0.25f;
    return lemniscate_point(sinf(t), cosf(t));
}
// All ten look-ahead points with ONE sinf / cosf pair (round 6; the ten pairs were a third of the Tracking step's instructions).
// The reference rounds every t_k = (progress + 5k) * dt * 0.25 to float32 on its own (tracking.py:196-197), so t_k is formed exactly
// as before; what changes is how sin / cos of it are evaluated: e_k = t_k - t_0 is a small angle (<= 45 dt * 0.25 plus the two
// roundings), known to ~1e-8, and sin(t_0 + e_k), cos(t_0 + e_k) follow from the angle-addition formulas with sin e_k / cos e_k as
// short Taylor sums (e_k <= 0.12 at dt = 0.01: the first dropped terms are e^7 / 5040 < 1e-10 and e^8 / 40320 < 1e-12).
// k = 0 is sinf / cosf themselves, bit for bit (the reward's ref_positions[:, 0], tracking.py:232).  Valid while 45 dt * 0.25 < 0.5.
AG_HD void lemniscate_refs(int progress, float dt, V3* r) {
    const float t0 = (float)progress * dt * 0.25f;
    const float s0 = sinf(t0), c0 = cosf(t0);
    r[0] = lemniscate_point(s0, c0);
#pragma unroll
    for (int k = 1; k < 10; ++k) {
        const float e = (float)(progress + 5 * k) * dt * 0.25f - t0;
        const float e2 = e * e;
        const float se = e * (1.0f + e2 * (-1.0f / 6.0f + e2 * (1.0f / 120.0f)));
        const float ce = 1.0f + e2 * (-0.5f + e2 * (1.0f / 24.0f + e2 * (-1.0f / 720.0f)));
        r[k] = lemniscate_point(s0 * ce + c0 * se, c0 * ce - s0 * se);
    }
}

struct StepOut {
    float rew;
    int done;
    int timeout;
    float terms[9];
    float cmd[4];
};

// compute_observations, hovering.py:337-342 / tracking.py:202-211: the noise-free part.
template <int TASK>
AG_HD void fill_clean_observations(const EnvState& s, const float* R, const StepParams& P, float* obs) {
#pragma unroll
    for (int i = 0; i < 9; ++i) obs[i] = R[i];
    obs[9] = s.p.x; obs[10] = s.p.y; obs[11] = s.p.z;
    obs[12] = s.v.x; obs[13] = s.v.y; obs[14] = s.v.z;
    obs[15] = s.w.x; obs[16] = s.w.y; obs[17] = s.w.z;
    if (TASK == TASK_TRACKING) {
        V3 r[10];
        lemniscate_refs(s.progress, P.dt, r);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            obs[18 + 3 * k + 0] = r[k].x - s.p.x;
            obs[18 + 3 * k + 1] = r[k].y - s.p.y;
            obs[18 + 3 * k + 2] = r[k].z - s.p.z;
        }
    }
}

// sigma_j for observation column j < 18 (add_noise, hovering.py:349-358)
AG_HD float noise_sigma(int j) { return j < 9 ? kSigMat : (j < 12 ? kSigPos : (j < 15 ? kSigVel : kSigAng)); }

// add_noise (hovering.py:343,349-358) then target subtraction (hovering.py:345; Tracking does not subtract).
// z = 18 standard normals.
template <int TASK>
AG_HD void apply_noise_and_target(float* obs, const float* z, const StepParams& P) {
    if (!P.noise_off) {
#pragma unroll
        for (int i = 0; i < 18; ++i) obs[i] += noise_sigma(i) * z[i];
    }
    if (TASK == TASK_HOVERING) {
#pragma unroll
        for (int i = 0; i < 18; ++i) obs[i] -= P.target[i];
    }
}

// compute_quadcopter_reward, hovering.py:371-459 / tracking.py:223-296
template <int TASK, int CTL>
AG_HD void compute_reward(const EnvState& s, const float* R, const float* a, const float* pre_a, const float* cmd,
                          const StepParams& P, StepOut& o) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr bool kHasThrust = (CTL == CTL_RATE || CTL == CTL_ATTI);
    const float c0 = clampf(cmd[0], 0.0f, 1.0f), c1 = clampf(cmd[1], 0.0f, 1.0f);
    const float c2 = clampf(cmd[2], 0.0f, 1.0f), c3 = clampf(cmd[3], 0.0f, 1.0f);
    const float effort = 0.1f * ((1.0f - c0) + (1.0f - c1) + (1.0f - c2) + (1.0f - c3)) * 0.25f;

    float d[A];
#pragma unroll
    for (int i = 0; i < A; ++i) d[i] = a[i] - pre_a[i];
    float cont, thrust_reward = 0.0f;
    if (!kHasThrust) {
        float n2 = 0.0f;
#pragma unroll
        for (int i = 0; i < A; ++i) n2 += d[i] * d[i];
        cont = 0.2f * fast_exp(-fast_sqrt(n2));
    } else {
        float n2 = 0.0f;
#pragma unroll
        for (int i = 0; i < A - 1; ++i) n2 += d[i] * d[i];
        const float dl = d[A - 1];
        if (TASK == TASK_HOVERING) {
            const float t3 = 3.0f * dl;
            cont = 0.2f * fast_exp(-fast_sqrt(n2)) + 0.5f * fast_rcp(1.0f + t3 * t3);
        } else {
            const float t2 = 2.0f * dl;
            cont = 0.1f * fast_exp(-fast_sqrt(n2)) + 0.5f * fast_rcp(1.0f + t2 * t2);
        }
        thrust_reward = 0.1f * (1.0f - fabsf(0.1533f - a[A - 1]));
    }

    // yaw from R: matrix_to_euler_angles(R,'XYZ')[2] = atan2(-R01, R00)
    const float yaw = atan2f(-R[1], R[0]);
    const float yd = yaw_diff(P.target_yaw, yaw) * (1.0f / kPi);
    const float spinnage = s.w.z * s.w.z;
    // ups = quat_axis(q, 2).z via quat_rotate, hovering.py:464-481: (2w^2-1) + 2 z^2
    const float ups_z = (2.0f * (s.q.w * s.q.w) - 1.0f) + 0.0f + s.q.z * s.q.z * 2.0f;
    const float hu = (ups_z + 1.0f) * 0.5f;
    const float ups_reward = hu * hu;

    int done = (s.progress >= P.max_episode_length - 1) ? 1 : 0;
    float reward;
    if (TASK == TASK_HOVERING) {
        const float rx = P.target[9] - s.p.x, ry = P.target[10] - s.p.y, rz = P.target[11] - s.p.z;
        const float pd2 = rx * rx + ry * ry + rz * rz;
        const float pos_diff = fast_sqrt(pd2);
        const float pd = 1.6f * pos_diff;
        const float pos_reward = 0.7f * fast_rcp(1.0f + pd * pd);
        // (rel/|rel|) . (v/|v|) with one rsq per vector (0/0 -> NaN as in the reference, Q8)
        const float ipd = fast_rsq(pd2);
        const float ivn = fast_rsq(s.v.x * s.v.x + s.v.y * s.v.y + s.v.z * s.v.z);
        const float dotp = (rx * ipd) * (s.v.x * ivn) + (ry * ipd) * (s.v.y * ivn) + (rz * ipd) * (s.v.z * ivn);
        // torch.clamp propagates NaN (the 0/0 of a zero velocity, quirk Q8); fminf/fmaxf would swallow it
        const float cdot = (dotp < -1.0f) ? -1.0f : ((dotp > 1.0f) ? 1.0f : dotp);
        const float angle = fabsf(acosf(cdot));
        const float vel_dir = 0.1f * fast_exp(-angle * (1.0f / kPi));
        const float y3 = 3.0f * yd;
        const float yaw_reward = fast_rcp(1.0f + y3 * y3);
        const float s3 = 3.0f * spinnage;
        const float spin_reward = fast_rcp(1.0f + s3 * s3);
        if (!kHasThrust)
            reward = cont + effort + pos_reward + pos_reward * (vel_dir + ups_reward + spin_reward + yaw_reward);
        else
            reward = cont + effort + thrust_reward + pos_reward + pos_reward * (vel_dir + ups_reward + spin_reward + yaw_reward);
        done = (pos_diff > 4.0f) ? 1 : done;
        done = (rz < -2.0f) ? 1 : done;
        done = (rz > 2.0f) ? 1 : done;
        done = (ups_z < 0.0f) ? 1 : done;
        o.terms[0] = cont; o.terms[1] = effort; o.terms[2] = thrust_reward; o.terms[3] = pos_reward;
        o.terms[4] = vel_dir; o.terms[5] = ups_reward; o.terms[6] = spin_reward; o.terms[7] = yaw_reward;
    } else {
        const V3 r0 = lemniscate_ref(s.progress, 0, P.dt);
        const float dx = r0.x - s.p.x, dy = r0.y - s.p.y, dz = r0.z - s.p.z;
        const float dist_norm = fast_sqrt(dx * dx + dy * dy + dz * dz);
        const float dn = 1.8f * dist_norm;
        const float dist_reward = fast_rcp(1.0f + dn * dn);
        const float y4 = 4.0f * yd;
        const float yaw_reward = fast_rcp(1.0f + y4 * y4);
        const float s2 = 2.0f * spinnage;
        const float spin_reward = fast_rcp(1.0f + s2 * s2);
        if (!kHasThrust)
            reward = cont + effort + dist_reward + dist_reward * (spin_reward + yaw_reward + ups_reward);
        else
            reward = cont + effort + thrust_reward + dist_reward + dist_reward * (spin_reward + yaw_reward + ups_reward);
        done = (dist_norm > 1.0f) ? 1 : done;
        o.terms[0] = dist_norm; o.terms[1] = dist_reward; o.terms[2] = yaw_reward; o.terms[3] = spin_reward;
        o.terms[4] = cont; o.terms[5] = thrust_reward; o.terms[6] = effort; o.terms[7] = ups_reward;
    }
    if (CTL == CTL_ATTI) done = (a[0] < 0.0f) ? 1 : done;  // hovering.py:445-446
    o.terms[8] = reward;
    o.rew = reward;
    o.done = done;
}

// time_out_buf, hovering.py:304: `progress_buf > max_episode_length`, evaluated AFTER reset_idx has zeroed the progress of
// every env that reached max - 1 (:435) - never true, so the PPO loop's time-out bootstrap (a2c_base.py:672-673) never
// fires (quirk Q3; reproduced by default).  P.fix_time_outs (AG_FLAG_FIX_TIME_OUTS, opt-in) flags instead the envs whose
// episode reached the time limit this step: progress_end = progress_buf after the increment, before the reset.
AG_HD int step_timeout(int progress_end, int progress_now, const StepParams& P) {
    return P.fix_time_outs ? ((progress_end >= P.max_episode_length - 1) ? 1 : 0)
                           : ((progress_now > P.max_episode_length) ? 1 : 0);
}

// ---------------------------------------------------------------------------
// One full env step (Hovering.step, hovering.py:286-308), in two halves so that a kernel can put the state stores
// between them (the state is final after the physics unless the env terminates):
//   env_step_physics : action map -> canonicalise -> controller -> wrench -> RK4 -> progress++
//   env_step_outputs : observation (+noise) -> reward / termination -> pre_actions -> in-place reset
//   raw_action : what the agent passed (A floats);  a[] : the processed action (out of the first half)
//   pre_a      : in: previous processed action; out: this step's (zeroed if the env reset)
//   EXT = parity mode: ext_noise[18] / ext_uniforms[12] supplied by the caller instead of Philox
//   CLEAN_OBS = obs[] is returned WITHOUT noise and target subtraction (added by the noise wave's data later)
// ---------------------------------------------------------------------------
template <int TASK, int CTL>
AG_HD void env_step_physics(EnvState& s, CtlState& c, const float* raw_action, const StepParams& P, float* a, float* cmd) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    // ---- pre_physics_step, hovering.py:212-216
    float lo[A], hi[A];
    action_limits<TASK, CTL>(lo, hi);
#pragma unroll
    for (int i = 0; i < A; ++i) a[i] = raw_action[i];
    if (CTL == CTL_RATE || CTL == CTL_ATTI) a[A - 1] = 0.5f + 0.5f * a[A - 1];
#pragma unroll
    for (int i = 0; i < A; ++i) a[i] = fmaxf(fminf(a[i], hi[i]), lo[i]);  // tensor_clamp, torch_utils.py:199-201
    // quaternion canonicalisation w >= 0, hovering.py:224-226
    if (s.q.w < 0.0f) { s.q.x = -s.q.x; s.q.y = -s.q.y; s.q.z = -s.q.z; s.q.w = -s.q.w; }
    // controller, hovering.py:234-254
    controller_update<CTL>(c, s, a, cmd);
    // wrench, hovering.py:256-277 (thrust zeroed for envs reset last step, reaction torque kept)
    float fz;
    V3 tau;
    body_wrench_from_cmd(cmd, s.was_reset ? 0.0f : 1.0f, fz, tau);
    // ---- gym.simulate
    rk4_step(s, fz, tau, P);
    // ---- progress, hovering.py:297
    s.progress += 1;
    s.was_reset = 0;
}

// compute_observations + compute_reward on the post-physics state (hovering.py:298-299,337-459); no state change.
template <int TASK, int CTL, bool EXT, bool CLEAN_OBS>
AG_HD void env_observe_reward(const EnvState& s, const float* a, const float* pre_a, const float* cmd, const StepParams& P,
                              uint32_t env_global, const float* ext_noise, float* obs, StepOut& o) {
    float R[9];
    quat_to_matrix(s.q, R);
    fill_clean_observations<TASK>(s, R, P, obs);
    if (!CLEAN_OBS) {   // CLEAN_OBS: noise + target are applied by the caller (wave-specialised kernel)
        float z[18];
        if (EXT) {
#pragma unroll
            for (int i = 0; i < 18; ++i) z[i] = ext_noise[i];
        } else if (!P.noise_off) {
            obs_noise_normals(P, env_global, z);
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) z[i] = 0.0f;
        }
        apply_noise_and_target<TASK>(obs, z, P);
    }
    compute_reward<TASK, CTL>(s, R, a, pre_a, cmd, P, o);
}

// reset_idx for a done env, hovering.py:300-302,310-335 (+ controller memory, pre_actions)
template <int CTL, bool EXT>
AG_HD void env_reset_done(EnvState& s, CtlState& c, float* pre_a, const StepParams& P, uint32_t env_global,
                          const float* ext_uniforms) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float u[12];
    if (EXT) {
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = ext_uniforms[i];
    } else {
        reset_uniforms(P, env_global, u);
    }
    reset_state_from_uniforms(s, u, P);
    ctl_reset(c, s);
#pragma unroll
    for (int i = 0; i < A; ++i) pre_a[i] = 0.0f;
}

template <int TASK, int CTL, bool EXT, bool CLEAN_OBS = false>
AG_HD void env_step(EnvState& s, CtlState& c, float* pre_a, const float* raw_action, const StepParams& P,
                    uint32_t env_global, const float* ext_noise, const float* ext_uniforms, float* obs, StepOut& o) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float a[A];
    env_step_physics<TASK, CTL>(s, c, raw_action, P, a, o.cmd);
    env_observe_reward<TASK, CTL, EXT, CLEAN_OBS>(s, a, pre_a, o.cmd, P, env_global, ext_noise, obs, o);
#pragma unroll
    for (int i = 0; i < A; ++i) pre_a[i] = a[i];  // hovering.py:369
    const int progress_end = s.progress;
    s.was_reset = o.done;
    if (o.done) env_reset_done<CTL, EXT>(s, c, pre_a, P, env_global, ext_uniforms);
    o.timeout = step_timeout(progress_end, s.progress, P);  // hovering.py:304 (never true by default, Q3)
}

// reset_idx(all) at creation / BaseTask.reset (base_task.py:107-111)
AG_HD void env_reset(EnvState& s, CtlState& c, float* pre_a, int num_actions, const StepParams& P, uint32_t env_global) {
    float u[12];
    reset_uniforms(P, env_global, u);
    reset_state_from_uniforms(s, u, P);
    ctl_reset(c, s);
    for (int i = 0; i < num_actions; ++i) pre_a[i] = 0.0f;
}

}  // namespace ag
