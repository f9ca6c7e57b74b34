// planning_math.hpp - per-env arithmetic of the Planning task, float32.
//
// Reference-owned arithmetic restated here (oracle/planning_ref.py, pinned by tests/golden/planning_*.npz):
//   airgym/envs/task/planning.py        reset_idx :63-136, compute_observations :186-214,
//                                       compute_quadcopter_reward :223-307, step order :138-184
//   airgym/envs/base/customized.py      pre_physics_step :216-298 (rate limits +-1; the clamped copy drives the
//                                       controller, self.actions keeps the thrust-remapped raw action),
//                                       dump_images :399-435 (depth post-processing, see planning_kernel.hip)
// Build-defined spec (IsaacGym's PhysX contacts and rasteriser are closed; oracle/planning_ref.py is the spec):
//   the analytic scene (40 capped cylinders from airgym_amd/assets/thin_trees.json, goal sphere, ground plane),
//   the pin-hole z-depth ray-cast and the sphere-vs-cylinder collision test.
// Host-compilable like env_math.hpp so tests/host_harness can check it on the GPU-less build box.
#pragma once

#include "env_math.hpp"

namespace ag {

enum : int { TASK_PLANNING = 2 };
enum : uint32_t { STREAM_IMG_ADD = 2, STREAM_IMG_MUL = 3, STREAM_IMG_KERNEL = 4, STREAM_VARIANT = 5 };

constexpr int kNumObst = 40;
constexpr int kNumVariants = 100;
constexpr int kCamW = 212, kCamH = 120, kCamPix = kCamW * kCamH;
constexpr float kCamFar = 5.0f;
constexpr float kCamOffX = 0.15f, kCamOffY = 0.0f, kCamOffZ = 0.1f;   // planning_config.py:60
constexpr float kRobotRadius = 0.2f, kGoalRadius = 0.2f;
constexpr float kLength = 8.0f, kWidth = 4.0f, kFlyHeight = 1.5f;     // planning.py:10-12
constexpr int kPlanResetUniforms = 3 * kNumObst + 1;
constexpr int kPlanNumObs = 16;
constexpr int kPlanNumTerms = 11;
// (W/2) / tan(87 deg / 2)
constexpr float kCamFx = 111.70069327978202f;
constexpr float kInf = __builtin_huge_valf();

struct Cyl { float cx, cy, cz, nx, ny, nz, r, h; };

// obstacle root pose (x, y, yaw) + variant row (centre3, axis3, radius, half length) -> world-frame capped cylinder
AG_HD Cyl world_cylinder(float x, float y, float yaw, const float* tab8) {
    const float c = cosf(yaw), s = sinf(yaw);
    Cyl w;
    w.cx = x + (c * tab8[0] - s * tab8[1]);
    w.cy = y + (s * tab8[0] + c * tab8[1]);
    w.cz = tab8[2];
    w.nx = c * tab8[3] - s * tab8[4];
    w.ny = s * tab8[3] + c * tab8[4];
    w.nz = tab8[5];
    w.r = tab8[6];
    w.h = tab8[7];
    return w;
}

// smallest t > 0 with o + t d on the capped cylinder, kInf if none (oracle: _ray_capped_cylinders)
AG_HD float ray_capped_cylinder(V3 o, V3 d, const Cyl& c) {
    const V3 oc{o.x - c.cx, o.y - c.cy, o.z - c.cz};
    const float dn = d.x * c.nx + d.y * c.ny + d.z * c.nz;
    const float on = oc.x * c.nx + oc.y * c.ny + oc.z * c.nz;
    const V3 dp{d.x - dn * c.nx, d.y - dn * c.ny, d.z - dn * c.nz};
    const V3 op{oc.x - on * c.nx, oc.y - on * c.ny, oc.z - on * c.nz};
    const float a = dp.x * dp.x + dp.y * dp.y + dp.z * dp.z;
    const float b = dp.x * op.x + dp.y * op.y + dp.z * op.z;
    const float cc = (op.x * op.x + op.y * op.y + op.z * op.z) - c.r * c.r;
    const float disc = b * b - a * cc;
    float best = kInf;
    if (disc >= 0.0f && a > 1e-12f) {
        const float sq = fast_sqrt(disc);
        const float ia = fast_rcp(a);
        const float t0 = (-b - sq) * ia, t1 = (-b + sq) * ia;
        if (t0 > 0.0f && fabsf(on + t0 * dn) <= c.h) best = fminf(best, t0);
        if (t1 > 0.0f && fabsf(on + t1 * dn) <= c.h) best = fminf(best, t1);
    }
    if (fabsf(dn) > 1e-12f) {
        const float idn = fast_rcp(dn);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sgn = k == 0 ? 1.0f : -1.0f;
            const float t = (sgn * c.h - on) * idn;
            const V3 p{op.x + t * dp.x, op.y + t * dp.y, op.z + t * dp.z};
            if (t > 0.0f && (p.x * p.x + p.y * p.y + p.z * p.z) <= c.r * c.r) best = fminf(best, t);
        }
    }
    return best;
}

AG_HD float point_cylinder_distance(V3 p, const Cyl& c) {
    const V3 d{p.x - c.cx, p.y - c.cy, p.z - c.cz};
    const float y = d.x * c.nx + d.y * c.ny + d.z * c.nz;
    const float rad = sqrtf(fmaxf((d.x * d.x + d.y * d.y + d.z * d.z) - y * y, 0.0f));
    const float dr = fmaxf(rad - c.r, 0.0f);
    const float dy = fmaxf(fabsf(y) - c.h, 0.0f);
    return sqrtf(dr * dr + dy * dy);
}

struct Camera { V3 o; float R[9]; };

AG_HD Camera make_camera(V3 pos, Q4 q) {
    Camera cam;
    const float r = q.w, i = q.x, j = q.y, k = q.z;
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    cam.R[0] = 1.0f - two_s * (j * j + k * k); cam.R[1] = two_s * (i * j - k * r); cam.R[2] = two_s * (i * k + j * r);
    cam.R[3] = two_s * (i * j + k * r); cam.R[4] = 1.0f - two_s * (i * i + k * k); cam.R[5] = two_s * (j * k - i * r);
    cam.R[6] = two_s * (i * k - j * r); cam.R[7] = two_s * (j * k + i * r); cam.R[8] = 1.0f - two_s * (i * i + j * j);
    cam.o = V3{pos.x + (cam.R[0] * kCamOffX + cam.R[1] * kCamOffY + cam.R[2] * kCamOffZ),
               pos.y + (cam.R[3] * kCamOffX + cam.R[4] * kCamOffY + cam.R[5] * kCamOffZ),
               pos.z + (cam.R[6] * kCamOffX + cam.R[7] * kCamOffY + cam.R[8] * kCamOffZ)};
    return cam;
}

// world direction of pixel (u, v): body direction (1, (W/2 - (u+.5))/fx, (H/2 - (v+.5))/fx); t along it IS the z-depth
AG_HD V3 pixel_direction(const Camera& cam, int u, int v) {
    constexpr float kInvFx = (float)(1.0 / 111.70069327978202);
    const float dy = ((float)kCamW / 2.0f - ((float)u + 0.5f)) * kInvFx;
    const float dz = ((float)kCamH / 2.0f - ((float)v + 0.5f)) * kInvFx;
    return V3{cam.R[0] + dy * cam.R[1] + dz * cam.R[2], cam.R[3] + dy * cam.R[4] + dz * cam.R[5],
              cam.R[6] + dy * cam.R[7] + dz * cam.R[8]};
}

// z-depth of one pixel against `n` cylinders, the ground plane and the goal sphere; kInf beyond the far plane
AG_HD float depth_pixel(const Camera& cam, V3 d, const Cyl* cyl, int n, V3 goal) {
    float t = kInf;
    for (int k = 0; k < n; ++k) t = fminf(t, ray_capped_cylinder(cam.o, d, cyl[k]));
    if (d.z < -1e-12f) {
        const float tg = -cam.o.z * fast_rcp(d.z);
        if (tg > 0.0f) t = fminf(t, tg);
    }
    const V3 oc{cam.o.x - goal.x, cam.o.y - goal.y, cam.o.z - goal.z};
    const float a = d.x * d.x + d.y * d.y + d.z * d.z;
    const float b = d.x * oc.x + d.y * oc.y + d.z * oc.z;
    const float c = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - kGoalRadius * kGoalRadius;
    const float disc = b * b - a * c;
    if (disc >= 0.0f) {
        const float ts = (-b - fast_sqrt(disc)) * fast_rcp(a);
        if (ts > 0.0f) t = fminf(t, ts);
    }
    return t <= kCamFar ? t : kInf;
}

// Per-(camera, cylinder) constants of the ray test: everything in ray_capped_cylinder that depends on the ray ORIGIN
// only (all pixels of an image share it).
struct CylView { float nx, ny, nz, on, opx, opy, opz, cc, r2, h; };

AG_HD CylView make_cyl_view(V3 o, const Cyl& c) {
    CylView w;
    const V3 oc{o.x - c.cx, o.y - c.cy, o.z - c.cz};
    w.nx = c.nx; w.ny = c.ny; w.nz = c.nz;
    w.on = oc.x * c.nx + oc.y * c.ny + oc.z * c.nz;
    w.opx = oc.x - w.on * c.nx; w.opy = oc.y - w.on * c.ny; w.opz = oc.z - w.on * c.nz;
    w.r2 = c.r * c.r;
    w.cc = (w.opx * w.opx + w.opy * w.opy + w.opz * w.opz) - w.r2;
    w.h = c.h;
    return w;
}

// identical arithmetic to ray_capped_cylinder with the origin terms precomputed
AG_HD float ray_cyl_view(V3 d, const CylView& c) {
    const float dn = d.x * c.nx + d.y * c.ny + d.z * c.nz;
    const V3 dp{d.x - dn * c.nx, d.y - dn * c.ny, d.z - dn * c.nz};
    const float a = dp.x * dp.x + dp.y * dp.y + dp.z * dp.z;
    const float b = dp.x * c.opx + dp.y * c.opy + dp.z * c.opz;
    const float disc = b * b - a * c.cc;
    float best = kInf;
    if (disc >= 0.0f && a > 1e-12f) {
        const float sq = fast_sqrt(disc);
        const float ia = fast_rcp(a);
        const float t0 = (-b - sq) * ia, t1 = (-b + sq) * ia;
        if (t0 > 0.0f && fabsf(c.on + t0 * dn) <= c.h) best = fminf(best, t0);
        if (t1 > 0.0f && fabsf(c.on + t1 * dn) <= c.h) best = fminf(best, t1);
    }
    if (fabsf(dn) > 1e-12f) {
        const float idn = fast_rcp(dn);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sgn = k == 0 ? 1.0f : -1.0f;
            const float t = (sgn * c.h - c.on) * idn;
            const V3 p{c.opx + t * dp.x, c.opy + t * dp.y, c.opz + t * dp.z};
            if (t > 0.0f && (p.x * p.x + p.y * p.y + p.z * p.z) <= c.r2) best = fminf(best, t);
        }
    }
    return best;
}

// same result as depth_pixel: cylinders whose conservative column interval [ulo, uhi] excludes column u are skipped
AG_HD float depth_pixel_culled(const Camera& cam, V3 d, const CylView* cyl, const int* ulo, const int* uhi, int u, int n, V3 goal) {
    float t = kInf;
    for (int k = 0; k < n; ++k) {
        if (u < ulo[k] || u > uhi[k]) continue;
        t = fminf(t, ray_cyl_view(d, cyl[k]));
    }
    if (d.z < -1e-12f) {
        const float tg = -cam.o.z * fast_rcp(d.z);
        if (tg > 0.0f) t = fminf(t, tg);
    }
    const V3 oc{cam.o.x - goal.x, cam.o.y - goal.y, cam.o.z - goal.z};
    const float a = d.x * d.x + d.y * d.y + d.z * d.z;
    const float b = d.x * oc.x + d.y * oc.y + d.z * oc.z;
    const float c = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - kGoalRadius * kGoalRadius;
    const float disc = b * b - a * c;
    if (disc >= 0.0f) {
        const float ts = (-b - fast_sqrt(disc)) * fast_rcp(a);
        if (ts > 0.0f) t = fminf(t, ts);
    }
    return t <= kCamFar ? t : kInf;
}

// ---------------------------------------------------------------------------
struct PlanExtra {
    V3 goal;
    float prev_related_dist;   // planning.py:183 (stored, never read by the reward)
    V3 pre_pos;                // pre_root_positions, planning.py:221
    float esdf;                // min pixel of the post-processed image (planning.py:162-163, quirk Q16)
};

struct PlanOut {
    float rew;
    int done;
    int timeout;
    float terms[kPlanNumTerms];   // continous_action, heading, speed, forward, alive, ups, z, esdf, thrust, reach_goal, reward
    float related_dist;
};

// Customized action limits, customized.py:93-123 (rate mode: +-1)
template <int CTL>
AG_HD void planning_action_limits(float* lo, float* hi) {
    if (CTL == CTL_RATE) {
        for (int i = 0; i < 3; ++i) { lo[i] = -1.0f; hi[i] = 1.0f; }
        lo[3] = 0.0f; hi[3] = 1.0f;
    } else {
        action_limits<TASK_HOVERING, CTL>(lo, hi);
    }
}

// self.actions of customized.py:224-229: raw action with the thrust channel remapped, NOT clamped
template <int CTL>
AG_HD void planning_process_action(const float* raw, float* a) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
#pragma unroll
    for (int i = 0; i < A; ++i) a[i] = raw[i];
    if (CTL == CTL_RATE || CTL == CTL_ATTI) a[A - 1] = 0.5f + 0.5f * a[A - 1];
}

// pre_physics_step + gym.simulate (customized.py:216-298, planning.py:146-151)
template <int CTL>
AG_HD void planning_physics(EnvState& s, CtlState& c, const float* raw_action, const StepParams& P) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float a[A], lo[A], hi[A], cl[A], cmd[4];
    planning_process_action<CTL>(raw_action, a);
    planning_action_limits<CTL>(lo, hi);
#pragma unroll
    for (int i = 0; i < A; ++i) cl[i] = fmaxf(fminf(a[i], hi[i]), lo[i]);
    if (s.q.w < 0.0f) { s.q.x = -s.q.x; s.q.y = -s.q.y; s.q.z = -s.q.z; s.q.w = -s.q.w; }
    controller_update<CTL>(c, s, cl, cmd);
    float fz;
    V3 tau;
    body_wrench_from_cmd(cmd, s.was_reset ? 0.0f : 1.0f, fz, tau);
    rk4_step(s, fz, tau, P);
}

// reset_idx, planning.py:63-136.  u[121]: per obstacle (x, y, yaw), then goal y.  Obstacles are written through
// ob (x, y, yaw at ob[j*stride + 0..2]).
AG_HD void planning_reset(EnvState& s, CtlState& c, PlanExtra& x, float* pre_a, int num_actions, const float* u,
                          float* ob, size_t stride) {
    for (int j = 0; j < kNumObst; ++j) {
        ob[j * stride + 0] = kLength * (2.0f * u[3 * j + 0] + -1.0f) + 0.0f;
        ob[j * stride + 1] = kWidth * (2.0f * u[3 * j + 1] + -1.0f) + 0.0f;
        ob[j * stride + 2] = kTwoPi * u[3 * j + 2] + -kPi;
    }
    x.goal = V3{kLength + 0.5f, 1.5f * (2.0f * u[3 * kNumObst] + -1.0f) + 0.0f, kFlyHeight};
    s.p = V3{-kLength - 0.5f, 0.0f, kFlyHeight};
    // yaw towards the goal; euler (0, 0, yaw) -> quaternion (0, 0, sin(yaw/2), cos(yaw/2))
    const float yaw = atan2f(x.goal.y - s.p.y, x.goal.x - s.p.x);
    s.q = Q4{0.0f, 0.0f, sinf(0.5f * yaw), cosf(0.5f * yaw)};
    s.v = V3{0.0f, 0.0f, 0.0f};
    s.w = V3{0.0f, 0.0f, 0.0f};
    s.progress = 0;
    s.was_reset = 1;
    x.prev_related_dist = 0.0f;
    x.pre_pos = V3{0.0f, 0.0f, 0.0f};
    ctl_reset(c, s);
    for (int i = 0; i < num_actions; ++i) pre_a[i] = 0.0f;
}

AG_HD void planning_reset_uniforms(const StepParams& P, uint32_t env_global, float* u /*[124]*/) {
    for (int b = 0; b < (kPlanResetUniforms + 3) / 4; ++b) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_RESET, (uint32_t)b, P.key0, P.key1);
        u[4 * b + 0] = u32_to_unit(r.x); u[4 * b + 1] = u32_to_unit(r.y);
        u[4 * b + 2] = u32_to_unit(r.z); u[4 * b + 3] = u32_to_unit(r.w);
    }
}

// progress++, compute_observations, compute_reward (planning.py:158-166,186-307) given the collision flag and x.esdf
template <int CTL>
AG_HD void planning_post(EnvState& s, PlanExtra& x, float* pre_a, const float* raw_action, int collided,
                         const StepParams& P, float* obs, PlanOut& o) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float a[A];
    planning_process_action<CTL>(raw_action, a);
    s.progress += 1;
    // ---- compute_observations
    const V3 fwd{x.goal.x - s.p.x, x.goal.y - s.p.y, x.goal.z - s.p.z};
    float R[9];
    {   // quaternion_to_matrix exactly as the oracle (division, not rcp: these feed the observation directly)
        const float r = s.q.w, i = s.q.x, j = s.q.y, k = s.q.z;
        const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
        R[0] = 1.0f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
        R[3] = two_s * (i * j + k * r); R[4] = 1.0f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
        R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.0f - two_s * (i * i + j * j);
    }
    const float yaw = atan2f(R[3], R[0]);
    const float cy = cosf(yaw), sy = sinf(yaw);
    // world_to_local built with stack(..., dim=2): W[i][j] = col_j[i] -> rows (cy, sy, 0), (-sy, cy, 0), (0, 0, 1)
    // (planning.py:195-199)
    float L[9];
    L[0] = cy * R[0] + sy * R[3]; L[1] = cy * R[1] + sy * R[4]; L[2] = cy * R[2] + sy * R[5];
    L[3] = -sy * R[0] + cy * R[3]; L[4] = -sy * R[1] + cy * R[4]; L[5] = -sy * R[2] + cy * R[5];
    L[6] = R[6]; L[7] = R[7]; L[8] = R[8];
    const float e0 = atan2f(-L[5], L[8]), e1 = asinf(L[2]), e2 = atan2f(-L[1], L[0]);
    const V3 pl{cy * fwd.x + sy * fwd.y, -sy * fwd.x + cy * fwd.y, fwd.z};
    const V3 vl{cy * s.v.x + sy * s.v.y, -sy * s.v.x + cy * s.v.y, s.v.z};
    const V3 wl{cy * s.w.x + sy * s.w.y, -sy * s.w.x + cy * s.w.y, s.w.z};
    const float pln = sqrtf(pl.x * pl.x + pl.y * pl.y + pl.z * pl.z);
    const V3 gdir{pl.x / pln, pl.y / pln, pl.z / pln};
    const float related = sqrtf(fwd.x * fwd.x + fwd.y * fwd.y + fwd.z * fwd.z);
    obs[0] = gdir.x; obs[1] = gdir.y; obs[2] = gdir.z;
    obs[3] = e0; obs[4] = e1; obs[5] = e2;
    obs[6] = vl.x; obs[7] = vl.y; obs[8] = vl.z;
    obs[9] = wl.x; obs[10] = wl.y; obs[11] = wl.z;
    obs[12] = a[0]; obs[13] = a[1]; obs[14] = a[2]; obs[15] = a[3];
    // ---- compute_quadcopter_reward
    float dn2 = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) { const float d = a[i] - pre_a[i]; dn2 += d * d; }
    const float cont = 0.2f * sqrtf(wl.x * wl.x + wl.y * wl.y + wl.z * wl.z) + 0.2f * sqrtf(dn2);
    const float thrust_reward = 0.5f * (1.0f - fabsf(0.1533f - a[A - 1]));
    const V3 gp{x.goal.x - x.pre_pos.x, x.goal.y - x.pre_pos.y, x.goal.z - x.pre_pos.z};
    const float forward_reward = 0.1f * (sqrtf(gp.x * gp.x + gp.y * gp.y + gp.z * gp.z) - related);
    const float heading = gdir.x * 1.0f + gdir.y * 0.0f + gdir.z * 0.0f;
    const float sv = vl.x - 1.0f;
    const float speed_reward = -0.5f * (1.0f - expf(-2.0f * (sv * sv)));
    const float z_reward = fminf(fminf(s.p.z - 1.8f, 0.0f), 1.2f - s.p.z);
    const float ups_z = (2.0f * (s.q.w * s.q.w) - 1.0f) + 0.0f + s.q.z * s.q.z * 2.0f;
    const float hu = (ups_z + 1.0f) / 2.0f;
    const float ups_reward = hu * hu;
    const float esdf_reward = 0.5f * (1.0f - expf(-0.5f * (x.esdf * x.esdf)));
    const float alive = x.esdf > 0.3f ? 0.0f : -1.0f;
    const bool reach = related < 0.3f;
    const float reach_reward = reach ? 200.0f : 0.0f;
    const float reward = cont + forward_reward + alive + esdf_reward + ups_reward + z_reward + speed_reward + heading
                         + thrust_reward + reach_reward;
    int done = (s.p.z < kFlyHeight - 0.3f) ? 1 : 0;
    done = (s.p.z > kFlyHeight + 0.3f) ? 1 : done;
    done = (s.p.x < -kLength - 0.5f) ? 1 : done;
    done = (s.p.x > kLength + 0.5f) ? 1 : done;
    done = (s.p.y < -kWidth) ? 1 : done;
    done = (s.p.y > kWidth) ? 1 : done;
    done = collided ? 1 : done;
    done = reach ? 1 : done;
    done = (heading < 0.25f) ? 1 : done;
    done = (s.progress >= P.max_episode_length - 1) ? 1 : done;
    o.terms[0] = cont; o.terms[1] = heading; o.terms[2] = speed_reward; o.terms[3] = forward_reward; o.terms[4] = alive;
    o.terms[5] = ups_reward; o.terms[6] = z_reward; o.terms[7] = esdf_reward; o.terms[8] = thrust_reward;
    o.terms[9] = reach_reward; o.terms[10] = reward;
    o.rew = reward;
    o.done = done;
    o.related_dist = related;
    // update prev (planning.py:216-221); prev_related_dist is set after the optional reset (planning.py:183)
#pragma unroll
    for (int i = 0; i < A; ++i) pre_a[i] = a[i];
    x.pre_pos = s.p;
    s.was_reset = done;
}

}  // namespace ag

// =====================================================================================================================
// Balloon and Avoid: the other two tasks of the Customized family.  Same pre_physics_step as
// Planning (planning_physics above); what follows restates airgym/envs/task/balloon.py and avoid.py (oracle/custom_ref.py,
// pinned by tests/golden/{balloon,avoid}_*.npz recorded from the reference's own methods).
// =====================================================================================================================
namespace ag {

enum : int { TASK_BALLOON = 3, TASK_AVOID = 4 };

constexpr int kBalloonNumObs = 18, kBalloonNumTerms = 6, kBalloonResetUniforms = 15;
constexpr int kAvoidNumObs = 16, kAvoidNumTerms = 8, kAvoidResetUniforms = 11;
constexpr float kCubeHalf = 0.15f;        // env_assets/cubes/1x1: unit cube under a 0.15 scale node
constexpr int kCustomMaxTerms = 8;

struct CustomOut {
    float rew;
    int done;
    int timeout;
    float terms[kCustomMaxTerms];
};

// F.normalize(v, dim=-1): v / max(|v|, 1e-12)
AG_HD V3 normalize_eps(V3 v) {
    const float n = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z), 1e-12f);
    return V3{v.x / n, v.y / n, v.z / n};
}

AG_HD void quat_to_matrix_div(Q4 q, float* R) {   // quaternion_to_matrix with a true division (feeds observations)
    const float r = q.w, i = q.x, j = q.y, k = q.z;
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1.0f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1.0f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.0f - two_s * (i * i + j * j);
}

// ---------------------------------------------------------------------------------------------------------- Balloon
// reset_idx, balloon.py:57-99.  u[15]: balloon x y z | root x y | root z | euler x y z | linvel(3) | angvel(3)
AG_HD void balloon_reset(EnvState& s, CtlState& c, V3& balloon, V3& pre_pos, float* pre_a, int num_actions, const float* u) {
    balloon.x = 0.5f * (2.0f * u[0] + -1.0f) + 2.5f;
    balloon.y = 2.0f * (2.0f * u[1] + -1.0f) + 0.0f;
    balloon.z = 0.3f * (2.0f * u[2] + -1.0f) + 1.0f;
    s.p.x = 0.1f * (2.0f * u[3] + -1.0f) + 0.0f;
    s.p.y = 0.1f * (2.0f * u[4] + -1.0f) + 0.0f;
    s.p.z = 0.2f * (2.0f * u[5] + -1.0f) + 1.0f;
    const float a0 = 0.1f * (kTwoPi * u[6] + -kPi);
    const float a1 = 0.1f * (kPi * u[7] + 0.0f);                 // torch_rand_float(0, pi)
    const float a2 = 0.2f * (kTwoPi * u[8] + -kPi);
    const float cx = cosf(0.5f * a0), sx = sinf(0.5f * a0);
    const float cy = cosf(0.5f * a1), sy = sinf(0.5f * a1);
    const float cz = cosf(0.5f * a2), sz = sinf(0.5f * a2);
    s.q.w = cx * cy * cz - sx * sy * sz;                          // euler 'XYZ' -> quaternion (w > 0 for these angles)
    s.q.x = sx * cy * cz + cx * sy * sz;
    s.q.y = cx * sy * cz - sx * cy * sz;
    s.q.z = cx * cy * sz + sx * sy * cz;
    s.v = V3{0.5f * (2.0f * u[9] + -1.0f), 0.5f * (2.0f * u[10] + -1.0f), 0.5f * (2.0f * u[11] + -1.0f)};
    s.w = V3{0.2f * (2.0f * u[12] + -1.0f), 0.2f * (2.0f * u[13] + -1.0f), 0.2f * (2.0f * u[14] + -1.0f)};
    s.progress = 0;
    s.was_reset = 1;
    pre_pos = V3{0.0f, 0.0f, 0.0f};
    ctl_reset(c, s);
    for (int i = 0; i < num_actions; ++i) pre_a[i] = 0.0f;
}

AG_HD void custom_reset_uniforms(const StepParams& P, uint32_t env_global, float* u /*[16]*/) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_RESET, (uint32_t)b, P.key0, P.key1);
        u[4 * b + 0] = u32_to_unit(r.x); u[4 * b + 1] = u32_to_unit(r.y);
        u[4 * b + 2] = u32_to_unit(r.z); u[4 * b + 3] = u32_to_unit(r.w);
    }
}

// progress++, compute_observations, compute_reward (balloon.py:129-165,172-237).  z[18]: standard normals of add_noise.
template <int CTL>
AG_HD void balloon_post(EnvState& s, V3 balloon, V3& pre_pos, float* pre_a, const float* raw_action, int collided,
                        const float* z, const StepParams& P, float* obs, CustomOut& o) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float a[A];
    planning_process_action<CTL>(raw_action, a);
    s.progress += 1;
    float R[9];
    quat_to_matrix_div(s.q, R);
    // (state + sigma * noise) - target: the balloon never rotates (static actor), its matrix is the identity
#pragma unroll
    for (int i = 0; i < 9; ++i) obs[i] = (R[i] + kSigMat * z[i]) - ((i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f);
    obs[9] = (s.p.x + kSigPos * z[9]) - balloon.x;
    obs[10] = (s.p.y + kSigPos * z[10]) - balloon.y;
    obs[11] = (s.p.z + kSigPos * z[11]) - balloon.z;
    obs[12] = s.v.x + kSigVel * z[12]; obs[13] = s.v.y + kSigVel * z[13]; obs[14] = s.v.z + kSigVel * z[14];
    obs[15] = s.w.x + kSigAng * z[15]; obs[16] = s.w.y + kSigAng * z[16]; obs[17] = s.w.z + kSigAng * z[17];
    // ---- compute_quadcopter_reward
    const V3 rel{balloon.x - s.p.x, balloon.y - s.p.y, balloon.z - s.p.z};
    const V3 dir = normalize_eps(rel);
    const float direction_yaw = atan2f(dir.y, dir.x);
    const float root_yaw = atan2f(-R[1], R[0]);                  // matrix_to_euler_angles(R, 'XYZ')[2]
    const float yaw_distance = fabsf(yaw_diff(root_yaw, direction_yaw));
    const float yd = 1.6f * yaw_distance;
    const float yaw_reward = 1.0f / (1.0f + yd * yd);
    const V3 rp{balloon.x - pre_pos.x, balloon.y - pre_pos.y, balloon.z - pre_pos.z};
    const float check = sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
    const float guidance = 30.0f * (sqrtf(rp.x * rp.x + rp.y * rp.y + rp.z * rp.z) - check);
    const float ups_z = (2.0f * (s.q.w * s.q.w) - 1.0f) + 0.0f + s.q.z * s.q.z * 2.0f;
    const float hu = (ups_z + 1.0f) / 2.0f;
    const float ups_reward = 0.5f * (hu * hu);
    const bool hit = check < 0.1f;
    const float hit_reward = hit ? 800.0f : 0.0f;
    float a2 = 0.0f, d2 = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) { a2 += a[i] * a[i]; const float d = a[i] - pre_a[i]; d2 += d * d; }
    const float effort = 0.1f * expf(-a2);
    const float smooth = 0.1f * expf(-sqrtf(d2));
    const float reward = guidance + yaw_reward + hit_reward + smooth + ups_reward + effort;
    int done = (s.progress >= P.max_episode_length - 1) ? 1 : 0;
    done = (a[A - 1] < -1.0f) ? 1 : done;
    done = (a[A - 1] > 1.0f) ? 1 : done;
    done = (rel.x < -0.2f) ? 1 : done;
    done = (s.v.x < 0.0f) ? 1 : done;
    done = (check > 4.0f) ? 1 : done;
    done = (s.p.z < 0.5f) ? 1 : done;
    done = (s.p.z > 1.5f) ? 1 : done;
    done = hit ? 1 : done;
    done = collided ? 1 : done;                                  // reset_on_collision, balloon.py:133-135
    o.terms[0] = guidance; o.terms[1] = hit_reward; o.terms[2] = smooth; o.terms[3] = effort; o.terms[4] = ups_reward;
    o.terms[5] = reward; o.terms[6] = 0.0f; o.terms[7] = 0.0f;
    o.rew = reward;
    o.done = done;
#pragma unroll
    for (int i = 0; i < A; ++i) pre_a[i] = a[i];
    pre_pos = s.p;
    s.was_reset = done;
}

// ------------------------------------------------------------------------------------------------------------ Avoid
// The thrown cube between two steps (build-defined: PhysX in the reference): ballistic semi-implicit Euler, inelastic landing.
AG_HD void avoid_object_step(V3& p, V3& v, float dt) {
    const bool parked = (p.x == -999.0f);
    const bool resting = (p.z <= kCubeHalf) && (fabsf(v.x) + fabsf(v.y) + fabsf(v.z) == 0.0f);
    if (parked || resting) return;
    v.z = v.z - kGrav * dt;
    p = V3{p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt};
    if (p.z <= kCubeHalf) { p.z = kCubeHalf; v = V3{0.0f, 0.0f, 0.0f}; }
}

AG_HD float point_box_distance(V3 p, V3 c, float half) {
    const float dx = fmaxf(fabsf(p.x - c.x) - half, 0.0f);
    const float dy = fmaxf(fabsf(p.y - c.y) - half, 0.0f);
    const float dz = fmaxf(fabsf(p.z - c.z) - half, 0.0f);
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

// reset_idx incl. calculate_object_velocity, avoid.py:58-163.  u[11]: mask | theta | aim xyz | root x y | root z | euler x y z
AG_HD void avoid_reset(EnvState& s, CtlState& c, V3& obj_p, V3& obj_v, V3& pre_pos, float* pre_a, int num_actions, const float* u) {
    if (u[0] < 0.8f) {
        const float theta = (kPi / 6.0f) * (2.0f * u[1] + -1.0f);
        obj_p = V3{4.2f * cosf(theta), 4.2f * sinf(theta), 1.4f};
        const V3 aim{0.3f * (2.0f * u[2] + -1.0f) + 0.0f, 0.3f * (2.0f * u[3] + -1.0f) + 0.0f, 0.3f * (2.0f * u[4] + -1.0f) + 1.0f};
        const float dx = aim.x - obj_p.x, dy = aim.y - obj_p.y;
        const float dist = sqrtf(dx * dx + dy * dy);
        const float ux = dx / dist, uy = dy / dist;
        const float t = dist / 4.5f;
        obj_v = V3{ux * 4.5f, uy * 4.5f, (aim.z - obj_p.z + 0.5f * kGrav * (t * t)) / t};
    } else {
        obj_p = V3{-999.0f, -999.0f, 0.0f};
        obj_v = V3{0.0f, 0.0f, 0.0f};
    }
    s.p = V3{0.2f * (2.0f * u[5] + -1.0f) + 0.0f, 0.2f * (2.0f * u[6] + -1.0f) + 0.0f, 0.2f * (2.0f * u[7] + -1.0f) + 1.0f};
    const float a0 = 0.01f * (kTwoPi * u[8] + -kPi), a1 = 0.01f * (kTwoPi * u[9] + -kPi), a2 = 0.05f * (kTwoPi * u[10] + -kPi);
    const float cx = cosf(0.5f * a0), sx = sinf(0.5f * a0);
    const float cy = cosf(0.5f * a1), sy = sinf(0.5f * a1);
    const float cz = cosf(0.5f * a2), sz = sinf(0.5f * a2);
    s.q.w = cx * cy * cz - sx * sy * sz;
    s.q.x = sx * cy * cz + cx * sy * sz;
    s.q.y = cx * sy * cz - sx * cy * sz;
    s.q.z = cx * cy * sz + sx * sy * cz;
    s.v = V3{0.0f, 0.0f, 0.0f};
    s.w = V3{0.0f, 0.0f, 0.0f};
    s.progress = 0;
    s.was_reset = 1;
    pre_pos = V3{0.0f, 0.0f, 0.0f};
    ctl_reset(c, s);
    for (int i = 0; i < num_actions; ++i) pre_a[i] = 0.0f;
}

// progress++, compute_observations, compute_reward (avoid.py:189-240,242-300); target = P.target (avoid_config.py:11)
template <int CTL>
AG_HD void avoid_post(EnvState& s, V3& pre_pos, float* pre_a, const float* raw_action, int collided, const StepParams& P,
                      float* obs, CustomOut& o) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    float a[A];
    planning_process_action<CTL>(raw_action, a);
    s.progress += 1;
    float R[9];
    quat_to_matrix_div(s.q, R);
    const float yaw = atan2f(R[3], R[0]);
    const float cy = cosf(yaw), sy = sinf(yaw);
    float L[9];
    L[0] = cy * R[0] + sy * R[3]; L[1] = cy * R[1] + sy * R[4]; L[2] = cy * R[2] + sy * R[5];
    L[3] = -sy * R[0] + cy * R[3]; L[4] = -sy * R[1] + cy * R[4]; L[5] = -sy * R[2] + cy * R[5];
    L[6] = R[6]; L[7] = R[7]; L[8] = R[8];
    obs[0] = s.p.x - P.target[9]; obs[1] = s.p.y - P.target[10]; obs[2] = s.p.z - P.target[11];
    obs[3] = atan2f(-L[5], L[8]); obs[4] = asinf(L[2]); obs[5] = atan2f(-L[1], L[0]);
    obs[6] = cy * s.v.x + sy * s.v.y; obs[7] = -sy * s.v.x + cy * s.v.y; obs[8] = s.v.z;
    obs[9] = cy * s.w.x + sy * s.w.y; obs[10] = -sy * s.w.x + cy * s.w.y; obs[11] = s.w.z;
    obs[12] = a[0]; obs[13] = a[1]; obs[14] = a[2]; obs[15] = a[3];
    // ---- compute_quadcopter_reward
    const V3 rel{P.target[9] - s.p.x, P.target[10] - s.p.y, P.target[11] - s.p.z};
    const float root_yaw = atan2f(-R[1], R[0]);
    const float rh = yaw_diff(P.target_yaw, root_yaw);
    const float distance = sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z + rh * rh);
    const float pd = 1.6f * distance;
    const float pose = 1.0f / (1.0f + pd * pd);
    const float ups_z = (2.0f * (s.q.w * s.q.w) - 1.0f) + 0.0f + s.q.z * s.q.z * 2.0f;
    const float hu = (ups_z + 1.0f) / 2.0f;
    const float ups_reward = hu * hu;
    const float spinnage = s.w.z * s.w.z;
    const float spin = 1.0f / (1.0f + spinnage * spinnage);
    float a2 = 0.0f, d2 = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) a2 += a[i] * a[i];
#pragma unroll
    for (int i = 0; i < A - 1; ++i) { const float d = a[i] - pre_a[i]; d2 += d * d; }
    const float effort = 0.1f * expf(-a2);
    const float thrust_reward = 0.05f * (1.0f - fabsf(0.1533f - a[A - 1]));
    const float smooth = 0.1f * expf(-sqrtf(d2));
    const float alive = collided ? -500.0f : 0.5f;
    const float reward = pose + pose * (ups_reward + spin) + effort + smooth + thrust_reward + alive;
    int done = (s.progress >= P.max_episode_length - 1) ? 1 : 0;
    done = (s.p.z < 0.3f) ? 1 : done;
    done = (s.p.z > 1.7f) ? 1 : done;
    done = (sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z) > 2.0f) ? 1 : done;
    done = (ups_z < 0.0f) ? 1 : done;
    done = collided ? 1 : done;                                  // reset_on_collision, avoid.py:191-193
    o.terms[0] = pose; o.terms[1] = ups_reward; o.terms[2] = spin; o.terms[3] = effort; o.terms[4] = smooth;
    o.terms[5] = thrust_reward; o.terms[6] = alive; o.terms[7] = reward;
    o.rew = reward;
    o.done = done;
#pragma unroll
    for (int i = 0; i < A; ++i) pre_a[i] = a[i];
    pre_pos = s.p;
    s.was_reset = done;
}

// smallest t > 0 where o + t d enters the axis-aligned cube (exit parameter for a ray starting inside); kInf = miss
AG_HD float ray_aabb(V3 o, V3 d, V3 c, float half) {
    float tmin = -kInf, tmax = kInf;
    bool ok = true;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, cc[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (fabsf(dd[ax]) <= 1e-12f) {
            ok = ok && (oo[ax] >= cc[ax] - half) && (oo[ax] <= cc[ax] + half);
        } else {
            const float inv = 1.0f / dd[ax];
            const float t0 = (cc[ax] - half - oo[ax]) * inv, t1 = (cc[ax] + half - oo[ax]) * inv;
            tmin = fmaxf(tmin, fminf(t0, t1));
            tmax = fminf(tmax, fmaxf(t0, t1));
        }
    }
    if (!(ok && tmax >= tmin && tmax > 0.0f)) return kInf;
    return tmin > 0.0f ? tmin : tmax;
}

// z-depth of one pixel of Avoid's scene: the cube and the ground plane
AG_HD float depth_pixel_box(const Camera& cam, V3 d, V3 box) {
    float t = ray_aabb(cam.o, d, box, kCubeHalf);
    if (d.z < -1e-12f) {
        const float tg = -cam.o.z * fast_rcp(d.z);
        if (tg > 0.0f) t = fminf(t, tg);
    }
    return t <= kCamFar ? t : kInf;
}

}  // namespace ag
