// TEST AID (never part of the product): compiles airgym_amd/csrc/env_math.hpp - the exact source the
// gfx950 kernel inlines - with g++ so that `pytest -m "not gpu"` can check the kernel arithmetic
// against the oracle on the GPU-less build box.  The shipped library has no CPU path; this harness is
// built into tests/host_harness/_build/ and loaded only by tests/test_host_harness.py.
#include <stdint.h>
#include <string.h>

#include "../../airgym_amd/csrc/env_math.hpp"
#include "../../airgym_amd/csrc/planning_math.hpp"

using namespace ag;

namespace {

struct Arrays {
    int n;
    float* root_states;   // [n,13]
    float* ctl_state;     // [n,12]
    float* pre_actions;   // [n,A]
    int32_t* progress;    // [n]
    int32_t* was_reset;   // [n]
};

void load(const Arrays& a, int i, int A, EnvState& s, CtlState& c, float* pre_a) {
    const float* r = a.root_states + (size_t)i * 13;
    s.p = V3{r[0], r[1], r[2]};
    s.q = Q4{r[3], r[4], r[5], r[6]};
    s.v = V3{r[7], r[8], r[9]};
    s.w = V3{r[10], r[11], r[12]};
    s.progress = a.progress[i];
    s.was_reset = a.was_reset[i];
    const float* cs = a.ctl_state + (size_t)i * 12;
    for (int j = 0; j < 3; ++j) {
        c.rate_int[j] = cs[j]; c.prev_rate[j] = cs[3 + j]; c.vel_int[j] = cs[6 + j]; c.prev_vel[j] = cs[9 + j];
    }
    for (int j = 0; j < A; ++j) pre_a[j] = a.pre_actions[(size_t)i * A + j];
}

void store(const Arrays& a, int i, int A, const EnvState& s, const CtlState& c, const float* pre_a) {
    float* r = a.root_states + (size_t)i * 13;
    r[0] = s.p.x; r[1] = s.p.y; r[2] = s.p.z;
    r[3] = s.q.x; r[4] = s.q.y; r[5] = s.q.z; r[6] = s.q.w;
    r[7] = s.v.x; r[8] = s.v.y; r[9] = s.v.z;
    r[10] = s.w.x; r[11] = s.w.y; r[12] = s.w.z;
    a.progress[i] = s.progress;
    a.was_reset[i] = s.was_reset;
    float* cs = a.ctl_state + (size_t)i * 12;
    for (int j = 0; j < 3; ++j) {
        cs[j] = c.rate_int[j]; cs[3 + j] = c.prev_rate[j]; cs[6 + j] = c.vel_int[j]; cs[9 + j] = c.prev_vel[j];
    }
    for (int j = 0; j < A; ++j) a.pre_actions[(size_t)i * A + j] = pre_a[j];
}

template <int TASK, int CTL>
void run_step(const Arrays& a, const StepParams& P, const float* actions, const float* noise, const float* uniforms,
              float* obs, float* rew, int32_t* done, int32_t* timeout, float* terms, float* cmd) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int NOBS = TaskTraits<TASK>::kNumObs;
    for (int i = 0; i < a.n; ++i) {
        EnvState s;
        CtlState c;
        memset(&c, 0, sizeof(c));
        float pre_a[A];
        load(a, i, A, s, c, pre_a);
        StepOut o;
        float ob[NOBS];
        const uint32_t eg = P.env_id_offset + (uint32_t)i;
        if (noise)
            env_step<TASK, CTL, true>(s, c, pre_a, actions + (size_t)i * A, P, eg, noise + (size_t)i * 18,
                                      uniforms + (size_t)i * 12, ob, o);
        else
            env_step<TASK, CTL, false>(s, c, pre_a, actions + (size_t)i * A, P, eg, nullptr, nullptr, ob, o);
        store(a, i, A, s, c, pre_a);
        for (int j = 0; j < NOBS; ++j) obs[(size_t)i * NOBS + j] = ob[j];
        rew[i] = o.rew;
        done[i] = o.done;
        timeout[i] = o.timeout;
        for (int t = 0; t < 9; ++t) terms[(size_t)i * 9 + t] = o.terms[t];
        for (int t = 0; t < 4; ++t) cmd[(size_t)i * 4 + t] = o.cmd[t];
    }
}

typedef void (*StepFn)(const Arrays&, const StepParams&, const float*, const float*, const float*, float*, float*,
                       int32_t*, int32_t*, float*, float*);

#define ROW(T) {run_step<T, 0>, run_step<T, 1>, run_step<T, 2>, run_step<T, 3>, run_step<T, 4>}
const StepFn kTable[2][5] = {ROW(0), ROW(1)};

}  // namespace

extern "C" {

int agh_step(int task, int ctl, int n, double dt, int max_len, const float* target18, uint64_t seed, uint32_t tick,
             uint32_t env_id_offset, int noise_off, float* root_states, float* ctl_state, float* pre_actions,
             int32_t* progress, int32_t* was_reset, const float* actions, const float* noise, const float* uniforms,
             float* obs, float* rew, int32_t* done, int32_t* timeout, float* terms, float* cmd) {
    if (task < 0 || task > 1 || ctl < 0 || ctl > 4) return -1;
    StepParams P = make_step_params(task, dt, max_len, target18, seed, env_id_offset, noise_off != 0);
    P.tick = tick;
    Arrays a{n, root_states, ctl_state, pre_actions, progress, was_reset};
    kTable[task][ctl](a, P, actions, noise, uniforms, obs, rew, done, timeout, terms, cmd);
    return 0;
}

int agh_reset_all(int task, int num_actions, int n, double dt, int max_len, const float* target18, uint64_t seed,
                  uint32_t tick, uint32_t env_id_offset, float* root_states, float* ctl_state, float* pre_actions,
                  int32_t* progress, int32_t* was_reset) {
    StepParams P = make_step_params(task, dt, max_len, target18, seed, env_id_offset, false);
    P.tick = tick;
    Arrays a{n, root_states, ctl_state, pre_actions, progress, was_reset};
    for (int i = 0; i < n; ++i) {
        EnvState s;
        CtlState c;
        float pre_a[5];
        env_reset(s, c, pre_a, num_actions, P, env_id_offset + (uint32_t)i);
        store(a, i, num_actions, s, c, pre_a);
    }
    return 0;
}

// AG_FLAG_STAGGER_PHASE: the initial progress reset_all_kernel gives env i (env_math.hpp stagger_progress)
int agh_stagger_progress(int n, int max_len, uint64_t seed, uint32_t tick, uint32_t env_id_offset, int32_t* progress) {
    float target18[18] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    StepParams P = make_step_params(0, 0.01, max_len, target18, seed, env_id_offset, false, false, true);
    P.tick = tick;
    for (int i = 0; i < n; ++i) progress[i] = stagger_progress(P, env_id_offset + (uint32_t)i);
    return 0;
}

// ---- Planning: the same per-env functions the gfx950 kernels inline (planning_math.hpp)
int agh_plan_render(const float* pos3, const float* quat4, const float* obst /*[40,4]*/, const float* table,
                    const float* goal3, float* out_hw /*[H][W] raw z-depth, inf = no hit*/) {
    Cyl cyl[kNumObst];
    for (int j = 0; j < kNumObst; ++j)
        cyl[j] = world_cylinder(obst[4 * j], obst[4 * j + 1], obst[4 * j + 2], table + ((int)obst[4 * j + 3] % kNumVariants) * 8);
    const Camera cam = make_camera(V3{pos3[0], pos3[1], pos3[2]}, Q4{quat4[0], quat4[1], quat4[2], quat4[3]});
    const V3 goal{goal3[0], goal3[1], goal3[2]};
    for (int v = 0; v < kCamH; ++v)
        for (int u = 0; u < kCamW; ++u) out_hw[v * kCamW + u] = depth_pixel(cam, pixel_direction(cam, u, v), cyl, kNumObst, goal);
    return 0;
}

}  // extern "C"

template <int CTL>
static void plan_step_t(int n, const StepParams& P, float* rs, float* cs, float* pa, int32_t* progress, int32_t* was_reset,
                        const float* actions, float* obst, float* goal, float* extra, const float* table,
                        const float* ext_u, float* obs, float* rew, int32_t* done, float* terms, float* coll) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    Arrays a{n, rs, cs, pa, progress, was_reset};
    for (int i = 0; i < n; ++i) {
        EnvState s;
        CtlState c;
        float pre_a[A];
        load(a, i, A, s, c, pre_a);
        planning_physics<CTL>(s, c, actions + (size_t)i * A, P);
        PlanExtra x;
        x.goal = V3{goal[3 * i], goal[3 * i + 1], goal[3 * i + 2]};
        x.pre_pos = V3{extra[5 * i], extra[5 * i + 1], extra[5 * i + 2]};
        x.esdf = extra[5 * i + 3];
        x.prev_related_dist = extra[5 * i + 4];
        float* ob = obst + (size_t)i * kNumObst * 4;
        int collided = (s.p.z <= kRobotRadius) ? 1 : 0;
        for (int j = 0; j < kNumObst; ++j) {
            const Cyl w = world_cylinder(ob[4 * j], ob[4 * j + 1], ob[4 * j + 2], table + ((int)ob[4 * j + 3] % kNumVariants) * 8);
            if (point_cylinder_distance(s.p, w) <= kRobotRadius) collided = 1;
        }
        PlanOut o;
        float ob16[kPlanNumObs];
        planning_post<CTL>(s, x, pre_a, actions + (size_t)i * A, collided, P, ob16, o);
        if (o.done) {
            float u[124];
            if (ext_u) for (int j = 0; j < kPlanResetUniforms; ++j) u[j] = ext_u[(size_t)i * kPlanResetUniforms + j];
            else planning_reset_uniforms(P, P.env_id_offset + (uint32_t)i, u);
            planning_reset(s, c, x, pre_a, A, u, ob, 4);
        }
        x.prev_related_dist = o.related_dist;
        store(a, i, A, s, c, pre_a);
        goal[3 * i] = x.goal.x; goal[3 * i + 1] = x.goal.y; goal[3 * i + 2] = x.goal.z;
        extra[5 * i] = x.pre_pos.x; extra[5 * i + 1] = x.pre_pos.y; extra[5 * i + 2] = x.pre_pos.z;
        extra[5 * i + 3] = x.esdf; extra[5 * i + 4] = x.prev_related_dist;
        for (int j = 0; j < kPlanNumObs; ++j) obs[(size_t)i * kPlanNumObs + j] = ob16[j];
        rew[i] = o.rew; done[i] = o.done; coll[i] = (float)collided;
        for (int t = 0; t < kPlanNumTerms; ++t) terms[(size_t)i * kPlanNumTerms + t] = o.terms[t];
    }
}

extern "C" {

int agh_plan_step(int ctl, int n, double dt, int max_len, uint64_t seed, uint32_t tick, uint32_t env_id_offset,
                  float* rs, float* cs, float* pa, int32_t* progress, int32_t* was_reset, const float* actions,
                  float* obst, float* goal, float* extra, const float* table, const float* ext_u, float* obs, float* rew,
                  int32_t* done, float* terms, float* coll) {
    float target[18] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    StepParams P = make_step_params(2, dt, max_len, target, seed, env_id_offset, true);
    P.tick = tick;
    switch (ctl) {
        case CTL_POS: plan_step_t<CTL_POS>(n, P, rs, cs, pa, progress, was_reset, actions, obst, goal, extra, table, ext_u, obs, rew, done, terms, coll); break;
        case CTL_VEL: plan_step_t<CTL_VEL>(n, P, rs, cs, pa, progress, was_reset, actions, obst, goal, extra, table, ext_u, obs, rew, done, terms, coll); break;
        case CTL_RATE: plan_step_t<CTL_RATE>(n, P, rs, cs, pa, progress, was_reset, actions, obst, goal, extra, table, ext_u, obs, rew, done, terms, coll); break;
        case CTL_PROP: plan_step_t<CTL_PROP>(n, P, rs, cs, pa, progress, was_reset, actions, obst, goal, extra, table, ext_u, obs, rew, done, terms, coll); break;
        default: return -1;
    }
    return 0;
}

void agh_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    const U4 r = philox4x32_10(c0, c1, c2, c3, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

}  // extern "C"

// ---- Balloon / Avoid (planning_math.hpp): post-physics half and reset, as custom_step_kernel calls them
template <int TASK, int CTL>
static void custom_post_t(int n, const StepParams& P, float* rs, float* pa, int32_t* progress, const float* actions, float* goal,
                          float* objvel, float* prepos, const float* coll, const float* noise, const float* ext_u, float* obs,
                          float* rew, int32_t* done, float* terms) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int NOBS = (TASK == TASK_BALLOON) ? kBalloonNumObs : kAvoidNumObs;
    constexpr int NU = (TASK == TASK_BALLOON) ? kBalloonResetUniforms : kAvoidResetUniforms;
    float cs[12] = {0};
    int32_t wr = 0;
    for (int i = 0; i < n; ++i) {
        Arrays a{1, rs + (size_t)i * 13, cs, pa + (size_t)i * A, progress + i, &wr};
        EnvState s;
        CtlState c;
        float pre_a[A];
        load(a, 0, A, s, c, pre_a);
        V3 tgt{goal[3 * i], goal[3 * i + 1], goal[3 * i + 2]}, ov{objvel[3 * i], objvel[3 * i + 1], objvel[3 * i + 2]};
        V3 pp{prepos[3 * i], prepos[3 * i + 1], prepos[3 * i + 2]};
        float ob[18];
        CustomOut o;
        const int collided = coll[i] != 0.0f;
        if (TASK == TASK_BALLOON) balloon_post<CTL>(s, tgt, pp, pre_a, actions + (size_t)i * A, collided, noise + (size_t)i * 18, P, ob, o);
        else avoid_post<CTL>(s, pp, pre_a, actions + (size_t)i * A, collided, P, ob, o);
        if (o.done && ext_u) {
            if (TASK == TASK_BALLOON) balloon_reset(s, c, tgt, pp, pre_a, A, ext_u + (size_t)i * NU);
            else avoid_reset(s, c, tgt, ov, pp, pre_a, A, ext_u + (size_t)i * NU);
        }
        store(a, 0, A, s, c, pre_a);
        goal[3 * i] = tgt.x; goal[3 * i + 1] = tgt.y; goal[3 * i + 2] = tgt.z;
        objvel[3 * i] = ov.x; objvel[3 * i + 1] = ov.y; objvel[3 * i + 2] = ov.z;
        prepos[3 * i] = pp.x; prepos[3 * i + 1] = pp.y; prepos[3 * i + 2] = pp.z;
        for (int j = 0; j < NOBS; ++j) obs[(size_t)i * NOBS + j] = ob[j];
        rew[i] = o.rew; done[i] = o.done;
        for (int t = 0; t < kCustomMaxTerms; ++t) terms[(size_t)i * kCustomMaxTerms + t] = o.terms[t];
    }
}

extern "C" int agh_custom_post(int task, int ctl, int n, int max_len, const float* target18, float* rs, float* pa,
                               int32_t* progress, const float* actions, float* goal, float* objvel, float* prepos,
                               const float* coll, const float* noise, const float* ext_u, float* obs, float* rew,
                               int32_t* done, float* terms) {
    StepParams P = make_step_params(0, 0.01, max_len, target18, 0, 0, false);
#define AGH_C(T, C) custom_post_t<T, C>(n, P, rs, pa, progress, actions, goal, objvel, prepos, coll, noise, ext_u, obs, rew, done, terms)
    if (task == TASK_BALLOON) {
        if (ctl == CTL_VEL) AGH_C(TASK_BALLOON, CTL_VEL); else if (ctl == CTL_RATE) AGH_C(TASK_BALLOON, CTL_RATE);
        else if (ctl == CTL_ATTI) AGH_C(TASK_BALLOON, CTL_ATTI); else return -1;
    } else if (task == TASK_AVOID) {
        if (ctl == CTL_VEL) AGH_C(TASK_AVOID, CTL_VEL); else if (ctl == CTL_RATE) AGH_C(TASK_AVOID, CTL_RATE); else return -1;
    } else return -1;
#undef AGH_C
    return 0;
}

extern "C" float agh_ray_aabb(const float* o3, const float* d3, const float* c3, float half) {
    return ray_aabb(V3{o3[0], o3[1], o3[2]}, V3{d3[0], d3[1], d3[2]}, V3{c3[0], c3[1], c3[2]}, half);
}
