// airgym_hip.hip - C ABI (include/airgym_hip.h) over the gfx950 step kernels.
//
// Host side of the hot path: arena carving, parameter block, launcher dispatch and the small
// non-hot kernels (reset-all, state pack/unpack, reset-id compaction).  No CPU execution path:
// every entry point that computes launches a HIP kernel on the handle's device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>

#include "../../include/airgym_hip.h"
#include "handle.hpp"
#include "kernel_args.hpp"
#include "planning_math.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define AG_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(AG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

constexpr int kPad = 256;

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

using Layout = AgLayout;

Layout make_layout(int n, int num_obs, bool terms, int task = 0) {
    const size_t np = align_up((size_t)n, kPad);
    Layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    for (int j = 0; j < 4; ++j) L.S[j] = take(np * 16);
    for (int j = 0; j < 4; ++j) L.C[j] = take(np * 16);
    L.PA = take(np * 16);
    L.PA4 = take(np * 4);
    L.obs = take(np * num_obs * 4);
    L.rew = take(np * 4);
    L.reset = take(np * 8);
    L.timeout = take(np);
    L.mask = take(np / 64 * 8);
    L.reset_ids = take(np * 4);
    L.reset_count = take(256);
    L.tick = take(256);
    const int nterms = (task == AG_TASK_PLANNING) ? 11 : (task == AG_TASK_AVOID ? 8 : (task == AG_TASK_BALLOON ? 6 : 9));
    for (int t = 0; t < 11; ++t) L.terms[t] = (terms && t < nterms) ? take(np * 4) : 0;
    L.cmd = (terms && task < AG_TASK_PLANNING) ? take(np * 16) : 0;
    L.OB = L.GOAL = L.PRP = L.image = L.collisions = L.table = 0;
    if (task >= AG_TASK_PLANNING) {      // the Customized family: Planning, Balloon, Avoid
        L.OB = take((size_t)(task == AG_TASK_PLANNING ? ag::kNumObst : 1) * np * 16);
        L.GOAL = take(np * 16);
        L.PRP = take(np * 16);
        L.collisions = take(np * 4);
        L.table = take((size_t)ag::kNumVariants * 8 * 4);
        if (task != AG_TASK_BALLOON) L.image = take((size_t)n * ag::kCamPix * 4);      // Balloon has no camera
    }
    L.total = off;
    return L;
}

const ag::EvalLauncher kEvalLaunchers[2][5] = {
    {ag::launch_eval_0_0, ag::launch_eval_0_1, ag::launch_eval_0_2, ag::launch_eval_0_3, ag::launch_eval_0_4},
    {ag::launch_eval_1_0, ag::launch_eval_1_1, ag::launch_eval_1_2, ag::launch_eval_1_3, ag::launch_eval_1_4},
};

const ag::StepLauncher kLaunchers[2][5] = {
    {ag::launch_step_0_0, ag::launch_step_0_1, ag::launch_step_0_2, ag::launch_step_0_3, ag::launch_step_0_4},
    {ag::launch_step_1_0, ag::launch_step_1_1, ag::launch_step_1_2, ag::launch_step_1_3, ag::launch_step_1_4},
};

}  // namespace

namespace {

// ------------------------------------------------------------------ small kernels
__global__ void reset_all_kernel(ag::KArgs k, int n_pad, int num_actions, int num_obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    ag::EnvState s;
    ag::CtlState c;
    float pre_a[AG_MAX_ACTIONS];
    ag::StepParams P = k.P;
    P.tick = *k.tick_in;
    if (i == 0) *k.tick_out = P.tick + 1u;
    ag::env_reset(s, c, pre_a, num_actions, P, P.env_id_offset + (uint32_t)i);
    if (P.stagger_phase) s.progress = ag::stagger_progress(P, P.env_id_offset + (uint32_t)i);   // AG_FLAG_STAGGER_PHASE
    ag::store_env(k, i, s);
    ag::store_ctl<ag::CTL_POS>(k, i, c);  // writes all four controller arrays
    k.PA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k.PA4[i] = 0.f;
    if (i < k.n) {
        k.rew[i] = 0.f;
        k.reset[i] = 1;      // base_task.py:75 reset_buf = ones; hovering.py:332
        k.timeout[i] = 0;
        if ((i & 63) == 0) k.mask[i >> 6] = 0ull;
    }
}

// reset_idx(env_ids) for a caller-chosen subset (hovering.py:310-335, tracking.py:159-192): re-randomise the listed envs with
// the counter RNG of the current tick, flag them reset (reset_buf = 1, bit in the ballot mask), clear progress / pre_actions /
// controller memory.  One thread per listed id; ids outside [0, n) are ignored.
__global__ void reset_ids_kernel(ag::KArgs k, const int* ids, int count, int num_actions) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    ag::StepParams P = k.P;
    P.tick = *k.tick_in;
    if (j == 0) *k.tick_out = P.tick + 1u;
    if (j >= count) return;
    const int i = ids[j];
    if (i < 0 || i >= k.n) return;
    ag::EnvState s;
    ag::CtlState c;
    float pre_a[AG_MAX_ACTIONS];
    ag::env_reset(s, c, pre_a, num_actions, P, P.env_id_offset + (uint32_t)i);
    ag::store_env(k, i, s);
    ag::store_ctl<ag::CTL_POS>(k, i, c);
    k.PA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k.PA4[i] = 0.f;
    k.reset[i] = 1;
    atomicOr(&k.mask[i >> 6], 1ull << (i & 63));
}

__global__ void get_state_kernel(ag::KArgs k, ag_state_view v, int num_actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k.n) return;
    ag::EnvState s;
    ag::load_env(k, i, s);
    if (v.root_states_dev) {
        float* r = v.root_states_dev + (size_t)i * 13;
        r[0] = s.p.x; r[1] = s.p.y; r[2] = s.p.z;
        r[3] = s.q.x; r[4] = s.q.y; r[5] = s.q.z; r[6] = s.q.w;
        r[7] = s.v.x; r[8] = s.v.y; r[9] = s.v.z;
        r[10] = s.w.x; r[11] = s.w.y; r[12] = s.w.z;
    }
    if (v.ctl_state_dev) {
        float* c = v.ctl_state_dev + (size_t)i * 12;
        for (int j = 0; j < 4; ++j) {
            const float4 x = k.C[j][i];
            c[3 * j] = x.x; c[3 * j + 1] = x.y; c[3 * j + 2] = x.z;
        }
    }
    if (v.pre_actions_dev) {
        const float4 p = k.PA[i];
        float* a = v.pre_actions_dev + (size_t)i * num_actions;
        a[0] = p.x; a[1] = p.y; a[2] = p.z; a[3] = p.w;
        if (num_actions == 5) a[4] = k.PA4[i];
    }
    if (v.progress_dev) v.progress_dev[i] = s.progress;
    if (v.was_reset_dev) v.was_reset_dev[i] = s.was_reset;
}

__global__ void set_state_kernel(ag::KArgs k, ag_state_view v, int num_actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k.n) return;
    ag::EnvState s;
    ag::load_env(k, i, s);
    if (v.root_states_dev) {
        const float* r = v.root_states_dev + (size_t)i * 13;
        s.p = ag::V3{r[0], r[1], r[2]};
        s.q = ag::Q4{r[3], r[4], r[5], r[6]};
        s.v = ag::V3{r[7], r[8], r[9]};
        s.w = ag::V3{r[10], r[11], r[12]};
    }
    if (v.progress_dev) s.progress = v.progress_dev[i];
    if (v.was_reset_dev) s.was_reset = v.was_reset_dev[i];
    ag::store_env(k, i, s);
    if (v.ctl_state_dev) {
        const float* c = v.ctl_state_dev + (size_t)i * 12;
        for (int j = 0; j < 4; ++j) k.C[j][i] = make_float4(c[3 * j], c[3 * j + 1], c[3 * j + 2], 0.f);
    }
    if (v.pre_actions_dev) {
        const float* a = v.pre_actions_dev + (size_t)i * num_actions;
        k.PA[i] = make_float4(a[0], a[1], a[2], a[3]);
        if (num_actions == 5) k.PA4[i] = a[4];
    }
}

// Ascending list of done env ids from the per-wavefront ballot words: the device-side equivalent
// of reset_buf.nonzero().squeeze(-1) (hovering.py:209,300).  One block; scan of popcounts in LDS.
__global__ __launch_bounds__(1024) void compact_reset_ids_kernel(const unsigned long long* mask, int n_words,
                                                                 int* ids, int* count) {
    __shared__ int scan[1024];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int w0 = 0; w0 < n_words; w0 += 1024) {
        const int w = w0 + tid;
        const unsigned long long m = (w < n_words) ? mask[w] : 0ull;
        const int cnt = __popcll(m);
        scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const int v = (tid >= off) ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int pos = base + scan[tid] - cnt;
        unsigned long long mm = m;
        while (mm) {
            const int b = __ffsll((long long)mm) - 1;
            ids[pos++] = w * 64 + b;
            mm &= mm - 1;
        }
        __syncthreads();
        if (tid == 1023) base += scan[1023];
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

__global__ void planning_get_state_kernel(ag::KArgs k, ag::PlanArgs pa, ag_planning_state_view v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k.n) return;
    if (v.obstacles_dev) {
        for (int j = 0; j < ag::kNumObst; ++j) {
            const float4 ob = pa.OB[(size_t)j * pa.n_pad + i];
            float* o = v.obstacles_dev + ((size_t)i * ag::kNumObst + j) * 4;
            o[0] = ob.x; o[1] = ob.y; o[2] = ob.z; o[3] = (float)__float_as_int(ob.w);
        }
    }
    const float4 g = pa.GOAL[i], e = pa.PRP[i];
    if (v.goal_dev) { float* o = v.goal_dev + (size_t)i * 3; o[0] = g.x; o[1] = g.y; o[2] = g.z; }
    if (v.extra_dev) { float* o = v.extra_dev + (size_t)i * 5; o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = e.w; o[4] = g.w; }
    if (v.object_vel_dev) { const float4 w = pa.OB[i]; float* o = v.object_vel_dev + (size_t)i * 3; o[0] = w.x; o[1] = w.y; o[2] = w.z; }
}

__global__ void planning_set_state_kernel(ag::KArgs k, ag::PlanArgs pa, ag_planning_state_view v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k.n) return;
    if (v.obstacles_dev) {
        for (int j = 0; j < ag::kNumObst; ++j) {
            const float* o = v.obstacles_dev + ((size_t)i * ag::kNumObst + j) * 4;
            pa.OB[(size_t)j * pa.n_pad + i] = make_float4(o[0], o[1], o[2], __int_as_float((int)o[3]));
        }
    }
    float4 g = pa.GOAL[i], e = pa.PRP[i];
    if (v.goal_dev) { const float* o = v.goal_dev + (size_t)i * 3; g.x = o[0]; g.y = o[1]; g.z = o[2]; }
    if (v.extra_dev) { const float* o = v.extra_dev + (size_t)i * 5; e = make_float4(o[0], o[1], o[2], o[3]); g.w = o[4]; }
    pa.GOAL[i] = g;
    pa.PRP[i] = e;
    if (v.object_vel_dev) { const float* o = v.object_vel_dev + (size_t)i * 3; pa.OB[i] = make_float4(o[0], o[1], o[2], 0.f); }
}

void fill_params(ag_env* h) {
    h->k.P = ag::make_step_params(h->cfg.task, h->cfg.dt, h->cfg.max_episode_length, h->cfg.target_state, h->cfg.seed,
                                  h->cfg.env_id_offset, (h->cfg.flags & AG_FLAG_OBS_NOISE_OFF) != 0,
                                  (h->cfg.flags & AG_FLAG_FIX_TIME_OUTS) != 0, (h->cfg.flags & AG_FLAG_STAGGER_PHASE) != 0);
}

int validate(const ag_config* cfg) {
    if (!cfg) return fail(AG_ERR_INVALID_ARG, "cfg is NULL");
    if (cfg->struct_size != sizeof(ag_config))
        return fail(AG_ERR_INVALID_ARG, "ag_config.struct_size mismatch (ABI): got " + std::to_string(cfg->struct_size) +
                                            ", expected " + std::to_string(sizeof(ag_config)));
    if (cfg->task < AG_TASK_HOVERING || cfg->task > AG_TASK_AVOID)
        return fail(AG_ERR_UNKNOWN_TASK, "Task with id " + std::to_string(cfg->task) + " was not registered");
    if (cfg->ctl_mode < AG_CTL_POS || cfg->ctl_mode > AG_CTL_PROP)
        return fail(AG_ERR_UNKNOWN_CTL, "unknown ctl_mode " + std::to_string(cfg->ctl_mode) + " (expected pos|vel|atti|rate|prop)");
    if ((cfg->flags & AG_FLAG_STAGGER_PHASE) && cfg->task != AG_TASK_HOVERING)
        return fail(AG_ERR_UNSUPPORTED, "AG_FLAG_STAGGER_PHASE: Hovering only (Tracking's reference point is a function of the "
                                        "progress counter: an env started mid-trajectory is out of bounds at once)");
    if ((cfg->task == AG_TASK_PLANNING || cfg->task == AG_TASK_AVOID) && cfg->ctl_mode == AG_CTL_ATTI)
        return fail(AG_ERR_UNSUPPORTED, "planning / avoid observations hold 4 action values (planning.py:214, avoid.py:232: the "
                                        "reference's `obs_buf[..., 12:16] = actions_local` raises on atti's [N,5] actions): "
                                        "ctl_mode atti is not supported for these tasks");
    if (cfg->num_envs <= 0) return fail(AG_ERR_INVALID_ARG, "num_envs must be > 0");
    if (!(cfg->dt > 0.0)) return fail(AG_ERR_INVALID_ARG, "dt must be > 0");
    return AG_OK;
}

int ensure_device(ag_env* h) {
    int cur = -1;
    AG_HIP_CHECK(hipGetDevice(&cur));
    if (cur != h->cfg.device) AG_HIP_CHECK(hipSetDevice(h->cfg.device));
    return AG_OK;
}

void bind_tick(ag_env* h, ag::KArgs& k) {
    uint32_t* slots = (uint32_t*)(h->arena + h->L.tick);
    k.tick_in = slots + h->parity;
    k.tick_out = slots + (h->parity ^ 1);
    h->parity ^= 1;
    h->tick += 1;
}

int do_step(ag_env* h, const float* actions, float* obs_out, float* rew_out, int64_t* reset_out,
            const float* noise, const float* uniforms, void* stream, uint8_t* done_u8 = nullptr,
            float* term_sums = nullptr, bool rollout_form = false, const ag::TailArgs* tail = nullptr, int num_steps = 1,
            uint8_t* timeout_steps = nullptr, bool force_multi = false) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (!actions && !tail) return fail(AG_ERR_INVALID_ARG, "actions_dev is NULL");
    if (actions && h->num_actions == 4 && ((uintptr_t)actions & 15)) return fail(AG_ERR_INVALID_ARG, "actions_dev must be 16-byte aligned");
    if (obs_out && ((uintptr_t)obs_out & 15)) return fail(AG_ERR_INVALID_ARG, "obs_out_dev must be 16-byte aligned");
    if (h->cfg.task < AG_TASK_PLANNING && (noise == nullptr) != (uniforms == nullptr))
        return fail(AG_ERR_INVALID_ARG, "noise_dev and reset_uniforms_dev must be given together");
    int rc = ensure_device(h);
    if (rc) return rc;
    ag::KArgs k = h->k;
    k.actions = actions;
    if (obs_out) k.obs = obs_out;
    if (rew_out) k.rew = rew_out;
    if (reset_out) k.reset = (long long*)reset_out;
    if (h->cfg.task >= AG_TASK_PLANNING) {
        const int task = h->cfg.task;
        if (task == AG_TASK_PLANNING && !h->table_set) return fail(AG_ERR_INVALID_ARG, "planning: call ag_planning_set_obstacle_table first");
        if (noise != nullptr && task != AG_TASK_BALLOON) return fail(AG_ERR_UNSUPPORTED, "planning / avoid: use ag_planning_step_with_uniforms");
        ag::PlanArgs pa = h->pa;
        pa.ext_uniforms = uniforms;
        k.ext_noise = noise;            // Balloon parity mode: [n,18] observation noise
        pa.debug_skip = h->force_render >> 1;      // non-zero in the experiments build only
        h->counter += 1;
        // cam_dt / dt = 4 (planning.py:153-156, avoid.py:181-185); Balloon has no onboard camera (balloon_config.py:52)
        const bool render = (task != AG_TASK_BALLOON) && (h->force_render || (h->counter % 4 == 0));
        h->force_render = 0;
        h->last_rendered = render ? 1 : 0;
        // every kernel of this step reads the same tick; only the last one publishes tick + 1
        uint32_t* slots = (uint32_t*)(h->arena + h->L.tick);
        k.tick_in = slots + h->parity;
        k.tick_out = slots + (h->parity ^ 1);
        const hipStream_t st = (hipStream_t)stream;
        auto phase = [&](int ph) {
            return task == AG_TASK_PLANNING ? ag::launch_planning_step(k, pa, h->cfg.ctl_mode, ph, st)
                                            : ag::launch_custom_step(k, pa, task, h->cfg.ctl_mode, ph, st);
        };
        hipError_t e;
        if (render) {
            e = phase(1);
            if (e == hipSuccess) e = (task == AG_TASK_PLANNING) ? ag::launch_planning_render(k, pa, st) : ag::launch_avoid_render(k, pa, st);
            if (e == hipSuccess) e = phase(2);
        } else {
            e = phase(0);
        }
        if (e != hipSuccess) return fail(AG_ERR_HIP, std::string("planning / balloon / avoid step launch: ") + hipGetErrorString(e));
        h->parity ^= 1;
        h->tick += 1;
        return AG_OK;
    }
    k.ext_noise = noise;
    k.ext_uniforms = uniforms;
    if (rollout_form) {   // u8 done flags, per-tile reward-term sums, no per-env term / cmd arrays
        k.reset_u8 = done_u8;
        k.term_sums = term_sums;
        k.cmd = nullptr;
    }
    k.num_steps = num_steps;
    k.force_multi = force_multi ? 1 : 0;
    k.timeout_steps = timeout_steps;
    bind_tick(h, k);
    h->tick += (uint64_t)(num_steps - 1);      // the launch advances the device tick by num_steps
    hipError_t e = kLaunchers[h->cfg.task][h->cfg.ctl_mode](k, tail, (hipStream_t)stream);
    if (e != hipSuccess) return fail(AG_ERR_HIP, std::string("step kernel launch: ") + hipGetErrorString(e));
    return AG_OK;
}

}  // namespace

extern "C" {

int ag_version(void) { return AG_VERSION; }

const char* ag_last_error(void) { return g_last_error.c_str(); }

int ag_num_obs(int task) {
    if (task == AG_TASK_HOVERING) return 18;  // hovering_config.py:14
    if (task == AG_TASK_TRACKING) return 48;  // tracking_config.py:13
    if (task == AG_TASK_PLANNING) return 16;  // planning_config.py:13
    if (task == AG_TASK_BALLOON) return 18;   // balloon_config.py:13
    if (task == AG_TASK_AVOID) return 16;     // avoid_config.py:13
    return fail(AG_ERR_UNKNOWN_TASK, "unknown task");
}

int ag_num_actions(int ctl_mode) {
    if (ctl_mode < AG_CTL_POS || ctl_mode > AG_CTL_PROP) return fail(AG_ERR_UNKNOWN_CTL, "unknown ctl_mode");
    return ctl_mode == AG_CTL_ATTI ? 5 : 4;  // hovering.py:47
}

int ag_default_episode_length(int task, double dt) {
    if (!(dt > 0.0)) return fail(AG_ERR_INVALID_ARG, "dt must be > 0");
    // tracking_config.py:17, hovering_config.py:17, planning_config.py:17
    // tracking_config.py:17, hovering_config.py:17, planning_config.py:17, balloon_config.py:17, avoid_config.py:17
    const double table[5] = {24.0, 36.0, 16.0, 8.0, 6.0};
    if (task < AG_TASK_HOVERING || task > AG_TASK_AVOID) return fail(AG_ERR_UNKNOWN_TASK, "unknown task");
    const double secs = table[task];
    return (int)(secs / dt);  // int(episode_length_s / dt), hovering.py:48
}

size_t ag_arena_bytes(const ag_config* cfg) {
    if (validate(cfg) != AG_OK) return 0;
    return make_layout(cfg->num_envs, ag_num_obs(cfg->task), (cfg->flags & AG_FLAG_REWARD_TERMS) != 0, cfg->task).total;
}

int ag_create(const ag_config* cfg, void* arena_dev, ag_handle* out) {
    if (!out) return fail(AG_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int rc = validate(cfg);
    if (rc) return rc;
    if (arena_dev && ((uintptr_t)arena_dev & 255)) return fail(AG_ERR_INVALID_ARG, "arena_dev must be 256-byte aligned");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(AG_ERR_NO_DEVICE, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(AG_ERR_NO_DEVICE, "device ordinal out of range");
    AG_HIP_CHECK(hipSetDevice(cfg->device));

    ag_env* h = new (std::nothrow) ag_env();
    if (!h) return fail(AG_ERR_INVALID_ARG, "out of host memory");
    h->cfg = *cfg;
    h->num_obs = ag_num_obs(cfg->task);
    h->num_actions = ag_num_actions(cfg->ctl_mode);
    if (h->cfg.max_episode_length <= 0) h->cfg.max_episode_length = ag_default_episode_length(cfg->task, cfg->dt);
    h->n_pad = (int)align_up((size_t)cfg->num_envs, kPad);
    const bool terms = (cfg->flags & AG_FLAG_REWARD_TERMS) != 0;
    h->L = make_layout(cfg->num_envs, h->num_obs, terms, cfg->task);
    h->owns_arena = (arena_dev == nullptr);
    if (h->owns_arena) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, h->L.total);
        if (e != hipSuccess) {
            delete h;
            return fail(AG_ERR_HIP, std::string("hipMalloc(arena): ") + hipGetErrorString(e));
        }
        h->arena = (char*)p;
    } else {
        h->arena = (char*)arena_dev;
    }
    ag::KArgs& k = h->k;
    memset(&k, 0, sizeof(k));
    for (int j = 0; j < 4; ++j) k.S[j] = (float4*)(h->arena + h->L.S[j]);
    for (int j = 0; j < 4; ++j) k.C[j] = (float4*)(h->arena + h->L.C[j]);
    k.PA = (float4*)(h->arena + h->L.PA);
    k.PA4 = (float*)(h->arena + h->L.PA4);
    k.obs = (float*)(h->arena + h->L.obs);
    k.rew = (float*)(h->arena + h->L.rew);
    k.reset = (long long*)(h->arena + h->L.reset);
    k.timeout = (uint8_t*)(h->arena + h->L.timeout);
    k.mask = (unsigned long long*)(h->arena + h->L.mask);
    const bool planning = (cfg->task >= AG_TASK_PLANNING);      // the Customized family (Planning, Balloon, Avoid)
    if (terms && !planning) {
        for (int t = 0; t < 9; ++t) k.terms[t] = (float*)(h->arena + h->L.terms[t]);
        k.cmd = (float4*)(h->arena + h->L.cmd);
    }
    memset(&h->pa, 0, sizeof(h->pa));
    h->table_set = false;
    h->counter = 0;
    h->force_render = 0;
    if (planning) {
        ag::PlanArgs& pa = h->pa;
        pa.OB = (float4*)(h->arena + h->L.OB);
        pa.GOAL = (float4*)(h->arena + h->L.GOAL);
        pa.PRP = (float4*)(h->arena + h->L.PRP);
        pa.image = h->L.image ? (float*)(h->arena + h->L.image) : nullptr;
        pa.collisions = (float*)(h->arena + h->L.collisions);
        pa.table = (const float*)(h->arena + h->L.table);
        pa.n_pad = h->n_pad;
        if (terms) for (int t = 0; t < 11; ++t) pa.terms[t] = h->L.terms[t] ? (float*)(h->arena + h->L.terms[t]) : nullptr;
    }
    k.n = cfg->num_envs;
    k.num_steps = 1;
    k.force_multi = 0;
    fill_params(h);
    h->tick = 0;
    h->parity = 0;
    // valid state from the start: everything randomised and flagged reset (base_task.py:75)
    hipError_t e = hipMemsetAsync(h->arena, 0, h->L.total, 0);
    if (e == hipSuccess) {
        *out = h;
        // planning needs its obstacle table first: the caller runs ag_planning_set_obstacle_table + ag_reset_all
        rc = (cfg->task == AG_TASK_PLANNING) ? AG_OK : ag_reset_all(h, nullptr);
        if (rc == AG_OK) e = hipStreamSynchronize(0);
    }
    if (e != hipSuccess || rc != AG_OK) {
        if (e != hipSuccess) fail(AG_ERR_HIP, std::string("arena init: ") + hipGetErrorString(e));
        *out = nullptr;
        if (h->owns_arena) (void)hipFree(h->arena);
        delete h;
        return rc != AG_OK ? rc : AG_ERR_HIP;
    }
    return AG_OK;
}

int ag_destroy(ag_handle h) {
    if (!h) return AG_OK;
    if (h->owns_arena && h->arena) (void)hipFree(h->arena);
    delete h;
    return AG_OK;
}

int ag_reset_all(ag_handle h, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    int rc = ensure_device(h);
    if (rc) return rc;
    ag::KArgs k = h->k;
    if (h->cfg.task == AG_TASK_PLANNING) {
        if (!h->table_set) return fail(AG_ERR_INVALID_ARG, "planning: call ag_planning_set_obstacle_table first");
        bind_tick(h, k);
        AG_HIP_CHECK(ag::launch_planning_reset_all(k, h->pa, h->num_actions, (hipStream_t)stream));
        return AG_OK;
    }
    if (h->cfg.task > AG_TASK_PLANNING) {
        bind_tick(h, k);
        AG_HIP_CHECK(ag::launch_custom_reset_all(k, h->pa, h->cfg.task, h->num_actions, (hipStream_t)stream));
        return AG_OK;
    }
    bind_tick(h, k);
    hipLaunchKernelGGL(reset_all_kernel, dim3(h->n_pad / 256), dim3(256), 0, (hipStream_t)stream, k, h->n_pad,
                       h->num_actions, h->num_obs);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_reset_envs(ag_handle h, const int32_t* env_ids_dev, int count, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (count < 0 || (count > 0 && !env_ids_dev)) return fail(AG_ERR_INVALID_ARG, "env_ids_dev is NULL or count < 0");
    if (count == 0) return AG_OK;
    int rc = ensure_device(h);
    if (rc) return rc;
    ag::KArgs k = h->k;
    bind_tick(h, k);
    if (h->cfg.task >= AG_TASK_PLANNING) {
        if (h->cfg.task == AG_TASK_PLANNING && !h->table_set) return fail(AG_ERR_INVALID_ARG, "planning: call ag_planning_set_obstacle_table first");
        AG_HIP_CHECK(ag::launch_custom_reset_ids(k, h->pa, h->cfg.task, h->num_actions, env_ids_dev, count, (hipStream_t)stream));
        return AG_OK;
    }
    hipLaunchKernelGGL(reset_ids_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, k, env_ids_dev, count,
                       h->num_actions);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_step(ag_handle h, const float* actions_dev, void* stream) {
    return do_step(h, actions_dev, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

int ag_step_into(ag_handle h, const float* actions_dev, float* obs_out_dev, float* rew_out_dev, int64_t* reset_out_dev,
                 void* stream) {
    return do_step(h, actions_dev, obs_out_dev, rew_out_dev, reset_out_dev, nullptr, nullptr, stream);
}

int ag_term_sum_tiles(int num_envs) { return num_envs > 0 ? (num_envs + 63) / 64 : 0; }

int ag_step_rollout(ag_handle h, const float* actions_dev, float* obs_out_dev, float* rew_out_dev, uint8_t* done_out_dev,
                    float* term_sums_dev, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (h->cfg.task >= AG_TASK_PLANNING) return fail(AG_ERR_UNSUPPORTED, "ag_step_rollout: hovering / tracking handles only");
    if (!done_out_dev) return fail(AG_ERR_INVALID_ARG, "done_out_dev is NULL");
    if (term_sums_dev && ((uintptr_t)term_sums_dev & 3)) return fail(AG_ERR_INVALID_ARG, "term_sums_dev must be 4-byte aligned");
    return do_step(h, actions_dev, obs_out_dev, rew_out_dev, nullptr, nullptr, nullptr, stream, done_out_dev, term_sums_dev, true);
}

int ag_step_multi(ag_handle h, const float* actions_dev, int num_steps, float* obs_out_dev, float* rew_out_dev,
                  uint8_t* done_out_dev, uint8_t* timeout_out_dev, float* term_sums_dev, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (h->cfg.task >= AG_TASK_PLANNING) return fail(AG_ERR_UNSUPPORTED, "ag_step_multi: hovering / tracking handles only");
    if (num_steps < 1 || num_steps > 65536) return fail(AG_ERR_INVALID_ARG, "num_steps must be in [1, 65536]");
    if (!obs_out_dev || !rew_out_dev || !done_out_dev)
        return fail(AG_ERR_INVALID_ARG, "obs_out_dev / rew_out_dev / done_out_dev is NULL (each holds num_steps slices)");
    if (term_sums_dev && ((uintptr_t)term_sums_dev & 3)) return fail(AG_ERR_INVALID_ARG, "term_sums_dev must be 4-byte aligned");
    // slice kk of actions / observations starts at kk * num_envs * width floats: 16-byte alignment of every slice
    if (num_steps > 1 && (((size_t)h->cfg.num_envs * h->num_obs) & 3))
        return fail(AG_ERR_UNSUPPORTED, "ag_step_multi with num_steps > 1 needs num_envs * num_obs to be a multiple of 4 "
                                        "(16-byte aligned observation slices)");
    return do_step(h, actions_dev, obs_out_dev, rew_out_dev, nullptr, nullptr, nullptr, stream, done_out_dev, term_sums_dev, true,
                   nullptr, num_steps, timeout_out_dev, /*force_multi=*/true);
}

int ag_step_rollout_fused(ag_handle h, const ag_rollout_tail* t, float* obs_out_dev, float* rew_out_dev, uint8_t* done_out_dev,
                          float* term_sums_dev, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (h->cfg.task >= AG_TASK_PLANNING) return fail(AG_ERR_UNSUPPORTED, "ag_step_rollout_fused: hovering / tracking handles only");
    if (!t || t->struct_size != sizeof(ag_rollout_tail)) return fail(AG_ERR_INVALID_ARG, "ag_rollout_tail is NULL or struct_size mismatch (ABI)");
    if (!done_out_dev) return fail(AG_ERR_INVALID_ARG, "done_out_dev is NULL");
    if (!t->heads_dev || !t->logstd_dev || !t->counter_dev || !t->actions_dev || !t->neglogp_dev || !t->values_dev || !t->mus_dev ||
        !t->sigmas_dev || !t->shaped_dev || !t->cur_rew_dev || !t->cur_shaped_dev || !t->cur_len_dev || !t->partials_dev)
        return fail(AG_ERR_INVALID_ARG, "ag_rollout_tail: a required device pointer is NULL");
    if ((t->vmean_dev == nullptr) != (t->vvar_dev == nullptr)) return fail(AG_ERR_INVALID_ARG, "vmean_dev and vvar_dev go together");
    if (t->horizon <= 0 || t->slot < 0) return fail(AG_ERR_INVALID_ARG, "horizon must be > 0 and slot >= 0");
    if (h->num_actions == 4 && (((uintptr_t)t->actions_dev | (uintptr_t)t->mus_dev | (uintptr_t)t->sigmas_dev) & 15))
        return fail(AG_ERR_INVALID_ARG, "actions_dev / mus_dev / sigmas_dev must be 16-byte aligned");
    if (term_sums_dev && ((uintptr_t)term_sums_dev & 3)) return fail(AG_ERR_INVALID_ARG, "term_sums_dev must be 4-byte aligned");
    ag::TailArgs ta;
    ta.heads = t->heads_dev; ta.logstd = t->logstd_dev; ta.vmean = t->vmean_dev; ta.vvar = t->vvar_dev; ta.veps = t->veps;
    ta.key0 = (uint32_t)(t->seed & 0xFFFFFFFFull); ta.key1 = (uint32_t)(t->seed >> 32);
    ta.counter = (const long long*)t->counter_dev; ta.horizon = t->horizon; ta.slot = t->slot; ta.id_offset = t->id_offset;
    ta.actions = t->actions_dev; ta.neglogp = t->neglogp_dev; ta.values = t->values_dev; ta.mus = t->mus_dev; ta.sigmas = t->sigmas_dev;
    ta.scale = t->scale; ta.shift = t->shift; ta.min_val = t->min_val; ta.max_val = t->max_val; ta.log_val = t->log_val;
    ta.gamma = t->gamma; ta.bootstrap = t->bootstrap_timeouts;
    ta.shaped = t->shaped_dev; ta.cur_rew = t->cur_rew_dev; ta.cur_shaped = t->cur_shaped_dev; ta.cur_len = t->cur_len_dev;
    ta.partials = t->partials_dev;
    return do_step(h, nullptr, obs_out_dev, rew_out_dev, nullptr, nullptr, nullptr, stream, done_out_dev, term_sums_dev, true, &ta);
}

int ag_eval_obs_reward(ag_handle h, const float* processed_actions_dev, const float* cmd_thrusts_dev, const float* noise_dev,
                       void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    if (!processed_actions_dev || !cmd_thrusts_dev) return fail(AG_ERR_INVALID_ARG, "processed_actions_dev / cmd_thrusts_dev is NULL");
    if (h->cfg.task >= AG_TASK_PLANNING) return fail(AG_ERR_UNSUPPORTED, "ag_eval_obs_reward: hovering / tracking handles only (use ag_planning_eval_post)");
    int rc = ensure_device(h);
    if (rc) return rc;
    ag::KArgs k = h->k;
    k.eval_actions = processed_actions_dev;
    k.eval_cmd = cmd_thrusts_dev;
    k.ext_noise = noise_dev;
    hipError_t e = kEvalLaunchers[h->cfg.task][h->cfg.ctl_mode](k, (hipStream_t)stream);
    if (e != hipSuccess) return fail(AG_ERR_HIP, std::string("eval kernel launch: ") + hipGetErrorString(e));
    return AG_OK;
}

int ag_step_with_inputs(ag_handle h, const float* actions_dev, const float* noise_dev, const float* reset_uniforms_dev,
                        void* stream) {
    if (!noise_dev || !reset_uniforms_dev) return fail(AG_ERR_INVALID_ARG, "noise_dev / reset_uniforms_dev is NULL");
    return do_step(h, actions_dev, nullptr, nullptr, nullptr, noise_dev, reset_uniforms_dev, stream);
}

int ag_get_buffers(ag_handle h, ag_buffers* out) {
    if (!h || !out) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    out->num_envs = h->cfg.num_envs;
    out->num_obs = h->num_obs;
    out->num_actions = h->num_actions;
    out->max_episode_length = h->cfg.max_episode_length;
    out->obs_dev = h->k.obs;
    out->rew_dev = h->k.rew;
    out->reset_dev = (int64_t*)h->k.reset;
    out->timeout_dev = h->k.timeout;
    out->reset_mask_dev = (uint64_t*)h->k.mask;
    out->reset_ids_dev = (int32_t*)(h->arena + h->L.reset_ids);
    out->reset_count_dev = (int32_t*)(h->arena + h->L.reset_count);
    for (int t = 0; t < 11; ++t)
        out->reward_terms_dev[t] = (h->cfg.task >= AG_TASK_PLANNING) ? h->pa.terms[t] : (t < 9 ? h->k.terms[t] : nullptr);
    out->cmd_thrusts_dev = (float*)h->k.cmd;
    return AG_OK;
}

int ag_get_state(ag_handle h, const ag_state_view* view, void* stream) {
    if (!h || !view) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    int rc = ensure_device(h);
    if (rc) return rc;
    hipLaunchKernelGGL(get_state_kernel, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->k,
                       *view, h->num_actions);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_set_state(ag_handle h, const ag_state_view* view, void* stream) {
    if (!h || !view) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    int rc = ensure_device(h);
    if (rc) return rc;
    hipLaunchKernelGGL(set_state_kernel, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->k,
                       *view, h->num_actions);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_compact_reset_ids(ag_handle h, void* stream) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    int rc = ensure_device(h);
    if (rc) return rc;
    const int n_words = (h->cfg.num_envs + 63) / 64;
    hipLaunchKernelGGL(compact_reset_ids_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, h->k.mask, n_words,
                       (int*)(h->arena + h->L.reset_ids), (int*)(h->arena + h->L.reset_count));
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_set_target_state(ag_handle h, const float* target_state18) {
    if (!h || !target_state18) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    memcpy(h->cfg.target_state, target_state18, sizeof(h->cfg.target_state));
    fill_params(h);
    return AG_OK;
}

uint64_t ag_get_tick(ag_handle h) { return h ? h->tick : 0; }

int ag_set_tick(ag_handle h, uint64_t tick) {
    if (!h) return fail(AG_ERR_INVALID_ARG, "handle is NULL");
    int rc = ensure_device(h);
    if (rc) return rc;
    const uint32_t t[2] = {(uint32_t)tick, (uint32_t)tick};
    AG_HIP_CHECK(hipDeviceSynchronize());
    AG_HIP_CHECK(hipMemcpy(h->arena + h->L.tick, t, sizeof(t), hipMemcpyHostToDevice));
    h->tick = tick;
    return AG_OK;
}

int ag_planning_set_obstacle_table(ag_handle h, const float* table_host, int n_variants) {
    if (!h || !table_host) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    if (h->cfg.task != AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning handle");
    if (n_variants != ag::kNumVariants) return fail(AG_ERR_INVALID_ARG, "the obstacle table must have 100 rows of 8 floats");
    int rc = ensure_device(h);
    if (rc) return rc;
    AG_HIP_CHECK(hipMemcpy(h->arena + h->L.table, table_host, (size_t)n_variants * 8 * sizeof(float), hipMemcpyHostToDevice));
    h->table_set = true;
    return AG_OK;
}

int ag_planning_get_buffers(ag_handle h, ag_planning_buffers* out) {
    if (!h || !out) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    if (h->cfg.task < AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning / balloon / avoid handle");
    out->image_dev = h->pa.image;
    out->collisions_dev = h->pa.collisions;
    return AG_OK;
}

int ag_planning_get_state(ag_handle h, const ag_planning_state_view* view, void* stream) {
    if (!h || !view) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    if (h->cfg.task < AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning / balloon / avoid handle");
    if (view->obstacles_dev && h->cfg.task != AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "obstacles_dev: planning handles only");
    if (view->object_vel_dev && h->cfg.task != AG_TASK_AVOID) return fail(AG_ERR_INVALID_ARG, "object_vel_dev: avoid handles only");
    hipLaunchKernelGGL(planning_get_state_kernel, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       h->k, h->pa, *view);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_planning_set_state(ag_handle h, const ag_planning_state_view* view, void* stream) {
    if (!h || !view) return fail(AG_ERR_INVALID_ARG, "NULL argument");
    if (h->cfg.task < AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning / balloon / avoid handle");
    if (view->obstacles_dev && h->cfg.task != AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "obstacles_dev: planning handles only");
    if (view->object_vel_dev && h->cfg.task != AG_TASK_AVOID) return fail(AG_ERR_INVALID_ARG, "object_vel_dev: avoid handles only");
    hipLaunchKernelGGL(planning_set_state_kernel, dim3((h->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       h->k, h->pa, *view);
    AG_HIP_CHECK(hipGetLastError());
    return AG_OK;
}

int ag_planning_step_with_uniforms(ag_handle h, const float* actions_dev, const float* reset_uniforms_dev, void* stream) {
    if (!h || h->cfg.task < AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning / balloon / avoid handle");
    return do_step(h, actions_dev, nullptr, nullptr, nullptr, nullptr, reset_uniforms_dev, stream);
}

int ag_planning_eval_post(ag_handle h, const float* actions_dev, const float* collisions_dev, const float* noise_dev,
                          void* stream) {
    if (!h || h->cfg.task < AG_TASK_PLANNING) return fail(AG_ERR_INVALID_ARG, "not a planning / balloon / avoid handle");
    if (!actions_dev || !collisions_dev) return fail(AG_ERR_INVALID_ARG, "actions_dev / collisions_dev is NULL");
    if (h->cfg.task == AG_TASK_PLANNING && !h->table_set) return fail(AG_ERR_INVALID_ARG, "planning: call ag_planning_set_obstacle_table first");
    if (noise_dev && h->cfg.task != AG_TASK_BALLOON) return fail(AG_ERR_INVALID_ARG, "noise_dev: balloon handles only (the others add no observation noise)");
    int rc = ensure_device(h);
    if (rc) return rc;
    ag::KArgs k = h->k;
    k.actions = actions_dev;
    ag::PlanArgs pa = h->pa;
    pa.ext_collisions = collisions_dev;
    k.ext_noise = noise_dev;
    bind_tick(h, k);
    hipError_t e = (h->cfg.task == AG_TASK_PLANNING)
                       ? ag::launch_planning_step(k, pa, h->cfg.ctl_mode, 2, (hipStream_t)stream)
                       : ag::launch_custom_step(k, pa, h->cfg.task, h->cfg.ctl_mode, 2, (hipStream_t)stream);
    if (e != hipSuccess) return fail(AG_ERR_HIP, std::string("planning post-phase launch: ") + hipGetErrorString(e));
    return AG_OK;
}

int ag_planning_last_step_rendered(ag_handle h) {
    if (!h || (h->cfg.task != AG_TASK_PLANNING && h->cfg.task != AG_TASK_AVOID)) return -1;
    return h->last_rendered;
}

int ag_planning_render_now(ag_handle h) {
    if (!h || (h->cfg.task != AG_TASK_PLANNING && h->cfg.task != AG_TASK_AVOID)) return fail(AG_ERR_INVALID_ARG, "not a planning / avoid handle");
    h->force_render = 1;
    return AG_OK;
}

}  // extern "C"
