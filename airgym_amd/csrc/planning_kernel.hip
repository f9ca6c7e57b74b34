// planning_kernel.hip - gfx950 kernels of the Planning task.
//
//   planning_step_kernel<CTL, PHASE>   one env per lane.  PHASE_PHYS = pre_physics_step + simulate
//                                      (customized.py:216-298, planning.py:146-151); PHASE_POST = progress++,
//                                      collision check, observations, reward/termination, in-place reset
//                                      (planning.py:158-183); PHASE_BOTH = both in one launch (steps without a
//                                      camera render, 3 of every 4).
//   planning_render_kernel             one WORKGROUP per env: ray-cast the 212x120 depth image into LDS, then the
//                                      reference's post-processing (customized.py:399-435: clip/normalise,
//                                      additive N(0,.1), multiplicative N(1,.3), random 5x5 kernel) and the min
//                                      pixel ("esdf_dist", planning.py:162-163).  The whole image (101 760 B)
//                                      lives in the CU's 160 KB LDS: it is written to HBM exactly once.
#include <hip/hip_runtime.h>

#include "kernel_args.hpp"
#include "planning_math.hpp"

// experiments build only (build.py --experiments; tools/render_probe.py): parts of the render kernel can be skipped for timing
#ifdef AG_EXPERIMENTS
#define AG_DEBUG_SKIP(pa, bit) ((pa).debug_skip & (bit))
#else
#define AG_DEBUG_SKIP(pa, bit) false
#endif

namespace ag {

enum : int { PHASE_BOTH = 0, PHASE_PHYS = 1, PHASE_POST = 2 };

__device__ __forceinline__ float int_as_f(int v) { return __int_as_float(v); }

template <int CTL, int PHASE>
__global__ __launch_bounds__(64) void planning_step_kernel(const KArgs k, const PlanArgs pa) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    __shared__ float tab[kNumVariants * 8];
    __shared__ float tile[64 * (kPlanNumObs + 1)];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 64 + tid;
    const bool active = i < k.n;
    if (PHASE != PHASE_PHYS) {
        for (int t = tid; t < kNumVariants * 8; t += 64) tab[t] = pa.table[t];
        __syncthreads();
    }
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (PHASE != PHASE_PHYS && blockIdx.x == 0 && tid == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;

    EnvState s;
    CtlState c;
    load_env(k, i, s);
    load_ctl<CTL>(k, i, c);
    float raw_a[A];
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

    if (PHASE != PHASE_POST) planning_physics<CTL>(s, c, raw_a, P);

    if (PHASE == PHASE_PHYS) {
        store_env(k, i, s);
        store_ctl<CTL>(k, i, c);
        return;
    }

    // ---- POST
    float pre_a[A];
    {
        const float4 p4 = k.PA[i];
        pre_a[0] = p4.x; pre_a[1] = p4.y; pre_a[2] = p4.z; pre_a[3] = p4.w;
        if (A == 5) pre_a[A - 1] = k.PA4[i];
    }
    PlanExtra x;
    {
        const float4 g = pa.GOAL[i], e = pa.PRP[i];
        x.goal = V3{g.x, g.y, g.z}; x.prev_related_dist = g.w;
        x.pre_pos = V3{e.x, e.y, e.z}; x.esdf = e.w;
    }
    // check_collisions (customized.py:393-397 -> analytic): robot sphere vs the 40 capped cylinders and the ground
    int collided = (s.p.z <= kRobotRadius) ? 1 : 0;
    if (pa.ext_collisions != nullptr) collided = (active && pa.ext_collisions[i] != 0.0f) ? 1 : 0;
    for (int j = 0; pa.ext_collisions == nullptr && j < kNumObst; ++j) {
        const float4 ob = pa.OB[(size_t)j * pa.n_pad + i];
        const float dx = s.p.x - ob.x, dy = s.p.y - ob.y;
        if (dx * dx + dy * dy < 9.0f) {   // every cylinder stays within 2.7 m (xy) of its root: farther ones cannot touch
            const Cyl w = world_cylinder(ob.x, ob.y, ob.z, &tab[(__float_as_int(ob.w) % kNumVariants) * 8]);
            if (point_cylinder_distance(s.p, w) <= kRobotRadius) collided = 1;
        }
    }
    float obs[kPlanNumObs];
    PlanOut o;
    planning_post<CTL>(s, x, pre_a, raw_a, collided, P, obs, o);
    if (o.done) {
        float u[124];
        if (pa.ext_uniforms != nullptr) {
            for (int j = 0; j < kPlanResetUniforms; ++j) u[j] = active ? pa.ext_uniforms[(size_t)i * kPlanResetUniforms + j] : 0.5f;
        } else {
            planning_reset_uniforms(P, env_global, u);
        }
        float* ob = reinterpret_cast<float*>(pa.OB) + (size_t)i * 4;
        planning_reset(s, c, x, pre_a, A, u, ob, (size_t)pa.n_pad * 4);
    }
    x.prev_related_dist = o.related_dist;   // planning.py:183 runs after the optional reset, for every env
    o.timeout = (s.progress > P.max_episode_length) ? 1 : 0;

    store_env(k, i, s);
    store_ctl<CTL>(k, i, c);
    k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
    if (A == 5) k.PA4[i] = pre_a[A - 1];
    pa.GOAL[i] = make_float4(x.goal.x, x.goal.y, x.goal.z, x.prev_related_dist);
    pa.PRP[i] = make_float4(x.pre_pos.x, x.pre_pos.y, x.pre_pos.z, x.esdf);
    const unsigned long long ballot = __ballot(active && o.done);
    if (active) {
        k.rew[i] = o.rew;
        k.reset[i] = (long long)o.done;
        k.timeout[i] = (uint8_t)o.timeout;
        pa.collisions[i] = (float)collided;
        if (tid == 0) k.mask[i >> 6] = ballot;
        if (pa.terms[0] != nullptr) {
#pragma unroll
            for (int t = 0; t < kPlanNumTerms; ++t) pa.terms[t][i] = o.terms[t];
        }
    }
    // obs rows [n,16]: stage through LDS, write 16-byte-per-lane contiguous lines
    constexpr int ST = kPlanNumObs + 1;
#pragma unroll
    for (int j = 0; j < kPlanNumObs; ++j) tile[tid * ST + j] = obs[j];
    __syncthreads();
    const int env0 = blockIdx.x * 64;
    const int valid = min(64, k.n - env0) * kPlanNumObs;
    float* out = k.obs + (size_t)env0 * kPlanNumObs;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = tid + it * 64;          // 64*16/4 = 256 float4
        const int e = 4 * m;
        if (e + 3 < valid) {
            const int row = e >> 4, col = e & 15;
            reinterpret_cast<float4*>(out)[m] = make_float4(tile[row * ST + col], tile[row * ST + col + 1],
                                                            tile[row * ST + col + 2], tile[row * ST + col + 3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
constexpr int kRenderThreads = 1024;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(v, off, 64);
        v = is_max ? fmaxf(v, o) : fminf(v, o);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < kRenderThreads / 64; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
    return r;
}

enum : int { SCENE_PLANNING = 0, SCENE_AVOID = 1 };

template <int SCENE>
__global__ __launch_bounds__(kRenderThreads) void planning_render_kernel(const KArgs k, const PlanArgs pa) {
    extern __shared__ float lds[];
    float* img = lds;                                   // [kCamW][kCamH]
    CylView* cyl = reinterpret_cast<CylView*>(lds + kCamPix);   // [40] x 10 floats
    float* red = lds + kCamPix + kNumObst * 10;         // [16]
    float* ker = red + 16;                              // [25]
    int* ncand = reinterpret_cast<int*>(ker + 25);
    int* ulo = ncand + 4;                               // [40] first image column a cylinder can touch
    int* uhi = ulo + kNumObst;                          // [40] last one
    const int env = blockIdx.x;
    const int tid = threadIdx.x;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    const uint32_t env_global = P.env_id_offset + (uint32_t)env;
    EnvState s;
    load_env(k, env, s);
    const Camera cam = make_camera(s.p, s.q);
    const float4 g4 = pa.GOAL[env];
    const V3 goal{g4.x, g4.y, g4.z};
    if (tid == 0) *ncand = 0;
    __syncthreads();
    if (SCENE == SCENE_PLANNING && tid < kNumObst) {
        const float4 ob = pa.OB[(size_t)tid * pa.n_pad + env];
        // a cylinder farther than far plane + its own extent from the camera cannot be seen: skip it for every pixel
        const float dx = ob.x - cam.o.x, dy = ob.y - cam.o.y;
        if (dx * dx + dy * dy < (kCamFar + 2.7f) * (kCamFar + 2.7f)) {
            const Cyl w = world_cylinder(ob.x, ob.y, ob.z, pa.table + (__float_as_int(ob.w) % kNumVariants) * 8);
            // conservative image-column interval: project both axis end points, inflated by the radius, onto the
            // image plane (u = W/2 - fx * y_c / x_c).  Any end point closer than 5 cm to the camera plane -> all columns.
            int lo = 0, hi = kCamW - 1;
            float rmin = kInf, rmax = -kInf;
            bool full = false;
            int behind = 0;
            float xc_min = kInf;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float sg = e == 0 ? w.h : -w.h;
                const V3 d{w.cx + sg * w.nx - cam.o.x, w.cy + sg * w.ny - cam.o.y, w.cz + sg * w.nz - cam.o.z};
                const float xc = cam.R[0] * d.x + cam.R[3] * d.y + cam.R[6] * d.z;
                const float yc = cam.R[1] * d.x + cam.R[4] * d.y + cam.R[7] * d.z;
                if (xc + w.r < 0.0f) ++behind;          // every pixel ray has a positive camera-x component
                xc_min = fminf(xc_min, xc);
                if (xc - w.r < 0.05f) { full = true; continue; }
                const float a0 = (yc - w.r) / (xc - w.r), a1 = (yc - w.r) / (xc + w.r);
                const float b0 = (yc + w.r) / (xc - w.r), b1 = (yc + w.r) / (xc + w.r);
                rmin = fminf(rmin, fminf(a0, a1));
                rmax = fmaxf(rmax, fmaxf(b0, b1));
            }
            if (!full) {
                lo = max(0, (int)floorf((float)kCamW / 2.0f - kCamFx * rmax) - 2);
                hi = min(kCamW - 1, (int)ceilf((float)kCamW / 2.0f - kCamFx * rmin) + 2);
            }
            // skipped: entirely behind the camera plane, or entirely beyond the far plane (z-depth > 5 m)
            if (lo <= hi && behind < 2 && xc_min - w.r <= kCamFar) {
                const int slot = atomicAdd(ncand, 1);
                cyl[slot] = make_cyl_view(cam.o, w);
                ulo[slot] = lo;
                uhi[slot] = hi;
            }
        }
    }
    if (tid < 25) {   // random 5x5 "blur" kernel: randint(0, 256) / 256, customized.py:417-419
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_IMG_KERNEL, (uint32_t)(tid >> 2), P.key0, P.key1);
        const uint32_t w = (tid & 3) == 0 ? r.x : ((tid & 3) == 1 ? r.y : ((tid & 3) == 2 ? r.z : r.w));
        ker[tid] = (float)(w >> 24) / 256.0f;
    }
    __syncthreads();
    const int n = *ncand;
    // ---- pass 1: ray-cast, clip to 4.5 m, normalise (customized.py:402-404); index p = u * H + v ([W][H] layout)
    float vmax = 0.0f;
    for (int p = tid; p < kCamPix; p += kRenderThreads) {
        const int u = p / kCamH, v = p - u * kCamH;
        float d;
        if (SCENE == SCENE_AVOID) d = depth_pixel_box(cam, pixel_direction(cam, u, v), goal);      // GOAL holds the cube position
        else d = AG_DEBUG_SKIP(pa, 1) ? 3.0f : depth_pixel_culled(cam, pixel_direction(cam, u, v), cyl, ulo, uhi, u, n, goal);
        d = d > 4.5f ? 4.5f : d;
        d = fminf(fmaxf(d, 0.0f), 4.5f) / 4.5f;
        img[p] = d;
        vmax = fmaxf(vmax, d);
    }
    float mx = block_reduce(vmax, red, true);
    // ---- pass 2: additive N(0, 0.1), clamp to [0, max] (customized.py:406-409); 4 pixels per Philox block
    vmax = 0.0f;
    for (int b = tid; b < (AG_DEBUG_SKIP(pa, 2) ? 0 : kCamPix / 4); b += kRenderThreads) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_IMG_ADD, (uint32_t)b, P.key0, P.key1);
        float z[4];
        box_muller(r.x, r.y, z[0], z[1]);
        box_muller(r.z, r.w, z[2], z[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = fminf(fmaxf(img[4 * b + q] + (0.0f + 0.1f * z[q]), 0.0f), mx);
            img[4 * b + q] = d;
            vmax = fmaxf(vmax, d);
        }
    }
    mx = block_reduce(vmax, red, true);
    // ---- pass 3: multiplicative N(1, 0.3), clamp to [0, max] (customized.py:411-414)
    for (int b = tid; b < (AG_DEBUG_SKIP(pa, 2) ? 0 : kCamPix / 4); b += kRenderThreads) {
        const U4 r = philox4x32_10(env_global, P.tick, STREAM_IMG_MUL, (uint32_t)b, P.key0, P.key1);
        float z[4];
        box_muller(r.x, r.y, z[0], z[1]);
        box_muller(r.z, r.w, z[2], z[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) img[4 * b + q] = fminf(fmaxf(img[4 * b + q] * (1.0f + 0.3f * z[q]), 0.0f), mx);
    }
    __syncthreads();
    // ---- pass 4: 5x5 cross-correlation with zero padding (F.conv2d, customized.py:416-424), min pixel
    float vmin = kInf;
    float* out = pa.image + (size_t)env * kCamPix;
    for (int p = tid; p < (AG_DEBUG_SKIP(pa, 4) ? 0 : kCamPix); p += kRenderThreads) {
        const int u = p / kCamH, v = p - u * kCamH;
        float acc = 0.0f;
        if (u >= 2 && u < kCamW - 2 && v >= 2 && v < kCamH - 2) {     // interior: no bounds checks (97 % of the pixels)
            const float* c0 = img + (u - 2) * kCamH + (v - 2);
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) acc += ker[a * 5 + b] * c0[a * kCamH + b];
        } else {
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const int uu = u + a - 2;
                if (uu < 0 || uu >= kCamW) continue;
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const int vv = v + b - 2;
                    if (vv >= 0 && vv < kCamH) acc += ker[a * 5 + b] * img[uu * kCamH + vv];
                }
            }
        }
        out[p] = acc;
        vmin = fminf(vmin, acc);
    }
    const float mn = block_reduce(vmin, red, false);
    if (tid == 0) {
        float4 e = pa.PRP[env];
        e.w = mn;
        pa.PRP[env] = e;
    }
}

__global__ void planning_reset_all_kernel(const KArgs k, const PlanArgs pa, int num_actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pa.n_pad) return;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (i == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    // obstacle variants: drawn once per env from the counter RNG, fixed for the env's lifetime
    for (int b = 0; b < kNumObst / 4; ++b) {
        const U4 r = philox4x32_10(env_global, 0xFFFFFFFFu, STREAM_VARIANT, (uint32_t)b, P.key0, P.key1);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        for (int q = 0; q < 4; ++q) {
            float4 ob = pa.OB[(size_t)(4 * b + q) * pa.n_pad + i];
            ob.w = __int_as_float((int)(w[q] % kNumVariants));
            pa.OB[(size_t)(4 * b + q) * pa.n_pad + i] = ob;
        }
    }
    EnvState s;
    CtlState c;
    PlanExtra x;
    float pre_a[5];
    float u[124];
    planning_reset_uniforms(P, env_global, u);
    planning_reset(s, c, x, pre_a, num_actions, u, reinterpret_cast<float*>(pa.OB) + (size_t)i * 4, (size_t)pa.n_pad * 4);
    // esdf_dist is re-derived from the (possibly stale) image every step (planning.py:162-163), so the value set at
    // planning.py:136 is never observed: keep the min of the image currently in memory (0 for the initial zeros)
    x.esdf = pa.PRP[i].w;
    store_env(k, i, s);
    store_ctl<CTL_POS>(k, i, c);
    k.PA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k.PA4[i] = 0.f;
    pa.GOAL[i] = make_float4(x.goal.x, x.goal.y, x.goal.z, 0.0f);
    pa.PRP[i] = make_float4(0.f, 0.f, 0.f, x.esdf);
    if (i < k.n) {
        k.rew[i] = 0.f;
        k.reset[i] = 1;
        k.timeout[i] = 0;
        pa.collisions[i] = 0.f;
        if ((i & 63) == 0) k.mask[i >> 6] = 0ull;
    }
}

size_t planning_render_lds_bytes() { return (size_t)(kCamPix + kNumObst * 10 + 16 + 25 + 4 + 2 * kNumObst + 8) * sizeof(float); }

template <int CTL>
static hipError_t launch_phase(const KArgs& k, const PlanArgs& pa, int phase, hipStream_t st) {
    const dim3 grid((k.n + 63) / 64), block(64);
    if (phase == PHASE_BOTH) hipLaunchKernelGGL((planning_step_kernel<CTL, PHASE_BOTH>), grid, block, 0, st, k, pa);
    else if (phase == PHASE_PHYS) hipLaunchKernelGGL((planning_step_kernel<CTL, PHASE_PHYS>), grid, block, 0, st, k, pa);
    else hipLaunchKernelGGL((planning_step_kernel<CTL, PHASE_POST>), grid, block, 0, st, k, pa);
    return hipGetLastError();
}

hipError_t launch_planning_step(const KArgs& k, const PlanArgs& pa, int ctl, int phase, hipStream_t st) {
    switch (ctl) {
        case CTL_POS: return launch_phase<CTL_POS>(k, pa, phase, st);
        case CTL_VEL: return launch_phase<CTL_VEL>(k, pa, phase, st);
        case CTL_RATE: return launch_phase<CTL_RATE>(k, pa, phase, st);
        case CTL_PROP: return launch_phase<CTL_PROP>(k, pa, phase, st);
        default: return hipErrorInvalidValue;   // atti has 5 actions: planning's 16-dim obs holds 4 (planning.py:214)
    }
}

// The dynamic-LDS limit is an attribute of (function, device): remembered per device ordinal, as in split_gemm.hip
template <int SCENE>
static hipError_t render_lds_attr(size_t lds) {
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(planning_render_kernel<SCENE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    return hipSuccess;
}

hipError_t launch_planning_render(const KArgs& k, const PlanArgs& pa, hipStream_t st) {
    const size_t lds = planning_render_lds_bytes();
    if (hipError_t e = render_lds_attr<SCENE_PLANNING>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(planning_render_kernel<SCENE_PLANNING>, dim3(k.n), dim3(kRenderThreads), lds, st, k, pa);
    return hipGetLastError();
}

hipError_t launch_avoid_render(const KArgs& k, const PlanArgs& pa, hipStream_t st) {
    const size_t lds = planning_render_lds_bytes();
    if (hipError_t e = render_lds_attr<SCENE_AVOID>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(planning_render_kernel<SCENE_AVOID>, dim3(k.n), dim3(kRenderThreads), lds, st, k, pa);
    return hipGetLastError();
}

hipError_t launch_planning_reset_all(const KArgs& k, const PlanArgs& pa, int num_actions, hipStream_t st) {
    hipLaunchKernelGGL(planning_reset_all_kernel, dim3(pa.n_pad / 256), dim3(256), 0, st, k, pa, num_actions);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Balloon / Avoid step kernel: one env per lane, same phase split as Planning.
//   Balloon: GOAL = (balloon xyz, -), PRP = (pre_root_positions, -); 18-dim noisy observation; always PHASE_BOTH.
//   Avoid  : GOAL = (cube xyz, -), OB slot 0 = (cube velocity, -), PRP = (pre_root_positions, min pixel); 16-dim
//            observation; PHYS / render / POST on camera steps like Planning.
// ---------------------------------------------------------------------------------------------------
template <int TASK, int CTL, int PHASE>
__global__ __launch_bounds__(64) void custom_step_kernel(const KArgs k, const PlanArgs pa) {
    constexpr int A = CtlTraits<CTL>::kNumActions;
    constexpr int NOBS = (TASK == TASK_BALLOON) ? kBalloonNumObs : kAvoidNumObs;
    constexpr int NTERMS = (TASK == TASK_BALLOON) ? kBalloonNumTerms : kAvoidNumTerms;
    constexpr int NU = (TASK == TASK_BALLOON) ? kBalloonResetUniforms : kAvoidResetUniforms;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 64 + tid;
    const bool active = i < k.n;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (PHASE != PHASE_PHYS && blockIdx.x == 0 && tid == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;

    EnvState s;
    CtlState c;
    load_env(k, i, s);
    load_ctl<CTL>(k, i, c);
    float raw_a[A];
#pragma unroll
    for (int j = 0; j < A; ++j) raw_a[j] = active ? k.actions[(size_t)i * A + j] : 0.0f;
    const float4 g4 = pa.GOAL[i];
    V3 tgt{g4.x, g4.y, g4.z};                   // balloon position / cube position
    V3 obj_v{0.0f, 0.0f, 0.0f};
    if (TASK == TASK_AVOID) { const float4 v4 = pa.OB[i]; obj_v = V3{v4.x, v4.y, v4.z}; }

    if (PHASE != PHASE_POST) {
        planning_physics<CTL>(s, c, raw_a, P);
        if (TASK == TASK_AVOID) avoid_object_step(tgt, obj_v, P.dt);
    }
    if (PHASE == PHASE_PHYS) {
        store_env(k, i, s);
        store_ctl<CTL>(k, i, c);
        pa.GOAL[i] = make_float4(tgt.x, tgt.y, tgt.z, 0.0f);
        pa.OB[i] = make_float4(obj_v.x, obj_v.y, obj_v.z, 0.0f);
        return;
    }
    float pre_a[A];
    {
        const float4 p4 = k.PA[i];
        pre_a[0] = p4.x; pre_a[1] = p4.y; pre_a[2] = p4.z; pre_a[3] = p4.w;
        if (A == 5) pre_a[A - 1] = k.PA4[i];
    }
    const float4 e4 = pa.PRP[i];
    V3 pre_pos{e4.x, e4.y, e4.z};
    // check_collisions (customized.py:393-397 -> analytic): ground plane; Avoid: + the cube (the balloon shares the
    // robot's collision mask and never touches it)
    int collided = (s.p.z <= kRobotRadius) ? 1 : 0;
    if (TASK == TASK_AVOID && point_box_distance(s.p, tgt, kCubeHalf) <= kRobotRadius) collided = 1;
    if (pa.ext_collisions != nullptr) collided = (active && pa.ext_collisions[i] != 0.0f) ? 1 : 0;
    float obs[NOBS];
    CustomOut o;
    if (TASK == TASK_BALLOON) {
        float z[18];
        if (k.ext_noise != nullptr) {
#pragma unroll
            for (int j = 0; j < 18; ++j) z[j] = active ? k.ext_noise[(size_t)i * 18 + j] : 0.0f;
        } else if (!P.noise_off) {
            obs_noise_normals(P, env_global, z);
        } else {
#pragma unroll
            for (int j = 0; j < 18; ++j) z[j] = 0.0f;
        }
        balloon_post<CTL>(s, tgt, pre_pos, pre_a, raw_a, collided, z, P, obs, o);
    } else {
        avoid_post<CTL>(s, pre_pos, pre_a, raw_a, collided, P, obs, o);
    }
    if (o.done) {
        float u[16];
        if (pa.ext_uniforms != nullptr) {
            for (int j = 0; j < NU; ++j) u[j] = active ? pa.ext_uniforms[(size_t)i * NU + j] : 0.5f;
        } else {
            custom_reset_uniforms(P, env_global, u);
        }
        if (TASK == TASK_BALLOON) balloon_reset(s, c, tgt, pre_pos, pre_a, A, u);
        else avoid_reset(s, c, tgt, obj_v, pre_pos, pre_a, A, u);
    }
    o.timeout = (s.progress > P.max_episode_length) ? 1 : 0;
    store_env(k, i, s);
    store_ctl<CTL>(k, i, c);
    k.PA[i] = make_float4(pre_a[0], pre_a[1], pre_a[2], pre_a[3]);
    if (A == 5) k.PA4[i] = pre_a[A - 1];
    pa.GOAL[i] = make_float4(tgt.x, tgt.y, tgt.z, 0.0f);
    if (TASK == TASK_AVOID) pa.OB[i] = make_float4(obj_v.x, obj_v.y, obj_v.z, 0.0f);
    pa.PRP[i] = make_float4(pre_pos.x, pre_pos.y, pre_pos.z, e4.w);
    const unsigned long long ballot = __ballot(active && o.done);
    if (active) {
        k.rew[i] = o.rew;
        k.reset[i] = (long long)o.done;
        k.timeout[i] = (uint8_t)o.timeout;
        pa.collisions[i] = (float)collided;
        if (tid == 0) k.mask[i >> 6] = ballot;
        if (pa.terms[0] != nullptr) {
#pragma unroll
            for (int t = 0; t < NTERMS; ++t) pa.terms[t][i] = o.terms[t];
        }
        float* out = k.obs + (size_t)i * NOBS;
#pragma unroll
        for (int j = 0; j < NOBS / 2; ++j) reinterpret_cast<float2*>(out)[j] = make_float2(obs[2 * j], obs[2 * j + 1]);
    }
}

template <int TASK>
__global__ void custom_reset_all_kernel(const KArgs k, const PlanArgs pa, int num_actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pa.n_pad) return;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (i == 0) *k.tick_out = P.tick + 1u;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    EnvState s;
    CtlState c;
    V3 tgt{0.f, 0.f, 0.f}, obj_v{0.f, 0.f, 0.f}, pre_pos{0.f, 0.f, 0.f};
    float pre_a[5];
    float u[16];
    custom_reset_uniforms(P, env_global, u);
    if (TASK == TASK_BALLOON) balloon_reset(s, c, tgt, pre_pos, pre_a, num_actions, u);
    else avoid_reset(s, c, tgt, obj_v, pre_pos, pre_a, num_actions, u);
    store_env(k, i, s);
    store_ctl<CTL_POS>(k, i, c);
    k.PA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k.PA4[i] = 0.f;
    pa.GOAL[i] = make_float4(tgt.x, tgt.y, tgt.z, 0.0f);
    pa.OB[i] = make_float4(obj_v.x, obj_v.y, obj_v.z, 0.0f);
    pa.PRP[i] = make_float4(0.f, 0.f, 0.f, pa.PRP[i].w);
    if (i < k.n) {
        k.rew[i] = 0.f;
        k.reset[i] = 1;
        k.timeout[i] = 0;
        pa.collisions[i] = 0.f;
        if ((i & 63) == 0) k.mask[i >> 6] = 0ull;
    }
}

template <int TASK, int CTL>
static hipError_t launch_custom_phase(const KArgs& k, const PlanArgs& pa, int phase, hipStream_t st) {
    const dim3 grid((k.n + 63) / 64), block(64);
    if (phase == PHASE_BOTH) hipLaunchKernelGGL((custom_step_kernel<TASK, CTL, PHASE_BOTH>), grid, block, 0, st, k, pa);
    else if (phase == PHASE_PHYS) hipLaunchKernelGGL((custom_step_kernel<TASK, CTL, PHASE_PHYS>), grid, block, 0, st, k, pa);
    else hipLaunchKernelGGL((custom_step_kernel<TASK, CTL, PHASE_POST>), grid, block, 0, st, k, pa);
    return hipGetLastError();
}

// task: TASK_BALLOON / TASK_AVOID.  Avoid has a 16-dim observation with 4 action slots: atti (5 actions) is refused, exactly
// like Planning - the reference itself fails there (`obs_buf[..., 12:16] = actions_local` with a [N,5] tensor, avoid.py:232).
hipError_t launch_custom_step(const KArgs& k, const PlanArgs& pa, int task, int ctl, int phase, hipStream_t st) {
    if (task == TASK_BALLOON) {
        switch (ctl) {
            case CTL_POS: return launch_custom_phase<TASK_BALLOON, CTL_POS>(k, pa, phase, st);
            case CTL_VEL: return launch_custom_phase<TASK_BALLOON, CTL_VEL>(k, pa, phase, st);
            case CTL_ATTI: return launch_custom_phase<TASK_BALLOON, CTL_ATTI>(k, pa, phase, st);
            case CTL_RATE: return launch_custom_phase<TASK_BALLOON, CTL_RATE>(k, pa, phase, st);
            case CTL_PROP: return launch_custom_phase<TASK_BALLOON, CTL_PROP>(k, pa, phase, st);
        }
    } else if (task == TASK_AVOID) {
        switch (ctl) {
            case CTL_POS: return launch_custom_phase<TASK_AVOID, CTL_POS>(k, pa, phase, st);
            case CTL_VEL: return launch_custom_phase<TASK_AVOID, CTL_VEL>(k, pa, phase, st);
            case CTL_RATE: return launch_custom_phase<TASK_AVOID, CTL_RATE>(k, pa, phase, st);
            case CTL_PROP: return launch_custom_phase<TASK_AVOID, CTL_PROP>(k, pa, phase, st);
        }
    }
    return hipErrorInvalidValue;
}

// reset_idx(env_ids) for a caller-chosen subset of a Planning / Balloon / Avoid handle (planning.py:63-136, balloon.py:57-99,
// avoid.py:91-163): same per-env reset functions as the in-kernel reset of the step.
template <int TASK>
__global__ void custom_reset_ids_kernel(const KArgs k, const PlanArgs pa, int num_actions, const int* ids, int count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    StepParams P = k.P;
    P.tick = *k.tick_in;
    if (j == 0) *k.tick_out = P.tick + 1u;
    if (j >= count) return;
    const int i = ids[j];
    if (i < 0 || i >= k.n) return;
    const uint32_t env_global = P.env_id_offset + (uint32_t)i;
    EnvState s;
    CtlState c;
    float pre_a[5];
    if (TASK == TASK_PLANNING) {
        PlanExtra x;
        float u[124];
        planning_reset_uniforms(P, env_global, u);
        planning_reset(s, c, x, pre_a, num_actions, u, reinterpret_cast<float*>(pa.OB) + (size_t)i * 4, (size_t)pa.n_pad * 4);
        pa.GOAL[i] = make_float4(x.goal.x, x.goal.y, x.goal.z, 0.0f);
        pa.PRP[i] = make_float4(0.f, 0.f, 0.f, pa.PRP[i].w);
    } else {
        V3 tgt{0.f, 0.f, 0.f}, obj_v{0.f, 0.f, 0.f}, pre_pos{0.f, 0.f, 0.f};
        float u[16];
        custom_reset_uniforms(P, env_global, u);
        if (TASK == TASK_BALLOON) balloon_reset(s, c, tgt, pre_pos, pre_a, num_actions, u);
        else avoid_reset(s, c, tgt, obj_v, pre_pos, pre_a, num_actions, u);
        pa.GOAL[i] = make_float4(tgt.x, tgt.y, tgt.z, 0.0f);
        pa.OB[i] = make_float4(obj_v.x, obj_v.y, obj_v.z, 0.0f);
        pa.PRP[i] = make_float4(0.f, 0.f, 0.f, pa.PRP[i].w);
    }
    store_env(k, i, s);
    store_ctl<CTL_POS>(k, i, c);
    k.PA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k.PA4[i] = 0.f;
    k.reset[i] = 1;
    atomicOr(&k.mask[i >> 6], 1ull << (i & 63));
}

hipError_t launch_custom_reset_ids(const KArgs& k, const PlanArgs& pa, int task, int num_actions, const int* ids, int count,
                                   hipStream_t st) {
    const dim3 grid((count + 255) / 256), block(256);
    if (task == TASK_PLANNING) hipLaunchKernelGGL(custom_reset_ids_kernel<TASK_PLANNING>, grid, block, 0, st, k, pa, num_actions, ids, count);
    else if (task == TASK_BALLOON) hipLaunchKernelGGL(custom_reset_ids_kernel<TASK_BALLOON>, grid, block, 0, st, k, pa, num_actions, ids, count);
    else hipLaunchKernelGGL(custom_reset_ids_kernel<TASK_AVOID>, grid, block, 0, st, k, pa, num_actions, ids, count);
    return hipGetLastError();
}

hipError_t launch_custom_reset_all(const KArgs& k, const PlanArgs& pa, int task, int num_actions, hipStream_t st) {
    if (task == TASK_BALLOON) hipLaunchKernelGGL(custom_reset_all_kernel<TASK_BALLOON>, dim3(pa.n_pad / 256), dim3(256), 0, st, k, pa, num_actions);
    else hipLaunchKernelGGL(custom_reset_all_kernel<TASK_AVOID>, dim3(pa.n_pad / 256), dim3(256), 0, st, k, pa, num_actions);
    return hipGetLastError();
}

}  // namespace ag
