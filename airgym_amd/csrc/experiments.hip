// experiments.hip - benchmark / diagnostic entry points (include/airgym_hip_debug.h).  NOT part of the shipped library:
// compiled only by `python airgym_amd/csrc/build.py --experiments` into libairgym_hip_exp.so, which tools/ load through
// AIRGYM_EXPERIMENTS=1.  Nothing here replaces reference behaviour.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/airgym_hip_debug.h"
#include "handle.hpp"

namespace {

// Diagnostic: same loads/stores as the Hovering/CTBR step (7 float4 in, 7 float4 + obs row + reward + flags out),
// no arithmetic.  Its duration is the launch + memory-latency floor any one-launch-per-step design pays.
__global__ __launch_bounds__(64) void touch_kernel(ag::KArgs k, const float* actions, int num_obs) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    ag::EnvState s;
    ag::CtlState c;
    ag::load_env(k, i, s);
    ag::load_ctl<ag::CTL_RATE>(k, i, c);
    const float4 pa = k.PA[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < k.n) a = reinterpret_cast<const float4*>(actions)[i];
    s.p.x += 1e-30f * (pa.x + a.x);
    ag::store_env(k, i, s);
    ag::store_ctl<ag::CTL_RATE>(k, i, c);
    k.PA[i] = a;
    if (i < k.n) {
        k.rew[i] = s.p.x;
        k.reset[i] = 0;
        k.timeout[i] = 0;
        float* o = k.obs + (size_t)i * num_obs;
        for (int j = 0; j < num_obs; j += 2) reinterpret_cast<float2*>(o)[j >> 1] = make_float2(s.p.y, s.p.z);
    }
}

// Diagnostic variants of touch_kernel: MODE 1 = non-temporal stores, 2 = non-temporal loads + stores, 3 = empty kernel
// (pure dependent-launch boundary).  Used by tools/touch_probe.py to price the kernel boundary.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void touch_variant_kernel(ag::KArgs k, const float* actions, int num_obs) {
    if (MODE == 3) return;
    const int i = blockIdx.x * 64 + threadIdx.x;
    nt_f4 in[7];
    const nt_f4* src[7] = {(const nt_f4*)k.S[0], (const nt_f4*)k.S[1], (const nt_f4*)k.S[2], (const nt_f4*)k.S[3],
                           (const nt_f4*)k.C[0], (const nt_f4*)k.C[1], (const nt_f4*)k.PA};
#pragma unroll
    for (int j = 0; j < 7; ++j) in[j] = (MODE == 2) ? __builtin_nontemporal_load(src[j] + i) : src[j][i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < k.n) a = reinterpret_cast<const float4*>(actions)[i];
    in[0].x += 1e-30f * (in[6].x + a.x);
    in[6] = nt_f4{a.x, a.y, a.z, a.w};
    nt_f4* dst[7] = {(nt_f4*)k.S[0], (nt_f4*)k.S[1], (nt_f4*)k.S[2], (nt_f4*)k.S[3], (nt_f4*)k.C[0], (nt_f4*)k.C[1], (nt_f4*)k.PA};
#pragma unroll
    for (int j = 0; j < 7; ++j) __builtin_nontemporal_store(in[j], dst[j] + i);
    if (i < k.n) {
        __builtin_nontemporal_store(in[0].x, k.rew + i);
        k.reset[i] = 0;
        k.timeout[i] = 0;
        float* o = k.obs + (size_t)i * num_obs;
        for (int j = 0; j < num_obs; j += 2) __builtin_nontemporal_store(nt_f2{in[0].y, in[0].z}, reinterpret_cast<nt_f2*>(o) + (j >> 1));
    }
}

// Diagnostic: where does the hardware put the waves of the step kernel's launch geometry (grid n/64 x 128 threads)?
// One uint2 per wave: HW_ID (wave / SIMD / CU / SE fields) and XCC_ID.
__global__ __launch_bounds__(128) void wave_placement_kernel(uint2* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID (gfx940+)
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 2 + (threadIdx.x >> 6)] = make_uint2(hw, xcc);
}

}  // namespace

extern "C" {

int ag_debug_planning_render_parts(ag_handle h, int skip_mask) {
    if (!h || h->cfg.task != AG_TASK_PLANNING) return AG_ERR_INVALID_ARG;
    if (skip_mask < 0 || skip_mask > 7) return AG_ERR_INVALID_ARG;
    h->force_render = 1 | (skip_mask << 1);
    return AG_OK;
}

int ag_debug_touch_variant(ag_handle h, const float* actions_dev, int mode, void* stream) {
    if (!h || !actions_dev) return AG_ERR_INVALID_ARG;
    const dim3 grid((h->cfg.num_envs + 63) / 64), block(64);
    const int nobs = h->num_obs;
    if (mode == 1) hipLaunchKernelGGL(touch_variant_kernel<1>, grid, block, 0, (hipStream_t)stream, h->k, actions_dev, nobs);
    else if (mode == 2) hipLaunchKernelGGL(touch_variant_kernel<2>, grid, block, 0, (hipStream_t)stream, h->k, actions_dev, nobs);
    else if (mode == 3) hipLaunchKernelGGL(touch_variant_kernel<3>, grid, block, 0, (hipStream_t)stream, h->k, actions_dev, nobs);
    else return AG_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

int ag_debug_wave_placement(ag_handle h, unsigned int* out_dev, void* stream) {
    if (!h || !out_dev) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(wave_placement_kernel, dim3((h->cfg.num_envs + 63) / 64), dim3(128), 0, (hipStream_t)stream,
                       (uint2*)out_dev);
    if (hipGetLastError() != hipSuccess) return AG_ERR_HIP;
    return AG_OK;
}

int ag_debug_touch(ag_handle h, const float* actions_dev, void* stream) {
    if (!h || !actions_dev) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(touch_kernel, dim3((h->cfg.num_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->k,
                       actions_dev, h->num_obs);
    if (hipGetLastError() != hipSuccess) return AG_ERR_HIP;
    return AG_OK;
}

}  // extern "C"
