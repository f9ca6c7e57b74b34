// tail_parts.hpp - the device bodies of the optimizer step's tail, shared by the kernels that run them one launch each
// (ppo_kernels.hip: sum_rows_stage2_kernel, adam_norm_kernel, adam_clip_step_kernel; split_gemm.hip: split_in_prepare_kernel) and by
// the fused tail (update_tail.hip: all four behind grid barriers in ONE launch).  One source for the arithmetic: the fused launch
// is bit-identical to the separate ones by construction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "split_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------
// partial-sum reductions (see ppo_kernels.hip, "All partial-sum reductions of one minibatch in TWO launches")
// ---------------------------------------------------------------------------------------------------
constexpr int kSumMaxGroups = 64;        // partial rows are first summed in up to 64 groups per job (>= 8 rows per group)
constexpr int kMaxSumJobs = AG_MAX_SUM_JOBS;

struct SumJobs {
    const float* in[kMaxSumJobs];
    float* out[kMaxSumJobs];
    int S[kMaxSumJobs];
    int n4[kMaxSumJobs];
    int groups[kMaxSumJobs];
    int block0_s1[kMaxSumJobs + 1];       // first flat block of each job in stage 1 (blocks = ceil(n4/64) * groups)
    int block0_s2[kMaxSumJobs + 1];       // ... and in stage 2 (blocks = ceil(n4/64))
    long long scratch_off[kMaxSumJobs];   // in floats, into scratch [sum over jobs of groups * n]
    float* scratch;
    int njobs;
};

__device__ __forceinline__ int find_job(const int* block0, int njobs, int b) {
    int j = 0;
    while (j + 1 < njobs && b >= block0[j + 1]) ++j;
    return j;
}

// stage 2, flat block `vb` (256 threads = 64 float4 columns x 4 group lanes; red: 256 float4 of LDS; ends with the block's threads
// past their last LDS read only after the caller's next barrier - callers that loop put a __syncthreads() between blocks)
__device__ __forceinline__ void sum_stage2_block(const SumJobs& k, int vb, float4* red) {
    const int j = find_job(k.block0_s2, k.njobs, vb);
    const int bx = vb - k.block0_s2[j];
    const int n4 = k.n4[j], G = k.groups[j];
    const int col = bx * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4) {
        const float4* src = reinterpret_cast<const float4*>(k.scratch + k.scratch_off[j]);
#pragma unroll 4
        for (int g = lane; g < G; g += 4) {
            const float4 v = src[(size_t)g * n4 + col];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < n4) {
        const float4 s0 = red[threadIdx.x], s1 = red[threadIdx.x + 64], s2 = red[threadIdx.x + 128], s3 = red[threadIdx.x + 192];
        float* dst = k.out[j] + (size_t)col * 4;
        dst[0] = (s0.x + s1.x) + (s2.x + s3.x); dst[1] = (s0.y + s1.y) + (s2.y + s3.y);
        dst[2] = (s0.z + s1.z) + (s2.z + s3.z); dst[3] = (s0.w + s1.w) + (s2.w + s3.w);
    }
}

// host side: ag_sum_job[] -> SumJobs (+ the flat block counts of the two stages); AG_OK or the error the entry points return
inline int build_sum_jobs(const ag_sum_job* jobs, int njobs, float* scratch, long long scratch_floats, SumJobs& k, int& b1, int& b2) {
    if (!jobs || !scratch || njobs <= 0) return AG_ERR_INVALID_ARG;
    if (njobs > kMaxSumJobs) return AG_ERR_UNSUPPORTED;
    k = SumJobs{};
    k.scratch = scratch;
    k.njobs = njobs;
    long long off = 0;
    b1 = b2 = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!jobs[j].partials_dev || !jobs[j].out_dev || jobs[j].rows <= 0 || jobs[j].n <= 0) return AG_ERR_INVALID_ARG;
        if (jobs[j].n % 4 != 0 || (reinterpret_cast<uintptr_t>(jobs[j].partials_dev) & 15)) return AG_ERR_UNSUPPORTED;
        k.in[j] = jobs[j].partials_dev;
        k.out[j] = jobs[j].out_dev;
        k.S[j] = jobs[j].rows;
        k.n4[j] = jobs[j].n / 4;
        int G = jobs[j].rows / 8;
        G = G < 1 ? 1 : (G > kSumMaxGroups ? kSumMaxGroups : G);
        k.groups[j] = G;
        k.scratch_off[j] = off;
        off += (long long)G * jobs[j].n;
        const int nbx = (k.n4[j] + 63) / 64;
        k.block0_s1[j] = b1;
        k.block0_s2[j] = b2;
        b1 += nbx * G;
        b2 += nbx;
    }
    k.block0_s1[njobs] = b1;
    k.block0_s2[njobs] = b2;
    if (off > scratch_floats || (reinterpret_cast<uintptr_t>(scratch) & 15)) return AG_ERR_INVALID_ARG;
    return AG_OK;
}

// ---------------------------------------------------------------------------------------------------
// clip-by-norm + Adam + KL-adaptive learning rate (see ppo_kernels.hip)
// ---------------------------------------------------------------------------------------------------
struct AdamArgs {
    float* p; float* g; float* m; float* v;
    double* state;   // unused by the kernels (kept for symmetry with the C entry point)
    int n;
    float beta1, beta2, eps, weight_decay, max_grad_norm;   // max_grad_norm <= 0: no clipping
    float kl_threshold, min_lr, max_lr;                     // kl_threshold <= 0: LR not adapted
};

constexpr int kAdamBlocks = 64;
constexpr int kAdamThreads = 256;

// partial sum of g^2 of virtual block `vb` of kAdamBlocks (256 threads; red: 4 floats of LDS); thread 0 returns the block's sum
__device__ __forceinline__ float adam_norm_block(const float* __restrict__ g, int n, int vb, float* red) {
    float ss = 0.f;
    for (int i = vb * kAdamThreads + threadIdx.x; i < n; i += kAdamBlocks * kAdamThreads) {
        const float x = g[i];
        ss += x * x;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kAdamThreads / 64; ++w) t += red[w];
    return t;
}

struct AdamScalars { float coef, step_size, bc2r; double lr, step; };

// what every thread of the update derives from the 64 partials and {lr, step}
__device__ __forceinline__ AdamScalars adam_scalars(const AdamArgs& k, const float* __restrict__ partial, double lr, double step_in) {
    float tot = 0.f;
    for (int w = 0; w < kAdamBlocks; ++w) tot += partial[w];
    const float norm = sqrtf(tot);
    AdamScalars s;
    s.coef = (k.max_grad_norm > 0.f) ? fminf(k.max_grad_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    s.lr = lr;
    s.step = step_in + 1.0;
    const double bc1 = 1.0 - pow((double)k.beta1, s.step);
    const double bc2 = 1.0 - pow((double)k.beta2, s.step);
    s.step_size = (float)(lr / bc1);
    s.bc2r = (float)(1.0 / sqrt(bc2));
    return s;
}

// the KL-adaptive rule (legacy schedule: evaluated every minibatch, applies to the NEXT step); one thread
__device__ __forceinline__ double adam_next_lr(const AdamArgs& k, double lr) {
    double nlr = lr;
    if (k.kl_threshold > 0.f) {
        const double kl = (double)k.g[k.n];
        if (kl > 2.0 * k.kl_threshold) nlr = fmax(lr / 1.5, (double)k.min_lr);
        if (kl < 0.5 * k.kl_threshold) nlr = fmin(lr * 1.5, (double)k.max_lr);
    }
    return nlr;
}

__device__ __forceinline__ void adam_update_element(const AdamArgs& k, const AdamScalars& s, int i) {
    float g = k.g[i] * s.coef;
    k.g[i] = g;
    const float p = k.p[i];
    if (k.weight_decay != 0.f) g += k.weight_decay * p;
    const float m = k.beta1 * k.m[i] + (1.f - k.beta1) * g;
    const float v = k.beta2 * k.v[i] + (1.f - k.beta2) * g * g;
    k.m[i] = m;
    k.v[i] = v;
    const float denom = sqrtf(v) * s.bc2r + k.eps;
    k.p[i] = p - s.step_size * (m / denom);
}

// ---------------------------------------------------------------------------------------------------
// weight images of the split GEMMs (see split_gemm.hip)
// ---------------------------------------------------------------------------------------------------
constexpr int BN = 256, BK = 16, KDIM = 256;
constexpr int B_UNITS = 3 * 2 * BN;       // 16-byte units per B stage
// Images of ag_split_gemm_input_prepare (the fused first layer, FIN > 0):
//   [0, kInImageW1Bytes): W1ext [block 8][K step 2][plane 3][h 2][feature 32] x 16 B, W1ext[f][d] = W1[f][d] (d < D), b1[f] (d = D), 0
//   then the forward planes of W2 in the chain's K order, laid out like ag_split_gemm_prepare's
constexpr int kInImageW1Bytes = 8 * 2 * 3 * 2 * 32 * 16;
constexpr int kInPrepareW1Units = 8 * 2 * 2 * 32;
__host__ __device__ constexpr int in_prepare_threads(bool with_bwd) { return kInPrepareW1Units + (with_bwd ? 2 : 1) * 16 * 2 * BN; }

// flat work item t of in_prepare_threads(planes_t != nullptr)
__device__ __forceinline__ void split_in_prepare_unit(int t, const float* __restrict__ W1, const float* __restrict__ b1, int D,
                                                      const float* __restrict__ W2, uint4* __restrict__ img, uint4* __restrict__ planes_t) {
    if (t < kInPrepareW1Units) {                            // W1ext: one (block, step, h, feature) unit triple per thread
        const int m = t & 31, h = (t >> 5) & 1, st = (t >> 6) & 1, b = t >> 7;
        const int f = 32 * b + m;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = 16 * st + 8 * h + i;
            v[i] = d < D ? W1[(size_t)f * D + d] : (d == D ? b1[f] : 0.0f);
        }
        uint4 p1, p2, p3;
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
        uint4* base = img + (size_t)((b * 2 + st) * 3) * 64 + h * 32 + m;
        base[0] = p1;
        base[64] = p2;
        base[128] = p3;
        return;
    }
    int unit = t - kInPrepareW1Units;                       // W2 planes: one (chunk, k-half, n) unit triple per thread
    const bool bwd = unit >= 16 * 2 * BN;                   // ... then (planes_t != null) the backward image: B[n][k] = W2[k][n], natural K
    if (bwd) unit -= 16 * 2 * BN;
    if (unit >= 16 * 2 * BN || (bwd && planes_t == nullptr)) return;
    const int n = unit % BN, h = (unit / BN) & 1, c = unit / (2 * BN);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (bwd) {
            v[i] = W2[(size_t)(c * BK + h * 8 + i) * BN + n];
        } else {                                            // forward image in the chain's K order
            const int f = 32 * (c >> 1) + 16 * (c & 1) + (i & 3) + 8 * (i >> 2) + 4 * h;
            v[i] = W2[(size_t)n * KDIM + f];
        }
    }
    uint4 p1, p2, p3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
    uint4* chunk = (bwd ? planes_t : img + kInImageW1Bytes / 16) + (size_t)c * B_UNITS;
    chunk[(0 * 2 + h) * BN + n] = p1;
    chunk[(1 * 2 + h) * BN + n] = p2;
    chunk[(2 * 2 + h) * BN + n] = p3;
}

}  // namespace
