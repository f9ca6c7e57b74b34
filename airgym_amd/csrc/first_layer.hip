// first_layer.hip - the first layer of the actor-critic trunk, h1 = ELU(xn W1^T + b1) behind the input normaliser
// (lib/network/mlp.py:36-39, first Linear + ELU; lib/core/running_mean_std.py:78-79), for input widths the forward GEMM cannot
// produce itself (Tracking's 48: split_gemm.hip's FIN block would need a 96 KB weight image beside 98 KB of stages), as a launch of
// its own ON THE MATRIX CORES.  Replaces ag_mlp_input_layer (vector-ALU, weights in registers: 131 us at D = 48, M = 196 608 - twice
// what its 277 MB of traffic cost) in the update of such configurations.
//
//   * Same arithmetic as every other product of the update: exact 3-way bf16 split of both operands, six MFMAs per product,
//     f32 accumulate (one MFMA with AG_SPLIT_PLANES = 1); the bias rides in an all-ones input column (K = D + 1, padded to KST x 16).
//   * Natural orientation: a wave owns 32 batch rows; per block of 32 features  h1[row, f] = x_ext[row, :] . W1ext[f, :]  with the
//     lane's row of inputs as A operand (split once per 32 rows, kept in registers) and the weights as B operand (image in LDS, read
//     as fragments).  Lane (feature, h) then holds rows (r & 3) + 8 (r >> 2) + 4 h in register r: every store instruction writes two
//     whole 128-byte row segments.
//   * 512 threads = 8 waves = 256 rows per workgroup trip (one workgroup per CU: the image is up to 96 KB; eight waves keep enough
//     loads and stores in flight), persistent over the row tiles; LDS = the weight image (24 KB per K step).
//   * Bound by its stores (h1: 1 KB per row): ~60 us at M = 196 608.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "split_common.hpp"

namespace {

constexpr int FL_C = 256;                                       // layer width
constexpr int fl_image_units(int kst) { return 8 * kst * 3 * 2 * 32; }      // [block 8][K step kst][plane 3][h 2][feature 32] x 16 B

int fl_ksteps(int D) { return D + 1 <= 32 ? 2 : (D + 1 <= 64 ? 4 : 0); }

__device__ __forceinline__ float fl_elu(float z) {      // ppo_kernels.hip elu1, split_gemm.hip sg_elu
    return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f;
}

#if AG_SPLIT_PLANES == 3
// W1 [256, D], b1 [256] -> the fragment image: unit (b, s, p, h, m) = plane p of W1ext[32 b + m][16 s + 8 h .. + 7],
// W1ext[f][d] = W1[f][d] (d < D), b1[f] (d = D), 0 behind
__global__ __launch_bounds__(256) void first_layer_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D, int kst,
                                                                  uint4* __restrict__ img) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 8 * kst * 2 * 32) return;
    const int m = t & 31, h = (t >> 5) & 1, st = (t >> 6) % kst, b = t / (64 * kst);
    const int f = 32 * b + m;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int d = 16 * st + 8 * h + i;
        v[i] = d < D ? W1[(size_t)f * D + d] : (d == D ? b1[f] : 0.0f);
    }
    uint4 p1, p2, p3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
    uint4* base = img + (size_t)((b * kst + st) * 3) * 64 + h * 32 + m;
    base[0] = p1;
    base[64] = p2;
    base[128] = p3;
}
#endif

template <int KST>
__global__ __launch_bounds__(512, 2) void first_layer_kernel(const float* __restrict__ obs, const double* __restrict__ mean,
                                                             const double* __restrict__ var, float eps, float clip,
                                                             const uint4* __restrict__ w1img, float* __restrict__ xn,
                                                             float* __restrict__ h1, int M, int D) {
    extern __shared__ uint4 w1s[];                         // the weight image
    __shared__ float nconst[2][64];                        // mean | sd of the input normaliser
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, fh = lane >> 5;
    const bool normalize = mean != nullptr;
    for (int u = tid; u < fl_image_units(KST); u += 512) w1s[u] = w1img[u];
    if (tid < 64) {
        const bool in = normalize && tid < D;
        nconst[0][tid] = in ? (float)mean[tid] : 0.0f;
        nconst[1][tid] = in ? sqrtf((float)var[tid] + eps) : 1.0f;
    }
    __syncthreads();
    const int ntiles = (M + 255) / 256;
    const bool vec2 = (D & 1) == 0;                        // rows are 8-byte aligned: float2 loads
    // the lane's input row of a tile, raw: fetched one tile ahead (a workgroup owns the CU - the 96 KB weight image - so nothing else
    // would cover the latency of these loads at the top of every tile)
    float xnext[KST][8];
#define AG_FL_FETCH(tile_)                                                                             \
    do {                                                                                               \
        const int rr_ = min((tile_) * 256 + wave * 32 + l31, M - 1);                                   \
        const float* xr_ = obs + (size_t)rr_ * D;                                                      \
        _Pragma("unroll") for (int s = 0; s < KST; ++s) {                                              \
            const int d0 = 16 * s + 8 * fh;                                                            \
            if (vec2 && d0 + 8 <= D) {                                                                 \
                _Pragma("unroll") for (int i2 = 0; i2 < 4; ++i2) {                                     \
                    const float2 v2 = reinterpret_cast<const float2*>(xr_ + d0)[i2];                   \
                    xnext[s][2 * i2] = v2.x;                                                           \
                    xnext[s][2 * i2 + 1] = v2.y;                                                       \
                }                                                                                      \
            } else {                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) xnext[s][i] = (d0 + i < D) ? xr_[d0 + i] : 0.0f; \
            }                                                                                          \
        }                                                                                              \
    } while (0)
#ifndef AG_FL_NO_PREFETCH                      // (A/B switch: fetch at the top of the tile that uses the rows)
    if ((int)blockIdx.x < ntiles) AG_FL_FETCH((int)blockIdx.x);
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row_raw = tile * 256 + wave * 32 + l31;
        const bool row_ok = row_raw < M;
        const int row = row_ok ? row_raw : M - 1;          // rows past M are computed on a copy of the last row, never stored
        bf16x8 xq[KST][3];
        float xcur[KST][8];
#ifdef AG_FL_NO_PREFETCH
        AG_FL_FETCH(tile);
#endif
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) xcur[s][i] = xnext[s][i];
#ifndef AG_FL_NO_PREFETCH
        if (tile + (int)gridDim.x < ntiles) AG_FL_FETCH(tile + (int)gridDim.x);
#endif
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            float xv[8];
            const int d0 = 16 * s + 8 * fh;
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = xcur[s][i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = d0 + i;
                float v = xv[i];
                if (normalize) {                           // ag_mlp_input_layer's arithmetic (running_mean_std.py:78-79)
                    const int dc = min(d, 63);
                    v = (v - nconst[0][dc]) / nconst[1][dc];
                    v = fminf(fmaxf(v, -clip), clip);
                    if (!vec2 && xn != nullptr && row_ok && d < D) xn[(size_t)row * D + d] = v;
                    xv[i] = v;
                }
            }
            if (normalize && vec2 && xn != nullptr && row_ok) {      // normalised inputs out, 8 bytes at a time
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    if (d0 + 2 * i2 < D) reinterpret_cast<float2*>(xn + (size_t)row * D + d0)[i2] = make_float2(xv[2 * i2], xv[2 * i2 + 1]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = d0 + i;
                xv[i] = d < D ? xv[i] : (d == D ? 1.0f : 0.0f);      // the all-ones column carries the bias
            }
            uint4 q1, q2, q3;
            split8(make_float4(xv[0], xv[1], xv[2], xv[3]), make_float4(xv[4], xv[5], xv[6], xv[7]), q1, q2, q3);
            xq[s][0] = *reinterpret_cast<const bf16x8*>(&q1);
            xq[s][1] = *reinterpret_cast<const bf16x8*>(&q2);
            xq[s][2] = *reinterpret_cast<const bf16x8*>(&q3);
        }
        // natural orientation: D[row, feature] = x_ext[row, :] . W1ext[feature, :] - A = the lane's row of inputs, B = the weight
        // fragments (the same image units read as the other operand).  Lane (feature l31, h) then holds rows (r & 3) + 8 (r >> 2) +
        // 4 h of the 32-row block in register r: one store instruction writes two full 128-byte row segments (a transposed tile -
        // lane = row - would write sixty-four 16-byte pieces per instruction: measured 35 % slower as a launch of its own)
        const int row0 = tile * 256 + wave * 32 + 4 * fh;
        float* hcol = h1 + (size_t)row0 * FL_C + l31;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            f32x16 hacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                bf16x8 wb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const uint4 u = w1s[(((b * KST + s) * 3 + p) * 2 + fh) * 32 + l31];
                    wb[p] = *reinterpret_cast<const bf16x8*>(&u);
                }
                AG_MFMA_SPLIT(hacc, xq[s][0], xq[s][1], xq[s][2], wb[0], wb[1], wb[2]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (row0 + rr < M) hcol[(size_t)rr * FL_C + 32 * b] = fl_elu(hacc[r]);
            }
        }
    }
}

int fl_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

template <int KST>
int launch_first_layer(const float* obs, const double* mean, const double* var, float eps, float clip, const void* image, float* xn,
                       float* h1, int M, int D, void* stream) {
    static bool attr_set[64] = {};
    auto* fn = first_layer_kernel<KST>;
    constexpr size_t lds_bytes = (size_t)fl_image_units(KST) * 16;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AG_ERR_HIP;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AG_ERR_HIP;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int ntiles = (M + 255) / 256;
    const int grid = ntiles < fl_cus() ? ntiles : fl_cus();
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds_bytes, (hipStream_t)stream, obs, mean, var, eps, clip, (const uint4*)image, xn, h1, M, D);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

}  // namespace

#if AG_SPLIT_PLANES == 3
extern "C" int ag_mlp_first_layer_supported(int D, int C) { return (C == FL_C && D >= 1 && fl_ksteps(D) != 0) ? 1 : 0; }

extern "C" long long ag_mlp_first_layer_image_bytes(int D) { return (long long)fl_image_units(fl_ksteps(D) ? fl_ksteps(D) : 2) * 16; }

extern "C" int ag_mlp_first_layer_prepare(const float* W1_dev, const float* b1_dev, int D, void* image_dev, void* stream) {
    if (!W1_dev || !b1_dev || !image_dev) return AG_ERR_INVALID_ARG;
    if (!ag_mlp_first_layer_supported(D, FL_C)) return AG_ERR_UNSUPPORTED;
    if ((uintptr_t)image_dev & 15) return AG_ERR_INVALID_ARG;
    const int kst = fl_ksteps(D), threads = 8 * kst * 2 * 32;
    hipLaunchKernelGGL(first_layer_prepare_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1_dev, b1_dev, D, kst,
                       (uint4*)image_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
#endif

extern "C" int AG_PREC(ag_mlp_first_layer)(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
                                           const void* image_dev, float* xn_dev, float* h1_dev, int M, int D, void* stream) {
    if (!obs_dev || !image_dev || !h1_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if ((mean_dev == nullptr) != (var_dev == nullptr)) return AG_ERR_INVALID_ARG;
    if ((mean_dev == nullptr) != (xn_dev == nullptr)) return AG_ERR_INVALID_ARG;      // header: xn is written iff the input is normalised here
    if (!ag_mlp_first_layer_supported(D, FL_C)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)image_dev & 15) || ((uintptr_t)h1_dev & 15)) return AG_ERR_INVALID_ARG;
    if ((D & 1) == 0 && ((uintptr_t)obs_dev & 7)) return AG_ERR_INVALID_ARG;
    return fl_ksteps(D) == 2 ? launch_first_layer<2>(obs_dev, mean_dev, var_dev, eps, clip, image_dev, xn_dev, h1_dev, M, D, stream)
                             : launch_first_layer<4>(obs_dev, mean_dev, var_dev, eps, clip, image_dev, xn_dev, h1_dev, M, D, stream);
}
