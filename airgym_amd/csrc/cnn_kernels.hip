// cnn_kernels.hip - ReLU followed by BatchNorm2d for the depth-image feature extractor of the Planning policy
// (reference: lib/network/cnn.py:3-33, three times Conv2d -> ReLU -> BatchNorm2d on [N, C, H, W] float32, C = 16 / 32 / 64).
//
// Why: at the Planning scale (16 384-image minibatches) these activations are 1.7 - 6.7 GB per tensor.  The library path is
// an elementwise ReLU (read + write), then MIOpen's spatial batch norm, whose kernels launch 16 - 64 workgroups for these
// shapes: 45 % of a PPO update (profiles/r02_planning_cnn_miopen_kernel_trace.md).  Here ReLU is folded into the batch-norm
// passes (its output is never materialised; the backward recomputes it from the convolution output), and every pass is
// spread over the whole chip: one WAVE per (image, channel) plane - contiguous H*W floats in NCHW - so loads are coalesced and
// the per-channel constants are wave-uniform.
//
//   forward, training:  ag_relu_bn_stats  : per-channel sum / sum of squares of relu(x)         (1 read)   -> partials
//                       (host: mean, biased var -> invstd; running stats updated as nn.BatchNorm2d does)
//                       ag_relu_bn_apply  : y = (relu(x) - mean) * invstd * gamma + beta          (1 read, 1 write)
//   forward, eval:      ag_relu_bn_apply with the running statistics
//   backward:           ag_relu_bn_bwd_reduce : dbeta = sum dy, dgamma = sum dy * xhat            (2 reads)  -> partials
//                       ag_relu_bn_bwd_dx     : dx = [x > 0] gamma invstd (dy - dbeta/m - xhat dgamma/m)   (2 reads, 1 write)
// Partials are [blocks, C, 2] (fixed order; the caller sums over dim 0 in float64 -> deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"

namespace {

constexpr int kPlanesPerBlock = 64;       // (image, channel) planes per workgroup: 4 waves x 16 planes each
constexpr int kMaxC = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wave walks plane p (HW contiguous floats); F(value index i, pointer offset) is applied through the vector width that the
// plane's alignment allows: float4 when HW % 4 == 0, float2 when HW % 2 == 0, scalar otherwise
template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<1> { typedef float type; };

// `wts` (optional, [N]): per-image multiplicities.  A minibatch that holds image i m_i times (the depth camera runs every 4th env
// step, so 3 of 4 consecutive rollout samples of an env carry the same image) has the batch statistics of the DISTINCT images
// weighted by m_i; the update then runs the convolutions on the distinct images only (lib/network/fused_relu_bn.py).
template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ wts,
                                                            float* __restrict__ partials, long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    __shared__ float acc[4][kMaxC][2];                      // one accumulator set per wave: fixed summation order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * kMaxC * 2; i += 256) (&acc[0][0][0])[i] = 0.0f;
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const vec_t* src = reinterpret_cast<const vec_t*>(x + p * HW);
        float s = 0.0f, q = 0.0f;
#pragma unroll 2
        for (int i = lane; i < nvec; i += 64) {
            const vec_t v = src[i];
            const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float r = fmaxf(f[e], 0.0f);
                s += r;
                q = fmaf(r, r, q);
            }
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) {
            const int c = (int)(p % C);
            const float wi = wts ? wts[p / C] : 1.0f;
            acc[wave][c][0] += wi * s;
            acc[wave][c][1] += wi * q;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 2; i += 256) {
        const int c = i >> 1, j = i & 1;
        partials[(size_t)blockIdx.x * C * 2 + i] = (acc[0][c][j] + acc[1][c][j]) + (acc[2][c][j] + acc[3][c][j]);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y,
                                                            long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float a = scale[c], b = shift[c];             // y = relu(x) * a + b,  a = gamma invstd,  b = beta - mean a
        const vec_t* src = reinterpret_cast<const vec_t*>(x + p * HW);
        vec_t* dst = reinterpret_cast<vec_t*>(y + p * HW);
#pragma unroll 2
        for (int i = lane; i < nvec; i += 64) {
            vec_t v = src[i];
            float* f = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = fmaf(fmaxf(f[e], 0.0f), a, b);
            dst[i] = v;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ partials, long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    __shared__ float acc[4][kMaxC][2];                      // one accumulator set per wave: fixed summation order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * kMaxC * 2; i += 256) (&acc[0][0][0])[i] = 0.0f;
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float mu = mean[c], is = invstd[c];
        const vec_t* gx = reinterpret_cast<const vec_t*>(x + p * HW);
        const vec_t* gd = reinterpret_cast<const vec_t*>(dy + p * HW);
        float s = 0.0f, q = 0.0f;
#pragma unroll 2
        for (int i = lane; i < nvec; i += 64) {
            const vec_t v = gx[i], d = gd[i];
            const float* f = reinterpret_cast<const float*>(&v);
            const float* g = reinterpret_cast<const float*>(&d);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xhat = (fmaxf(f[e], 0.0f) - mu) * is;
                s += g[e];
                q = fmaf(g[e], xhat, q);
            }
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) {
            acc[wave][c][0] += s;
            acc[wave][c][1] += q;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 2; i += 256) {
        const int c = i >> 1, j = i & 1;
        partials[(size_t)blockIdx.x * C * 2 + i] = (acc[0][c][j] + acc[1][c][j]) + (acc[2][c][j] + acc[3][c][j]);
    }
}

// coef [C][4] = {mean, invstd, gamma * invstd, 1 / m} ; sums [C][2] = {dbeta, dgamma}
// With multiplicities (`wts`, see relu_bn_stats_kernel) dy is the gradient SUMMED over the copies of an image and the two mean
// terms, which every copy receives, are scaled by the image's multiplicity: dx_i = [x > 0] g (dy_i - m_i db / m - m_i xhat dg / m).
template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ coef, const float* __restrict__ sums,
                                                             const float* __restrict__ wts, float* __restrict__ dx,
                                                             long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float mu = coef[c * 4 + 0], is = coef[c * 4 + 1], gi = coef[c * 4 + 2];
        const float rm = coef[c * 4 + 3] * (wts ? wts[p / C] : 1.0f);
        const float db = sums[c * 2 + 0] * rm, dg = sums[c * 2 + 1] * rm;
        const vec_t* gx = reinterpret_cast<const vec_t*>(x + p * HW);
        const vec_t* gd = reinterpret_cast<const vec_t*>(dy + p * HW);
        vec_t* out = reinterpret_cast<vec_t*>(dx + p * HW);
#pragma unroll 2
        for (int i = lane; i < nvec; i += 64) {
            const vec_t v = gx[i];
            vec_t d = gd[i];
            const float* f = reinterpret_cast<const float*>(&v);
            float* g = reinterpret_cast<float*>(&d);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xhat = (fmaxf(f[e], 0.0f) - mu) * is;
                const float dr = gi * (g[e] - db - xhat * dg);
                g[e] = f[e] > 0.0f ? dr : 0.0f;              // ReLU' at 0 is 0, as torch's threshold_backward
            }
            out[i] = d;
        }
    }
}

// Last layer of the extractor: BatchNorm is followed by a global average pool (AdaptiveAvgPool2d((1, 1)), cnn.py:14), so the layer's
// output is only needed as plane means, which follow from the per-plane sums of relu(x) that the convolution's epilogue emits
// (conv_kernels.hip, `stats`): mean_hw(relu(x) scale + shift) = scale S1 / HW + shift.  The backward of that pool:
// the gradient of the pooled output reaches every pixel of a plane as the same number dyp[plane] (already
// divided by HW by the caller), so relu_bn_bwd_dx needs no dy tensor: dx = [x > 0] g (dyp - m_i db / m - m_i xhat dg / m).
template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_bwd_dx_plane_kernel(const float* __restrict__ dyp, const float* __restrict__ x,
                                                                   const float* __restrict__ coef, const float* __restrict__ sums,
                                                                   const float* __restrict__ wts, float* __restrict__ dx,
                                                                   long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float mu = coef[c * 4 + 0], is = coef[c * 4 + 1], gi = coef[c * 4 + 2];
        const float rm = coef[c * 4 + 3] * (wts ? wts[p / C] : 1.0f);
        const float dg = sums[c * 2 + 1] * rm;
        const float d0 = dyp[p] - sums[c * 2 + 0] * rm;
        const vec_t* gx = reinterpret_cast<const vec_t*>(x + p * HW);
        vec_t* out = reinterpret_cast<vec_t*>(dx + p * HW);
#pragma unroll 2
        for (int i = lane; i < nvec; i += 64) {
            vec_t v = gx[i];
            float* f = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xhat = (fmaxf(f[e], 0.0f) - mu) * is;
                f[e] = f[e] > 0.0f ? gi * (d0 - xhat * dg) : 0.0f;
            }
            out[i] = v;
        }
    }
}

int vec_width(const void* a, const void* b, const void* c, int HW) {
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    if ((HW & 3) == 0 && (bits & 15) == 0) return 4;
    if ((HW & 1) == 0 && (bits & 7) == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int ag_relu_bn_planes_per_block(void) { return kPlanesPerBlock; }

#define AG_BN_CHECK(N_, C_, HW_)                                                                   \
    if ((N_) <= 0 || (C_) <= 0 || (HW_) <= 0) return AG_ERR_INVALID_ARG;                           \
    if ((C_) > kMaxC) return AG_ERR_UNSUPPORTED;                                                   \
    const long long planes = (long long)(N_) * (C_);                                               \
    const long long nblocks = (planes + kPlanesPerBlock - 1) / kPlanesPerBlock;                    \
    if (nblocks > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;                                         \
    const dim3 grid((unsigned)nblocks), block(256)

#define AG_BN_DISPATCH(KERNEL, W, ...)                                                             \
    do {                                                                                           \
        if ((W) == 4) hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);      \
        else if ((W) == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);               \
    } while (0)

extern "C" int ag_relu_bn_stats_weighted(const float* x_dev, const float* weights_dev, float* partials_dev, int N, int C, int HW,
                                         void* stream) {
    if (!x_dev || !partials_dev) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, nullptr, nullptr, HW);
    AG_BN_DISPATCH(relu_bn_stats_kernel, w, x_dev, weights_dev, partials_dev, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_relu_bn_stats(const float* x_dev, float* partials_dev, int N, int C, int HW, void* stream) {
    return ag_relu_bn_stats_weighted(x_dev, nullptr, partials_dev, N, C, HW, stream);
}

extern "C" int ag_relu_bn_apply(const float* x_dev, const float* scale_dev, const float* shift_dev, float* y_dev, int N, int C,
                                int HW, void* stream) {
    if (!x_dev || !scale_dev || !shift_dev || !y_dev) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, y_dev, nullptr, HW);
    AG_BN_DISPATCH(relu_bn_apply_kernel, w, x_dev, scale_dev, shift_dev, y_dev, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_relu_bn_bwd_reduce(const float* dy_dev, const float* x_dev, const float* mean_dev, const float* invstd_dev,
                                     float* partials_dev, int N, int C, int HW, void* stream) {
    if (!dy_dev || !x_dev || !mean_dev || !invstd_dev || !partials_dev) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, dy_dev, nullptr, HW);
    AG_BN_DISPATCH(relu_bn_bwd_reduce_kernel, w, dy_dev, x_dev, mean_dev, invstd_dev, partials_dev, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_relu_bn_bwd_dx_weighted(const float* dy_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                                          const float* weights_dev, float* dx_dev, int N, int C, int HW, void* stream) {
    if (!dy_dev || !x_dev || !coef_dev || !sums_dev || !dx_dev) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, dy_dev, dx_dev, HW);
    AG_BN_DISPATCH(relu_bn_bwd_dx_kernel, w, dy_dev, x_dev, coef_dev, sums_dev, weights_dev, dx_dev, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_relu_bn_bwd_dx(const float* dy_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                                 float* dx_dev, int N, int C, int HW, void* stream) {
    return ag_relu_bn_bwd_dx_weighted(dy_dev, x_dev, coef_dev, sums_dev, nullptr, dx_dev, N, C, HW, stream);
}

extern "C" int ag_relu_bn_bwd_dx_plane(const float* dyp_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                                       const float* weights_dev, float* dx_dev, int N, int C, int HW, void* stream) {
    if (!dyp_dev || !x_dev || !coef_dev || !sums_dev || !dx_dev) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, dx_dev, nullptr, HW);
    AG_BN_DISPATCH(relu_bn_bwd_dx_plane_kernel, w, dyp_dev, x_dev, coef_dev, sums_dev, weights_dev, dx_dev, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
