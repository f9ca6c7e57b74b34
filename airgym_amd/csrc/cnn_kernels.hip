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

// sum over the wave, the same value in every lane: four DPP row rotations (no LDS round trips), then the four row sums through
// scalar registers; fixed order
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
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
// border bookkeeping of a value at element index idx of an [H][W] plane (see plane_border_sums_kernel): b = {row 0, last row, column 0,
// (0,0), (H-1,0)}.  (idx + 0.5) / W in float is exact for the plane sizes here (idx < 2^16, W >= 2).
__device__ __forceinline__ void border_add(float (&b)[5], float v, int idx, int H, int W, float inv_w) {
    const int oy = (int)(((float)idx + 0.5f) * inv_w), ox = idx - oy * W;
    b[0] += oy == 0 ? v : 0.f;
    b[1] += oy == H - 1 ? v : 0.f;
    b[2] += ox == 0 ? v : 0.f;
    b[3] += idx == 0 ? v : 0.f;
    b[4] += idx == (H - 1) * W ? v : 0.f;
}

template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ coef, const float* __restrict__ sums,
                                                             const float* __restrict__ wts, float* __restrict__ dx,
                                                             float* __restrict__ psum, float* __restrict__ bsum, int W,
                                                             long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    const int H = bsum ? HW / W : 0;
    const float inv_w = bsum ? 1.0f / (float)W : 0.f;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float mu = coef[c * 4 + 0], is = coef[c * 4 + 1], gi = coef[c * 4 + 2];
        const float rm = coef[c * 4 + 3] * (wts ? wts[p / C] : 1.0f);
        const float db = sums[c * 2 + 0] * rm, dg = sums[c * 2 + 1] * rm;
        float ps = 0.0f;                                    // sum of dx over the plane: the bias gradient of the convolution behind x
        float bs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};            // and its border sums (bsum)
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
                ps += g[e];
                if (bsum) border_add(bs, g[e], i * VEC + e, H, W, inv_w);
            }
            out[i] = d;
        }
        if (psum) {
            ps = wave_sum(ps);
            if (lane == 0) psum[p] = ps;
        }
        if (bsum) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float t = wave_sum(bs[j]);
                if (lane == 0) bsum[p * 5 + j] = t;
            }
        }
    }
}

// Last layer of the extractor: BatchNorm is followed by a global average pool (AdaptiveAvgPool2d((1, 1)), cnn.py:14), so the layer's
// output is only needed as plane means, which follow from the per-plane sums of relu(x) that the convolution's epilogue emits
// (conv_kernels.hip, `stats`): mean_hw(relu(x) scale + shift) = scale S1 / HW + shift.  The backward of that pool:
// the gradient of the pooled output reaches every pixel of a plane as the same number dyp[plane] (already
// divided by HW by the caller), so relu_bn_bwd_dx needs no dy tensor: dx = [x > 0] g (dyp - m_i db / m - m_i xhat dg / m).
// The result is affine in x on each plane: dx = [x > 0] (Bp x + Kp), Bp = -g invstd dg m_i / m, Kp = g (dyp - m_i db / m + invstd mean
// dg m_i / m).  The border sums re-read the two border rows and the first column of x (cache hits) instead of classifying every element
// on the way (0.40 -> 0.30 ms per 4 750 images at 64 x 27 x 15).
template <int VEC>
__global__ __launch_bounds__(256) void relu_bn_bwd_dx_plane_kernel(const float* __restrict__ dyp, const float* __restrict__ x,
                                                                   const float* __restrict__ coef, const float* __restrict__ sums,
                                                                   const float* __restrict__ wts, float* __restrict__ dx,
                                                                   float* __restrict__ psum, float* __restrict__ bsum, int W,
                                                                   long long planes, int C, int HW) {
    typedef typename VecT<VEC>::type vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    const int nvec = HW / VEC;
    const int H = bsum ? HW / W : 0;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const int c = (int)(p % C);
        const float mu = coef[c * 4 + 0], is = coef[c * 4 + 1], gi = coef[c * 4 + 2];
        const float rm = coef[c * 4 + 3] * (wts ? wts[p / C] : 1.0f);
        const float dg = sums[c * 2 + 1] * rm;
        const float d0 = dyp[p] - sums[c * 2 + 0] * rm;
        const float bp = -gi * is * dg, kp = gi * fmaf(is * mu, dg, d0);
        const vec_t* gx = reinterpret_cast<const vec_t*>(x + p * HW);
        vec_t* out = reinterpret_cast<vec_t*>(dx + p * HW);
        const float* xp = x + p * HW;
        auto val = [&](float t) { return t > 0.0f ? fmaf(t, bp, kp) : 0.0f; };
        float ps = 0.0f;
        // border values first (independent of everything else: all of a plane's loads are in flight together); lanes cover a row /
        // a column when W, H <= 64, else the loops below
        const bool small_border = bsum && W <= 64 && H <= 64;
        float t0 = 0.f, tl = 0.f, tc = 0.f;
        if (small_border) {
            if (lane < W) { t0 = xp[lane]; tl = xp[(H - 1) * W + lane]; }
            if (lane < H) tc = xp[lane * W];
        }
        if (nvec <= 8 * 64) {
            vec_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (lane + 64 * j < nvec) v[j] = gx[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (lane + 64 * j < nvec) {
                    float* f = reinterpret_cast<float*>(&v[j]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        f[e] = val(f[e]);
                        ps += f[e];
                    }
                    out[lane + 64 * j] = v[j];
                }
            }
        } else {
#pragma unroll 4
            for (int i = lane; i < nvec; i += 64) {
                vec_t v = gx[i];
                float* f = reinterpret_cast<float*>(&v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    f[e] = val(f[e]);
                    ps += f[e];
                }
                out[i] = v;
            }
        }
        if (psum) {
            ps = wave_sum(ps);
            if (lane == 0) psum[p] = ps;
        }
        if (bsum) {
            float r0 = 0.f, rl = 0.f, c0 = 0.f, k0, kl;
            if (small_border) {
                r0 = lane < W ? val(t0) : 0.f;
                rl = lane < W ? val(tl) : 0.f;
                c0 = lane < H ? val(tc) : 0.f;
                k0 = r0;            // lane 0: element (0, 0) / (H - 1, 0)
                kl = rl;
            } else {
                for (int ox = lane; ox < W; ox += 64) {
                    r0 += val(xp[ox]);
                    rl += val(xp[(H - 1) * W + ox]);
                }
                for (int oy = lane; oy < H; oy += 64) c0 += val(xp[oy * W]);
                k0 = val(xp[0]);
                kl = val(xp[(H - 1) * W]);
            }
            r0 = wave_sum(r0);
            rl = wave_sum(rl);
            c0 = wave_sum(c0);
            if (lane == 0) {
                bsum[p * 5 + 0] = r0;
                bsum[p * 5 + 1] = rl;
                bsum[p * 5 + 2] = c0;
                bsum[p * 5 + 3] = k0;
                bsum[p * 5 + 4] = kl;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The per-channel arithmetic between the big passes (each entry point replaces 10 - 30 launches of [C]-sized torch operations per
// layer and step).  Column sums run in two stages - kColBlocks workgroups over row ranges, then one workgroup over their float64
// partials - in a fixed order: deterministic.
constexpr int kColBlocks = 128, kSmallThreads = 1024;

// stage 1: part[block][W] (double) = sums over the block's rows of rows[r][col] * wts[r / G] (wts NULL = 1).  W <= 128.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ rows, const float* __restrict__ wts, long long R, int G, int W,
                                                     double* __restrict__ part) {
    __shared__ double s_part[256];
    const int t = threadIdx.x, stripes = 256 / W, col = t % W, stripe = t / W;
    const long long per = (R + gridDim.x - 1) / gridDim.x, r0 = (long long)blockIdx.x * per, r1 = (r0 + per < R) ? r0 + per : R;
    double acc = 0.0;
    if (stripe < stripes) {
        long long r = r0 + stripe;
        for (; r + 3LL * stripes < r1; r += 4LL * stripes) {        // four independent loads in flight
            const float v0 = rows[r * W + col], v1 = rows[(r + stripes) * W + col], v2 = rows[(r + 2 * stripes) * W + col],
                        v3 = rows[(r + 3 * stripes) * W + col];
            const float w0 = wts ? wts[r / G] : 1.f, w1 = wts ? wts[(r + stripes) / G] : 1.f, w2 = wts ? wts[(r + 2 * stripes) / G] : 1.f,
                        w3 = wts ? wts[(r + 3 * stripes) / G] : 1.f;
            acc += (double)v0 * (double)w0;
            acc += (double)v1 * (double)w1;
            acc += (double)v2 * (double)w2;
            acc += (double)v3 * (double)w3;
        }
        for (; r < r1; r += stripes) acc += (double)rows[r * W + col] * (double)(wts ? wts[r / G] : 1.f);
    }
    s_part[t] = acc;
    __syncthreads();
    if (t < W) {
        double tot = 0.0;
        for (int k = 0; k < stripes; ++k) tot += s_part[k * W + t];
        part[(size_t)blockIdx.x * W + t] = tot;
    }
}

// stage 1 of the pooled layer's backward: part[block][2c] = sum_n a[n][c], part[block][2c + 1] = sum_n a[n][c] b[n][c]; also
// dyp = a * inv_hw.  C <= 64.
__global__ __launch_bounds__(256) void colsum_pair_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, int C,
                                                          float inv_hw, double* __restrict__ part, float* __restrict__ dyp) {
    __shared__ double s_part[2][256];
    const int t = threadIdx.x, stripes = 256 / C, c = t % C, stripe = t / C;
    const long long per = (n + gridDim.x - 1) / gridDim.x, r0 = (long long)blockIdx.x * per, r1 = (r0 + per < n) ? r0 + per : n;
    double a0 = 0.0, a1 = 0.0;
    if (stripe < stripes) {
        for (long long r = r0 + stripe; r < r1; r += stripes) {
            const float av = a[r * C + c], bv = b[r * C + c];
            dyp[r * C + c] = av * inv_hw;
            a0 += (double)av;
            a1 += (double)av * (double)bv;
        }
    }
    s_part[0][t] = a0;
    s_part[1][t] = a1;
    __syncthreads();
    if (t < 2 * C) {
        const int cc = t >> 1, j = t & 1;
        double tot = 0.0;
        for (int k = 0; k < stripes; ++k) tot += s_part[j][k * C + cc];
        part[(size_t)blockIdx.x * 2 * C + t] = tot;
    }
}

// stage 2 helper: tot[W] (LDS, double) = sum over the kColBlocks partial rows
__device__ void sum_partials(const double* __restrict__ part, int W, double* s_part, double* s_tot) {
    const int t = threadIdx.x, stripes = kSmallThreads / W, col = t % W, stripe = t / W;
    double acc = 0.0;
    for (int r = stripe; r < kColBlocks; r += stripes) acc += part[(size_t)r * W + col];
    s_part[t] = acc;
    __syncthreads();
    if (t < W) {
        double tot = 0.0;
        for (int k = 0; k < stripes; ++k) tot += s_part[k * W + t];
        s_tot[t] = tot;
    }
    __syncthreads();
}

// forward: coef [4][C] = {mean, invstd, scale = gamma invstd, shift = beta - mean scale} from the batch sums (training; running
// statistics updated as nn.BatchNorm2d does: momentum, unbiased variance) or from the running statistics (eval)
__global__ __launch_bounds__(kSmallThreads) void bn_finalize_kernel(const double* __restrict__ part, int C, double m,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   long long* __restrict__ num_batches, float momentum, double eps,
                                                                   int training, float* __restrict__ coef) {
    __shared__ double s_part[kSmallThreads], s_tot[128];
    const int t = threadIdx.x;
    if (training) sum_partials(part, 2 * C, s_part, s_tot);
    if (t < C) {
        double mean, var;
        if (training) {
            mean = s_tot[2 * t] / m;
            var = s_tot[2 * t + 1] / m - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const double unbiased = var * (m / (m - 1.0 > 1.0 ? m - 1.0 : 1.0));
            running_mean[t] = (float)((1.0 - (double)momentum) * (double)running_mean[t] + (double)momentum * mean);
            running_var[t] = (float)((1.0 - (double)momentum) * (double)running_var[t] + (double)momentum * unbiased);
        } else {
            mean = (double)running_mean[t];
            var = (double)running_var[t];
        }
        const double invstd = 1.0 / sqrt(var + eps);
        const double scale = (double)gamma[t] * invstd, shift = (double)beta[t] - mean * scale;
        coef[t] = (float)mean;
        coef[C + t] = (float)invstd;
        coef[2 * C + t] = (float)scale;
        coef[3 * C + t] = (float)shift;
    }
    if (t == 0 && training && num_batches) *num_batches += 1;
}

// last layer: plane1 [n][C] = per-image sum of relu(y) (over the G bands), pooled = scale plane1 / HW + shift
__global__ __launch_bounds__(256) void bn_pool_kernel(const float* __restrict__ stats, long long total, int G, int C,
                                                      const float* __restrict__ coef, float inv_hw, float* __restrict__ plane1,
                                                      float* __restrict__ pooled) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= total) return;
    const long long i = u / C;
    const int c = (int)(u - i * C);
    float s1 = 0.f;
    for (int g = 0; g < G; ++g) s1 += stats[((i * G + g) * C + c) * 2];
    plane1[u] = s1;
    pooled[u] = coef[2 * C + c] * (s1 * inv_hw) + coef[3 * C + c];
}

// backward: sums [C][2] = {dbeta, dgamma} and the table the next kernel wants.  mode 0 / 1: part = column sums of
// ag_relu_bn_bwd_reduce's partials; mode 0: tab {mean, invstd, gamma invstd, 1 / m} (ag_relu_bn_bwd_dx), mode 1: tab {A, B, C, 0}
// (ag_cnn_conv1_wgrad: A = gamma invstd, B = -A invstd dgamma / m, C = A (invstd mean dgamma - dbeta) / m).  mode 2 (pooled layer):
// part = {sum_n dpool, sum_n dpool plane1}: dbeta = the first, dgamma = invstd (inv_hw the second - mean the first); tab as mode 0.
__global__ __launch_bounds__(kSmallThreads) void bn_bwd_prep_kernel(const double* __restrict__ part, int C, const float* __restrict__ coef_fwd,
                                                                   const float* __restrict__ gamma, double m, int mode, double inv_hw,
                                                                   float* __restrict__ sums, float* __restrict__ tab,
                                                                   float* __restrict__ dgamma_out, float* __restrict__ dbeta_out) {
    __shared__ double s_part[kSmallThreads], s_tot[128];
    const int t = threadIdx.x;
    sum_partials(part, 2 * C, s_part, s_tot);
    if (t < C) {
        const double mean = (double)coef_fwd[t], invstd = (double)coef_fwd[C + t];
        double db = s_tot[2 * t], dg = s_tot[2 * t + 1];
        if (mode == 2) dg = invstd * (inv_hw * dg - mean * db);
        sums[2 * t] = (float)db;
        sums[2 * t + 1] = (float)dg;
        if (dgamma_out) {           // the parameter gradients written where the optimizer reads them (no accumulation launch)
            dgamma_out[t] = (float)dg;
            dbeta_out[t] = (float)db;
        }
        const double a = (double)gamma[t] * invstd;
        if (mode != 1) {
            tab[4 * t + 0] = (float)mean;
            tab[4 * t + 1] = (float)invstd;
            tab[4 * t + 2] = (float)a;
            tab[4 * t + 3] = (float)(1.0 / m);
        } else {
            // the float32 sums are what ag_relu_bn_bwd_dx would read: keep the two paths on the same inputs
            const double dbf = (double)(float)db, dgf = (double)(float)dg;
            tab[4 * t + 0] = (float)a;
            tab[4 * t + 1] = (float)(-a * invstd * dgf / m);
            tab[4 * t + 2] = (float)(a * (invstd * mean * dgf - dbf) / m);
            tab[4 * t + 3] = 0.f;
        }
    }
}

// Weighted per-pixel moments of a batch of images for the input normaliser (reference: RunningMeanStd.forward -> update on the
// image observation, running_mean_std.py:34-60): partial[chunk][2][D] (double) = sums over the chunk's rows of w x and w x^2, rows
// read through `index` straight out of the frame store (NULL = identity), w = image multiplicity (NULL = 1).  One pass over the
// images; the torch formulation (float64 temporaries of the whole batch, two passes) was ~1.5 ms per 4 750 images.
constexpr int kMomentChunks = 32;
__global__ __launch_bounds__(256) void wide_moments_kernel(const float* __restrict__ x, const long long* __restrict__ index,
                                                           const float* __restrict__ wts, long long rows, long long D,
                                                           double* __restrict__ partial) {
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    const int chunk = blockIdx.y;
    double s = 0.0, q = 0.0;
    long long r = chunk;
    for (; r + 3LL * kMomentChunks < rows; r += 4LL * kMomentChunks) {
        float v[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long rr = r + (long long)k * kMomentChunks;
            v[k] = x[(index ? index[rr] : rr) * D + col];
            w[k] = wts ? wts[rr] : 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double wv = (double)w[k] * (double)v[k];
            s += wv;
            q += wv * (double)v[k];
        }
    }
    for (; r < rows; r += kMomentChunks) {
        const double v = (double)x[(index ? index[r] : r) * D + col], w = wts ? (double)wts[r] : 1.0;
        s += w * v;
        q += w * v * v;
    }
    partial[((size_t)chunk * 2 + 0) * D + col] = s;
    partial[((size_t)chunk * 2 + 1) * D + col] = q;
}

// Border sums of a gradient tensor dz [planes][H][W]: out[plane] = {sum of row 0, sum of row H-1, sum of column 0, dz[0][0], dz[H-1][0]}.
// Why: the two reductions of a ReLU + BatchNorm backward (sum dy, sum dy * xhat over all pixels) need no pass over dy when the layer
// feeds a convolution: with y = the BatchNorm output (the convolution's zero-padded input), dy = conv^T(dz) and dw = the weight gradient,
//     sum_p dy[ci,p] y[ci,p] = sum_{co,tap} w[co,ci,tap] dw[co,ci,tap]        (both sides are the same bilinear form)
//     sum_p dy[ci,p]         = sum_{co,tap} w[co,ci,tap] S[co,tap],   S[co,tap] = sum of dz[co] over the output pixels whose tap lies
//                                                                       inside the image = total minus these border sums
// and xhat = (y - beta) / gamma.  One wave per plane; reads 2 rows + 1 column.
__global__ __launch_bounds__(256) void plane_border_sums_kernel(const float* __restrict__ dz, float* __restrict__ out, long long planes,
                                                                int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p0 = (long long)blockIdx.x * kPlanesPerBlock;
    for (int k = wave; k < kPlanesPerBlock; k += 4) {
        const long long p = p0 + k;
        if (p >= planes) break;
        const float* z = dz + p * H * W;
        float r0 = 0.f, rl = 0.f, c0 = 0.f;
        for (int ox = lane; ox < W; ox += 64) {
            r0 += z[ox];
            rl += z[(size_t)(H - 1) * W + ox];
        }
        for (int oy = lane; oy < H; oy += 64) c0 += z[(size_t)oy * W];
        r0 = wave_sum(r0);
        rl = wave_sum(rl);
        c0 = wave_sum(c0);
        if (lane == 0) {
            out[p * 5 + 0] = r0;
            out[p * 5 + 1] = rl;
            out[p * 5 + 2] = c0;
            out[p * 5 + 3] = z[0];
            out[p * 5 + 4] = z[(size_t)(H - 1) * W];
        }
    }
}

// The identities above in one workgroup: from total [CO] (sum of dz per output channel = the convolution's bias gradient) and border
// [CO][5] (plane_border_sums_kernel's rows summed over the images) S[co][tap] is formed, and for every input channel c
//     part[c] = { sum_dy = sum_{co,tap} w S,   (sum_{co,tap} w dw - beta[c] sum_dy) / gamma[c] }     (float64 accumulation)
// which is what ag_relu_bn_bwd_reduce + the column sums would have produced for the ReLU + BatchNorm in front of the convolution.
__global__ __launch_bounds__(kSmallThreads) void bn_sums_from_conv_kernel(const float* __restrict__ w, const float* __restrict__ dw,
                                                                         const float* __restrict__ total,
                                                                         const float* __restrict__ border, int CO, int C,
                                                                         int last_row_out, const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, float* __restrict__ part) {
    __shared__ double s_b[64][6];            // per output channel: total, row 0, last row, column 0, corner (0,0), corner (last,0)
    __shared__ double s_S[64][9];
    __shared__ double s_acc[kSmallThreads][2];
    const int t = threadIdx.x;
    if (t < CO * 6) {
        const int co = t / 6, k = t - co * 6;
        s_b[co][k] = (double)(k == 0 ? total[co] : border[co * 5 + (k - 1)]);
    }
    __syncthreads();
    if (t < CO * 9) {
        const int co = t / 9, tap = t - co * 9, ky = tap / 3, kx = tap - ky * 3;
        double v = s_b[co][0];
        if (ky == 0) v -= s_b[co][1];
        if (kx == 0) v -= s_b[co][3];
        if (ky == 0 && kx == 0) v += s_b[co][4];
        if (last_row_out && ky == 2) {
            v -= s_b[co][2];
            if (kx == 0) v += s_b[co][5];
        }
        s_S[co][tap] = v;
    }
    __syncthreads();
    // stage 2: thread = (c, stripe over co)
    const int stripes = kSmallThreads / C, c = t % C, stripe = t / C;
    double a0 = 0.0, a1 = 0.0;
    if (stripe < stripes) {
        for (int co = stripe; co < CO; co += stripes) {
            const float* wp = w + ((size_t)co * C + c) * 9;
            const float* dp = dw + ((size_t)co * C + c) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const double wv = (double)wp[tap];
                a0 += wv * s_S[co][tap];
                a1 += wv * (double)dp[tap];
            }
        }
    }
    s_acc[t][0] = a0;
    s_acc[t][1] = a1;
    __syncthreads();
    if (t < C) {
        double sd = 0.0, sy = 0.0;
        for (int k = 0; k < stripes; ++k) {
            sd += s_acc[k * C + t][0];
            sy += s_acc[k * C + t][1];
        }
        double g = (double)gamma[t];
        if (g > -1e-30 && g < 1e-30) g = 1e-30;
        part[2 * t] = (float)sd;
        part[2 * t + 1] = (float)((sy - (double)beta[t] * sd) / g);
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
                                          const float* weights_dev, float* dx_dev, float* plane_sums_dev, float* border_sums_dev,
                                          int W, int N, int C, int HW, void* stream) {
    if (!dy_dev || !x_dev || !coef_dev || !sums_dev || !dx_dev || (border_sums_dev && (W <= 1 || HW % W != 0 || HW >= 65536)))
        return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, dy_dev, dx_dev, HW);
    AG_BN_DISPATCH(relu_bn_bwd_dx_kernel, w, dy_dev, x_dev, coef_dev, sums_dev, weights_dev, dx_dev, plane_sums_dev, border_sums_dev, W,
                   planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_relu_bn_bwd_dx(const float* dy_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                                 float* dx_dev, int N, int C, int HW, void* stream) {
    return ag_relu_bn_bwd_dx_weighted(dy_dev, x_dev, coef_dev, sums_dev, nullptr, dx_dev, nullptr, nullptr, 0, N, C, HW, stream);
}

extern "C" int ag_relu_bn_bwd_dx_plane(const float* dyp_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                                       const float* weights_dev, float* dx_dev, float* plane_sums_dev, float* border_sums_dev, int W,
                                       int N, int C, int HW, void* stream) {
    if (!dyp_dev || !x_dev || !coef_dev || !sums_dev || !dx_dev || dx_dev == x_dev ||
        (border_sums_dev && (W <= 1 || HW % W != 0 || HW >= 65536)))
        return AG_ERR_INVALID_ARG;          // (not in place: the border sums read x again)
    AG_BN_CHECK(N, C, HW);
    const int w = vec_width(x_dev, dx_dev, nullptr, HW);
    AG_BN_DISPATCH(relu_bn_bwd_dx_plane_kernel, w, dyp_dev, x_dev, coef_dev, sums_dev, weights_dev, dx_dev, plane_sums_dev,
                   border_sums_dev, W, planes, C, HW);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" long long ag_bn_scratch_doubles(void) { return (long long)kColBlocks * 128; }

extern "C" int ag_bn_finalize(const float* stats_dev, const float* weights_dev, long long n, int G, int C, double m,
                              const float* gamma_dev, const float* beta_dev, float* running_mean_dev, float* running_var_dev,
                              long long* num_batches_dev, float momentum, double eps, int training, float* coef_dev,
                              float* plane1_dev, float* pooled_dev, int HW, double* scratch_dev, void* stream) {
    if (!gamma_dev || !beta_dev || !running_mean_dev || !running_var_dev || !coef_dev || (training && (!stats_dev || !scratch_dev)) ||
        n <= 0 || G <= 0 || (!plane1_dev) != (!pooled_dev) || (pooled_dev && !stats_dev) || HW <= 0)
        return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > kMaxC || (256 % (2 * C)) != 0) return AG_ERR_UNSUPPORTED;
    if (training)
        hipLaunchKernelGGL(colsum_kernel, dim3(kColBlocks), dim3(256), 0, (hipStream_t)stream, stats_dev, weights_dev, n * G, G, 2 * C,
                           scratch_dev);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(kSmallThreads), 0, (hipStream_t)stream, scratch_dev, C, m, gamma_dev, beta_dev,
                       running_mean_dev, running_var_dev, num_batches_dev, momentum, eps, training, coef_dev);
    if (pooled_dev) {
        const long long total = n * C;
        hipLaunchKernelGGL(bn_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats_dev, total, G, C,
                           coef_dev, 1.0f / (float)HW, plane1_dev, pooled_dev);
    }
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_bn_bwd_prep(const float* partials_dev, long long blocks, int C, const float* coef_fwd_dev, const float* gamma_dev,
                              double m, int mode, float* sums_dev, float* tab_dev, float* dgamma_out_dev, float* dbeta_out_dev,
                              double* scratch_dev, void* stream) {
    if (!partials_dev || !coef_fwd_dev || !gamma_dev || !sums_dev || !tab_dev || !scratch_dev || blocks <= 0 || (mode != 0 && mode != 1) ||
        (!dgamma_out_dev) != (!dbeta_out_dev))
        return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > kMaxC || (256 % (2 * C)) != 0) return AG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(colsum_kernel, dim3(kColBlocks), dim3(256), 0, (hipStream_t)stream, partials_dev, (const float*)nullptr, blocks, 1,
                       2 * C, scratch_dev);
    hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(1), dim3(kSmallThreads), 0, (hipStream_t)stream, scratch_dev, C, coef_fwd_dev, gamma_dev, m,
                       mode, 0.0, sums_dev, tab_dev, dgamma_out_dev, dbeta_out_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_bn_pool_bwd_prep(const float* dpool_dev, const float* plane1_dev, long long n, int C, const float* coef_fwd_dev,
                                   const float* gamma_dev, double m, int HW, float* sums_dev, float* tab_dev, float* dyp_dev,
                                   float* dgamma_out_dev, float* dbeta_out_dev, double* scratch_dev, void* stream) {
    if (!dpool_dev || !plane1_dev || !coef_fwd_dev || !gamma_dev || !sums_dev || !tab_dev || !dyp_dev || !scratch_dev || n <= 0 || HW <= 0 ||
        (!dgamma_out_dev) != (!dbeta_out_dev))
        return AG_ERR_INVALID_ARG;
    if (C <= 0 || C > kMaxC || (256 % C) != 0) return AG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(colsum_pair_kernel, dim3(kColBlocks), dim3(256), 0, (hipStream_t)stream, dpool_dev, plane1_dev, n, C,
                       1.0f / (float)HW, scratch_dev, dyp_dev);
    hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(1), dim3(kSmallThreads), 0, (hipStream_t)stream, scratch_dev, C, coef_fwd_dev, gamma_dev, m,
                       2, 1.0 / (double)HW, sums_dev, tab_dev, dgamma_out_dev, dbeta_out_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_weighted_moments_chunks(void) { return kMomentChunks; }

extern "C" int ag_weighted_moments(const float* x_dev, const long long* index_dev, const float* weights_dev, long long rows, long long D,
                                   double* partial_dev, void* stream) {
    if (!x_dev || !partial_dev || rows <= 0 || D <= 0) return AG_ERR_INVALID_ARG;
    if ((D + 255) / 256 > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wide_moments_kernel, dim3((unsigned)((D + 255) / 256), kMomentChunks), dim3(256), 0, (hipStream_t)stream, x_dev,
                       index_dev, weights_dev, rows, D, partial_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_plane_border_sums(const float* dz_dev, float* out_dev, int N, int C, int H, int W, void* stream) {
    if (!dz_dev || !out_dev || H <= 0 || W <= 0) return AG_ERR_INVALID_ARG;
    AG_BN_CHECK(N, C, H * W);
    hipLaunchKernelGGL(plane_border_sums_kernel, grid, block, 0, (hipStream_t)stream, dz_dev, out_dev, planes, H, W);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_bn_sums_from_conv(const float* w_dev, const float* dw_dev, const float* total_dev, const float* border_dev, int cout,
                                    int cin, int hin, const float* gamma_dev, const float* beta_dev, float* part_dev, void* stream) {
    if (!w_dev || !dw_dev || !total_dev || !border_dev || !gamma_dev || !beta_dev || !part_dev || hin <= 0) return AG_ERR_INVALID_ARG;
    if (cout <= 0 || cout > 64 || cin <= 0 || cin > 64) return AG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_sums_from_conv_kernel, dim3(1), dim3(kSmallThreads), 0, (hipStream_t)stream, w_dev, dw_dev, total_dev, border_dev,
                       cout, cin, hin & 1, gamma_dev, beta_dev, part_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
