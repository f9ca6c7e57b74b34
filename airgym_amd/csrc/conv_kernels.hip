// conv_kernels.hip - the three stride-2 convolutions of the depth-image feature extractor of the Planning policy, forward,
// input gradient and weight gradient, on NCHW float32 tensors as torch holds them
// (reference: lib/network/cnn.py:3-33: Conv2d(1,16,5,s2,p2) -> Conv2d(16,32,3,s2,p1) -> Conv2d(32,64,3,s2,p1) on (1,212,120) images).
//
// Why: with the library (MIOpen fp32) every convolution call of a PPO minibatch is an NHWC implicit GEMM wrapped in
// NCHW<->NHWC transposes of 0.5 - 1.9 GB activations; the transposes alone are 8.6 ms of a 31.7 ms minibatch step and the
// kernels another 12.6 ms (profiles/r03_planning_cnn_dedup_kernel_trace.md).  These kernels read and write NCHW directly.
//
// Arithmetic: exact float32.  The two 3x3 layers run on the f32-input matrix instruction v_mfma_f32_16x16x4_f32 (each product
// and accumulation is an fmaf; 157 TFLOP/s peak = the f32 vector rate, with no VALU issue slots spent on it); the 5x5 first
// layer has one input channel and 16 outputs, is bound by the 1.9 GB it writes, and its forward runs on the vector ALU with the
// weights in scalar registers; its weight gradient (K = all output pixels) runs on the same MFMA.
//
// Common scheme of the stride-2 kernels: a workgroup stages a band of input rows in LDS with the columns DE-INTERLEAVED by
// parity (c = ix + pad; E[j] = column 2j, O[j] = column 2j + 1), so that the stride-2 taps of 16 neighbouring output pixels are
// 16 consecutive floats of E or O (conflict-free ds_read_b32, the 16x16x4 operand layout: lane l supplies element
// [l & 15][k = l >> 4]).  Plane / row strides are chosen so that the two 16-lane quarters of a 32-lane LDS group fall into
// different banks (stride = 16 mod 32, or odd strides with a +16 quarter offset).  ReLU + BatchNorm of the PREVIOUS layer can be
// applied while staging (y = max(x, 0) * scale[c] + shift[c]; padding stays 0), so the normalised activation never exists in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "split_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AG_MFMA4(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// Staging loads go through buffer descriptors: a 32-bit byte offset per lane and hardware bounds checking - an out-of-range
// offset (kOob) returns 0, which is how padding rows, band tails and tail units are produced without a branch per load.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float2 buf_f2(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
// the same with a wave-uniform byte offset added by the hardware (soffset; not part of the bounds check, so kOob stays out of range)
__device__ __forceinline__ float buf_f1(__amdgpu_buffer_rsrc_t r, unsigned off, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
__device__ __forceinline__ float2 buf_f2(__amdgpu_buffer_rsrc_t r, unsigned off, int soff) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
}
__device__ __forceinline__ float4 buf_f4(__amdgpu_buffer_rsrc_t r, unsigned off, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
}
__device__ __forceinline__ float4 buf_f4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// sum over the 16 lanes of a DPP row (a lane quarter), result in every lane of the row; fixed order
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

constexpr int pad16mod32(int v) { return v + ((16 - (v % 32)) + 32) % 32; }
constexpr int make_odd(int v) { return v | 1; }

// ------------------------------------------------------------------------------------------------------------------------
// weight packing (tiny; once per call)
//   forward : wp[ch][tap][kk][q][co] = w[co][ci = 8 ch + 4 kk + q][tap]       (one LDS row per (tap, k-step, lane quarter))
//   dgrad   : wd[ch][tap][c][ci]     = w[co = 16 ch + c][ci][tap]
//   conv1   : w1[tap][co]            = w[co][0][tap]
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * cin * cout) return;
    const int co = i % cout;
    int r = i / cout;
    const int q = r % 4; r /= 4;
    const int kk = r % 2; r /= 2;
    const int tap = r % 9;
    const int ch = r / 9;
    wp[i] = w[((size_t)co * cin + (8 * ch + 4 * kk + q)) * 9 + tap];
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wd, int cin, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * cin * cout) return;
    const int ci = i % cin;
    int r = i / cin;
    const int c = r % 16; r /= 16;
    const int tap = r % 9;
    const int ch = r / 9;
    wd[i] = w[((size_t)(16 * ch + c) * cin + ci) * 9 + tap];
}

__global__ void pack_conv1_kernel(const float* __restrict__ w, float* __restrict__ w1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 400) return;
    w1[i] = w[(i % 16) * 25 + i / 16];
}

// ------------------------------------------------------------------------------------------------------------------------
// forward of a 3x3 / stride 2 / pad 1 layer.  Workgroup = (image, band of 2 WAVES output rows); wave w owns output rows
// 2w, 2w + 1 of the band, all COUT channels: D[co][pixel] tiles of 16 x 16, K = 8 input channels per LDS chunk x 9 taps.
template <int CIN, int COUT, int HIN, int WIN, int WAVES, bool APPLY>
__global__ __launch_bounds__(WAVES * 64) void conv_s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ y,
                                                                float* __restrict__ stats, int bands) {
    constexpr int HO = (HIN - 1) / 2 + 1, WO = WIN / 2;
    constexpr int NT = WAVES * 64, ROWS = 2 * WAVES, IN_ROWS = 2 * ROWS + 1;
    constexpr int NBT = (WO + 15) / 16, RT = COUT / 16;
    constexpr int EO = 16 * NBT + 1;          // E[0 .. 16 NBT]: E[0] is the left padding, E[j + 1] = column 2j + 1
    constexpr int RS = EO + 16 * NBT;         // O[0 .. 16 NBT - 1]: O[j] = column 2j
    constexpr int PS = pad16mod32(IN_ROWS * RS);
    constexpr int QS = COUT + 16;             // = 16 mod 32 for COUT = 32, 64
    constexpr int W2 = WIN / 2;
    static_assert(WIN % 2 == 0 && COUT % 16 == 0 && CIN % 8 == 0, "shape");
    __shared__ __attribute__((aligned(16))) float s_in[8 * PS + 64];
    __shared__ __attribute__((aligned(16))) float s_w[72 * QS];
    __shared__ float s_ss[2 * CIN];
    __shared__ float s_red[WAVES][COUT][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int oy0 = band * ROWS;
    const float* xin = x + (size_t)n * CIN * HIN * WIN;

    f32x4 acc[2][NBT][RT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[r][bt][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging through registers, slot-major (see the weight-gradient kernel): the loads of chunk ch + 1 are issued before the
    // MFMA loop of chunk ch and land in LDS after it; a thread owns NS (row, column pair) positions and walks the 8 planes.
    constexpr int SLOTS = IN_ROWS * W2, NS = (SLOTS + NT - 1) / NT;
    constexpr int W_UNITS = 72 * COUT / 4, W_IT = (W_UNITS + NT - 1) / NT;
    float2 vin[NS][8];
    float4 vw[W_IT];
    const __amdgpu_buffer_rsrc_t rx = buf_of(xin, CIN * HIN * WIN * 4), rw = buf_of(wp, 9 * CIN * COUT * 4);
    unsigned xoff[NS];
    float* in_dst[NS];
    bool in_act[NS], in_inside[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int sl = tid + k * NT;
        const int row = sl / W2, j = sl - row * W2;
        const int iy = 2 * oy0 - 1 + row;
        in_act[k] = sl < SLOTS;
        in_inside[k] = in_act[k] && iy >= 0 && iy < HIN;
        xoff[k] = in_inside[k] ? (unsigned)((iy * WIN + 2 * j) * 4) : kOob;
        in_dst[k] = s_in + row * RS + j;
        if (in_act[k] && j == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) in_dst[k][p * PS] = 0.f;       // E[0]: the left padding column, never overwritten
        }
    }
    unsigned woff[W_IT];
    float* w_dst[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int u = tid + it * NT;
        const int r = u / (COUT / 4), c4 = u - r * (COUT / 4);
        woff[it] = u < W_UNITS ? (unsigned)(u * 16) : kOob;
        w_dst[it] = s_w + r * QS + 4 * c4;
    }
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int p = 0; p < 8; ++p) vin[k][p] = buf_f2(rx, xoff[k], (ch * 8 + p) * (HIN * WIN * 4));
#pragma unroll
        for (int it = 0; it < W_IT; ++it) vw[it] = buf_f4(rw, woff[it], ch * (72 * COUT * 4));
    };
    auto stash = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (!in_act[k]) continue;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float2 v = vin[k][p];
                if (APPLY) {
                    if (in_inside[k]) {
                        const float sc = s_ss[ch * 8 + p], sh = s_ss[CIN + ch * 8 + p];
                        v.x = fmaxf(v.x, 0.f) * sc + sh;
                        v.y = fmaxf(v.y, 0.f) * sc + sh;
                    }
                }
                in_dst[k][p * PS + EO] = v.x;
                in_dst[k][p * PS + 1] = v.y;
            }
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (tid + it * NT < W_UNITS) *reinterpret_cast<float4*>(w_dst[it]) = vw[it];
    };
    if (APPLY) {
        for (int u = tid; u < CIN; u += NT) { s_ss[u] = scale[u]; s_ss[CIN + u] = shift[u]; }
    }
    fetch(0);
    for (int ch = 0; ch < CIN / 8; ++ch) {
        __syncthreads();
        stash(ch);
        __syncthreads();
        if (ch + 1 < CIN / 8) fetch(ch + 1);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay ahead of the MFMA loop
        if (oy0 + 2 * wave >= HO) continue;      // both rows of this wave lie below the image (last band): staging only
        const float* bb = s_in + q * PS + (4 * wave) * RS + m;
        const float* ab = s_w + q * QS + m;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int xo = (kx == 1 ? EO : 0) + (kx == 2 ? 1 : 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = ab[((tap * 2 + kk) * 4) * QS + 16 * rt];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int bt = 0; bt < NBT; ++bt) {
                        const float b = bb[(4 * kk) * PS + (2 * r + ky) * RS + xo + 16 * bt];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[r][bt][rt] = AG_MFMA4(a[rt], b, acc[r][bt][rt]);
                    }
            }
        }
    }
    // epilogue: + bias, store; with `stats`, the workgroup's sums of relu(y) and relu(y)^2 per output channel over its valid pixels
    // (what the following ReLU + BatchNorm needs as batch statistics, and - last layer - the global average pool as plane sums)
    float ssum[RT][4], ssq[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ssum[rt][i] = ssq[rt][i] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + 2 * wave + r;
        if (oy >= HO) continue;
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
            const int ox = 16 * bt + m;
            if (ox >= WO) continue;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = 16 * rt + 4 * q + i;
                    const float v = acc[r][bt][rt][i] + bias[co];
                    y[(((size_t)n * COUT + co) * HO + oy) * WO + ox] = v;
                    const float rl = fmaxf(v, 0.f);
                    ssum[rt][i] += rl;
                    ssq[rt][i] = fmaf(rl, rl, ssq[rt][i]);
                }
        }
    }
    if (stats) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = row_sum16(ssum[rt][i]), b = row_sum16(ssq[rt][i]);
                if (m == 0) {
                    s_red[wave][16 * rt + 4 * q + i][0] = a;
                    s_red[wave][16 * rt + 4 * q + i][1] = b;
                }
            }
        __syncthreads();
        for (int u = tid; u < COUT * 2; u += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += (&s_red[w][0][0])[u];
            stats[(size_t)blockIdx.x * COUT * 2 + u] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// forward of a 3x3 / stride 2 / pad 1 layer on the BF16 matrix cores at float32 accuracy (round 6): every f32 operand as an exact
// three-way bf16 split, six v_mfma_f32_32x32x16_bf16 per product block (split_common.hpp; split_gemm.hip has the error analysis:
// < 2^-23 |ab| dropped per product, i.e. what an f32 FMA chain rounds away anyway).  Six bf16 MFMAs cost 6/16 of the f32-input
// MFMA the kernel above runs on: the matrix work of a layer falls from ~0.5 ms to ~0.19 ms per 4 750 images and the kernel can
// become a streaming kernel (2.9 GB / 1.5 GB of activations per pass) - IF the vector work of the split (5.5 instructions per staged
// value) runs BESIDE the matrix work.  Hence producer / consumer waves:
//
//   D[co 32][pixel 32] += W[co][k 16] X[k 16][pixel]   per tap and 16 input channels
//   A = weights: pre-split once per call into the exact fragment image (pack_fwd_split_kernel); a consumer wave holds the fragments
//       of its 32 output channels in REGISTERS for the life of the persistent workgroup (9 taps x CIN / 16 x 3 planes x 4
//       registers; the [32 -> 64] layer keeps 12 of its 18 (tap, K step) pairs there and reads 6 from an LDS copy);
//   B = activations: a band of 9 input rows (4 output rows) is staged in LDS channel-innermost - one 16-byte unit = 8 consecutive
//       channels of one position, three planes - with the columns de-interleaved by parity as above (E[j + 1] = column 2j + 1, O[j] =
//       column 2j, E[0] = left padding), so the 32 lanes of a fragment read 32 (or 2 x 16) consecutive units: conflict-free
//       ds_read_b128.  ReLU + BatchNorm of the previous layer and the split are applied once per staged element.
//   Workgroup = 4 PRODUCER waves + 4 CONSUMER waves, one per CU, persistent over whole images, TWO band tiles in LDS:
//       while the consumers run the 54 / 108 MFMAs and the epilogue of band i out of one tile, the producers convert band i + 1
//       (loaded into registers two iterations earlier) into the other and issue the loads of band i + 3; one barrier per band.  Every
//       SIMD hosts a consumer and one or two producers, so the split's vector instructions issue under the other wave's MFMAs.
//   Pixel tile = 32 slots: one output row of up to 32 pixels (WO = 30), or two rows of up to 16 (WO = 15).  Consumer wave = (pixel
//   tile of the band, 32-channel output tile).  A producer wave stages ONE 8-channel group (scale / shift in scalar registers).
// Epilogue as above: + bias, store, per-(image, band) sums of relu(y), relu(y)^2 per output channel (transposing lane reduction,
// fixed order; written by the producers one band later).
template <int CIN, int COUT, int WIN>
struct SplitFwdShape {
    static constexpr int WO = WIN / 2, PXR = (WO <= 16) ? 2 : 1, TW = 32 / PXR, ROWS = 4, IN_ROWS = 2 * ROWS + 1, PT = ROWS / PXR;
    static constexpr int KG = CIN / 8, KS = CIN / 16, CT = COUT / 32;
    // 4 producer + 4 consumer waves.  (Measured for [16 -> 32]: 8 producers at three waves per SIMD with 7 of the 9 taps' fragments in
    // registers - 940 against 986 us in isolation, no difference inside the Planning update: 970 - 981 against 966 - 972 ms per epoch.)
    static constexpr int NCONS = PT * CT, NPROD = 4, NT = (NCONS + NPROD) * 64;
    static constexpr int RS = 2 * TW + 1, PLANE = KG * IN_ROWS * RS, TILE = 3 * PLANE;
    static constexpr int AREG = (KS == 1) ? 9 : 12;                    // (tap, K step) pairs of weight fragments held in registers
};

// wimg[ct][tap][ks][plane][lane] (16-byte units): lane l supplies row co = 32 ct + (l & 31), k = 16 ks + 8 (l >> 5) .. + 7 of tap
__global__ void pack_fwd_split_kernel(const float* __restrict__ w, uint4* __restrict__ wimg, int cin, int cout) {
    const int ks_n = cin / 16, ct_n = cout / 32;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ct_n * 9 * ks_n * 64) return;
    const int lane = t & 63;
    int r = t >> 6;
    const int ks = r % ks_n; r /= ks_n;
    const int tap = r % 9;
    const int ct = r / 9;
    const int co = 32 * ct + (lane & 31), c0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = w[((size_t)co * cin + (c0 + i)) * 9 + tap];
    uint4 p1, p2, p3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
    uint4* dst = wimg + ((size_t)((ct * 9 + tap) * ks_n + ks) * 3) * 64 + lane;
    dst[0] = p1;
    dst[64] = p2;
    dst[128] = p3;
}

// lane exchange with the lane whose index differs in bit B (inside each 32-lane half): DPP where a pattern exists, ds_swizzle
// (bit-mask mode, no LDS memory touched) otherwise
template <int B>
__device__ __forceinline__ float xchg_bit(float v) {
    const int x = __builtin_bit_cast(int, v);
    if constexpr (B == 0) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
    else if constexpr (B == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
    else if constexpr (B == 3) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false)); // row_ror:8
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(x, ((1 << B) << 10) | 0x1F));                          // xor (1 << B)
}
// One step of the transposing reduction: V[2m], V[2m + 1] of this lane and of its partner in bit B become ONE value per lane - the
// partner-pair sum of V[2m] in lanes with bit B clear, of V[2m + 1] in lanes with it set.  After the five steps over 32 values, lane l
// of a 32-lane half holds the half's total of V[l & 31] (fixed order: deterministic).
template <int B, int LEN>
__device__ __forceinline__ void treduce_step(float* v, bool bit) {
#pragma unroll
    for (int m = 0; m < LEN / 2; ++m) {
        const float keep = bit ? v[2 * m + 1] : v[2 * m];
        const float send = bit ? v[2 * m] : v[2 * m + 1];
        v[m] = keep + xchg_bit<B>(send);
    }
}

// Workgroup barrier that waits for this wave's LDS traffic only - never for the global loads a producer has in flight for the bands
// ahead.  (This is what __syncthreads() compiles to with this toolchain in the default workgroup mode - s_waitcnt lgkmcnt(0); s_barrier -
// spelled out here because the kernel's load pipeline depends on it.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

template <int CIN, int COUT, int HIN, int WIN, bool APPLY>
__global__ __launch_bounds__((SplitFwdShape<CIN, COUT, WIN>::NT)) void conv_s2_fwd_split_kernel(
    const float* __restrict__ x, const uint4* __restrict__ wimg, const float* __restrict__ bias, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ y, float* __restrict__ stats, int n_images, int image_major) {
    using S = SplitFwdShape<CIN, COUT, WIN>;
    constexpr int HO = (HIN - 1) / 2 + 1, WO = S::WO, W2 = WIN / 2;
    constexpr int PXR = S::PXR, TW = S::TW, ROWS = S::ROWS, IN_ROWS = S::IN_ROWS, PT = S::PT;
    constexpr int KG = S::KG, KS = S::KS, CT = S::CT, RS = S::RS, PLANE = S::PLANE, TILE = S::TILE;
    constexpr int NCONS = S::NCONS, NPROD = S::NPROD, AREG = S::AREG, ALDS = 9 * KS - AREG;
    constexpr int bands = (HO + ROWS - 1) / ROWS;          // (a constant: the item -> (image, band) divisions are multiplications)
    static_assert(WIN % 2 == 0 && CIN % 16 == 0 && COUT % 32 == 0 && WO <= 32 && PT * PXR == ROWS && NPROD % KG == 0, "shape");
    __shared__ uint4 s_in[2 * TILE];
    __shared__ float s_red[2][PT][COUT][2];
    __shared__ uint4 s_w[ALDS > 0 ? CT * ALDS * 3 * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < NCONS;
    // Item order.  image_major (large batches): a workgroup walks WHOLE images, band after band (image blockIdx.x, blockIdx.x +
    // gridDim.x, ...): its 16 / 32 input planes and 32 / 64 output planes are then sequential streams in memory, the row a band shares
    // with the next one is still in cache, and the 480-byte pieces a band writes into every output plane are continued by the same
    // workgroup a band later.  Otherwise (fewer than ~8 images per workgroup: the tail of whole images would idle CUs) items are dealt
    // round robin: item blockIdx.x, blockIdx.x + gridDim.x, ...
    constexpr int B_ = (HO + ROWS - 1) / ROWS;
    const int items = n_images * B_;
    const int my_images = (int)blockIdx.x < n_images ? (n_images - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nv = image_major ? my_images * B_ : ((int)blockIdx.x < items ? (items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0);
    auto item_of = [&](int v) {
        if (!image_major) return (int)blockIdx.x + v * (int)gridDim.x;
        const int q = v / B_;
        return ((int)blockIdx.x + q * (int)gridDim.x) * B_ + (v - q * B_);
    };
    // ---- set-up common to both roles: the LDS copy of the weight fragments that do not live in registers, the padding columns
    if (ALDS > 0) {
        for (int u = tid; u < CT * ALDS * 3 * 64; u += S::NT) {
            const int l = u & 63, p = (u >> 6) % 3, r = (u >> 6) / 3;
            const int pair = AREG + r % ALDS, ct_ = r / ALDS;                  // pair = tap * KS + ks
            s_w[u] = wimg[((size_t)(ct_ * 9 * KS + pair) * 3 + p) * 64 + l];
        }
    }
    for (int u = tid; u < 2 * 3 * KG * IN_ROWS; u += S::NT) s_in[(size_t)u * RS] = make_uint4(0u, 0u, 0u, 0u);   // E[0] of every row, both tiles
    __syncthreads();

    if (!consumer) {
        // =========================== producer: global -> registers -> (ReLU + BatchNorm, split) -> LDS tile =======================
        constexpr int WPG = NPROD / KG;                              // producer waves per 8-channel group
        constexpr int UNITS = IN_ROWS * W2, NU = (UNITS + WPG * 64 - 1) / (WPG * 64);
        const int pw = wave - NCONS, kg = pw / WPG, gt = (pw - kg * WPG) * 64 + lane;
        float sc8[8], sh8[8];
        if (APPLY) {
#pragma unroll
            for (int p = 0; p < 8; ++p) { sc8[p] = scale[kg * 8 + p]; sh8[p] = shift[kg * 8 + p]; }
        }
        int urow[NU];
        unsigned uoff[NU];
        int udst[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = gt + k * WPG * 64;
            const int row = u / W2, j = u - row * W2;
            urow[k] = u < UNITS ? row : -1000;
            uoff[k] = (unsigned)((row * WIN + 2 * j) * 4);
            udst[k] = (kg * IN_ROWS + row) * RS + j;
        }
        // two register sets: the loads of band i + 2 are issued at the START of the iteration in which band i + 1 is converted (from the
        // other set), so a band's loads have more than a whole band time to arrive - one band of look-ahead was not enough, a band
        // (~2 us) being about one loaded memory latency
        float2 vinA[NU][8], vinB[NU][8];
        auto fetch = [&](float2 (&vin)[NU][8], int item_) {
            const int n_ = item_ / bands, top = 2 * (item_ - n_ * bands) * ROWS - 1;      // input row of staged row 0
            const __amdgpu_buffer_rsrc_t rx = buf_of(x + ((size_t)n_ * CIN + kg * 8) * HIN * WIN, 8 * HIN * WIN * 4);
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                const int iy = top + urow[k];
                const unsigned off = (iy >= 0 && iy < HIN) ? uoff[k] + (unsigned)(top * WIN * 4) : kOob;
#pragma unroll
                for (int p = 0; p < 8; ++p) vin[k][p] = buf_f2(rx, off, p * (HIN * WIN * 4));
            }
        };
        auto stash = [&](float2 (&vin)[NU][8], int item_, int par) {
            const int n_ = item_ / bands, top = 2 * (item_ - n_ * bands) * ROWS - 1;
            uint4* tile = s_in + (size_t)par * TILE;
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                if (urow[k] < 0) continue;
                const int iy = top + urow[k];
                float2 v[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) v[p] = vin[k][p];
                if (APPLY) {
                    if (iy >= 0 && iy < HIN) {                    // (padding rows stay zero)
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            v[p].x = fmaxf(v[p].x, 0.f) * sc8[p] + sh8[p];
                            v[p].y = fmaxf(v[p].y, 0.f) * sc8[p] + sh8[p];
                        }
                    }
                }
                uint4 o1, o2, o3, e1, e2, e3;
#ifdef AG_CONVF_SKELETON      /* timing experiment: memory traffic + barriers only (results wrong by construction) */
                o1 = make_uint4(__float_as_uint(v[0].x), __float_as_uint(v[1].x), __float_as_uint(v[2].x), __float_as_uint(v[3].x));
                o2 = make_uint4(__float_as_uint(v[4].x), __float_as_uint(v[5].x), __float_as_uint(v[6].x), __float_as_uint(v[7].x));
                e1 = make_uint4(__float_as_uint(v[0].y), __float_as_uint(v[1].y), __float_as_uint(v[2].y), __float_as_uint(v[3].y));
                e2 = make_uint4(__float_as_uint(v[4].y), __float_as_uint(v[5].y), __float_as_uint(v[6].y), __float_as_uint(v[7].y));
                o3 = o1; e3 = e1;
#else
                split8(make_float4(v[0].x, v[1].x, v[2].x, v[3].x), make_float4(v[4].x, v[5].x, v[6].x, v[7].x), o1, o2, o3);
                split8(make_float4(v[0].y, v[1].y, v[2].y, v[3].y), make_float4(v[4].y, v[5].y, v[6].y, v[7].y), e1, e2, e3);
#endif
                uint4* base = tile + udst[k];
                base[TW + 1] = o1;             // O[j]     = column 2j
                base[PLANE + TW + 1] = o2;
                base[2 * PLANE + TW + 1] = o3;
                base[1] = e1;                  // E[j + 1] = column 2j + 1
                base[PLANE + 1] = e2;
                base[2 * PLANE + 1] = e3;
            }
        };
        const int ptid = tid - NCONS * 64;
        auto flush_stats = [&](int item_, int par) {
            for (int u = ptid; u < COUT * 2; u += NPROD * 64) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < PT; ++w) t += (&s_red[par][w][0][0])[u];
                stats[(size_t)item_ * COUT * 2 + u] = t;
            }
        };
        if (nv > 0) {
            fetch(vinA, item_of(0));
            if (nv > 1) fetch(vinB, item_of(1));
            stash(vinA, item_of(0), 0);
        }
        lds_barrier();                                             // tile 0 published
        // iteration v: the consumers work on tile v & 1 (item v of this workgroup); this wave issues the loads of item v + 2, converts
        // item v + 1 into tile (v + 1) & 1 and writes out the statistics of item v - 1
        for (int v = 0; v < nv; v += 2) {
            if (v + 2 < nv) fetch(vinA, item_of(v + 2));
            if (v + 1 < nv) stash(vinB, item_of(v + 1), 1);
            if (stats && v > 0) flush_stats(item_of(v - 1), 1);
            lds_barrier();
            if (v + 1 >= nv) break;
            if (v + 3 < nv) fetch(vinB, item_of(v + 3));
            if (v + 2 < nv) stash(vinA, item_of(v + 2), 0);
            if (stats) flush_stats(item_of(v), 0);
            lds_barrier();
        }
        if (stats && nv > 0) flush_stats(item_of(nv - 1), (nv - 1) & 1);
        return;
    }

    // =============================== consumer: 9 taps x KS x 6 MFMAs per band, epilogue ============================================
    const int ct = wave % CT, pt = wave / CT;
    const int slot = lane & 31, pr = slot / TW, ox = slot - pr * TW, kh = lane >> 5;
    bf16x8 a[AREG][3];
#pragma unroll
    for (int pair = 0; pair < AREG; ++pair)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 u = wimg[((size_t)(ct * 9 * KS + pair) * 3 + p) * 64 + lane];
            a[pair][p] = __builtin_bit_cast(bf16x8, u);
        }
    const uint4* wl = s_w + (size_t)(ct * ALDS * 3) * 64 + lane;
    float bs[16];                              // the lane's 16 output channels' bias
#pragma unroll
    for (int i = 0; i < 16; ++i) bs[i] = bias[32 * ct + 8 * (i >> 2) + 4 * kh + (i & 3)];
    const int frag0 = (kh * IN_ROWS + 2 * (pt * PXR + pr)) * RS + ox;       // K step ks adds 2 ks channel groups

    lds_barrier();                                               // tile 0 published
    for (int k = 0; k < nv; ++k) {
        const int item = item_of(k);
        const int n = item / bands, band = item - n * bands;
        const int oy0 = band * ROWS;
        const uint4* tile = s_in + (size_t)(k & 1) * TILE;
        f32x16 acc0, acc1;                     // two chains: consecutive product blocks do not wait for each other
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        const bool tile_live = oy0 + pt * PXR < HO;         // (the last band: tiles below the image only take part in the barriers)
#ifdef AG_CONVF_SKELETON
        if (tile_live) { const uint4 q0 = tile[frag0]; acc0[0] = __uint_as_float(q0.x); acc1[1] = __uint_as_float(q0.y); }
        if (false) {
#else
        if (tile_live) {
#endif
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int co_ = (kx == 1 ? TW + 1 : 0) + (kx == 2 ? 1 : 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int pair = tap * KS + ks;
                    const uint4* q = tile + frag0 + (2 * ks * IN_ROWS + ky) * RS + co_;
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, q[0]);
                    const bf16x8 b2 = __builtin_bit_cast(bf16x8, q[PLANE]);
                    const bf16x8 b3 = __builtin_bit_cast(bf16x8, q[2 * PLANE]);
                    bf16x8 a1, a2, a3;
                    if (pair < AREG) {
                        a1 = a[pair < AREG ? pair : 0][0]; a2 = a[pair < AREG ? pair : 0][1]; a3 = a[pair < AREG ? pair : 0][2];
                    } else {
                        const uint4* wq = wl + (size_t)((pair - AREG) * 3) * 64;
                        a1 = __builtin_bit_cast(bf16x8, wq[0]);
                        a2 = __builtin_bit_cast(bf16x8, wq[64]);
                        a3 = __builtin_bit_cast(bf16x8, wq[128]);
                    }
                    if (pair & 1) { AG_MFMA_SPLIT(acc1, a1, a2, a3, b1, b2, b3); }
                    else { AG_MFMA_SPLIT(acc0, a1, a2, a3, b1, b2, b3); }
                }
            }
        }
        // ---- epilogue: + bias, store, sums of relu(y) and relu(y)^2 over the tile's valid pixels
        {
            const int oy = oy0 + pt * PXR + pr;
            const bool valid = tile_live && oy < HO && ox < WO;
            // one buffer store per accumulator register: the lane part of the address (pixel, and the 4-channel step of the lane's half)
            // in the vector offset - out of range for lanes without a pixel, which the hardware then drops -, the register's channel in
            // the scalar offset
            const __amdgpu_buffer_rsrc_t ry = buf_of(y + ((size_t)n * COUT + 32 * ct) * HO * WO, 32 * HO * WO * 4);
            const unsigned voff = valid ? (unsigned)(((4 * kh) * HO * WO + oy * WO + ox) * 4) : kOob;
            float red[32];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = (acc0[i] + acc1[i]) + bs[i];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff, (8 * (i >> 2) + (i & 3)) * (HO * WO * 4), 0);
                const float rl = valid ? fmaxf(v, 0.f) : 0.f;
                red[2 * i] = rl;
                red[2 * i + 1] = rl * rl;
            }
            if (stats) {
                treduce_step<0, 32>(red, (lane & 1) != 0);
                treduce_step<1, 16>(red, (lane & 2) != 0);
                treduce_step<2, 8>(red, (lane & 4) != 0);
                treduce_step<3, 4>(red, (lane & 8) != 0);
                treduce_step<4, 2>(red, (lane & 16) != 0);
                // lane l of a half now holds the half's total of value (l & 31): register i = l >> 1 (bits 1..4), statistic l & 1
                const int i_ = slot >> 1;
                s_red[k & 1][pt][32 * ct + 8 * (i_ >> 2) + 4 * kh + (i_ & 3)][slot & 1] = red[0];
            }
        }
        lds_barrier();                       // tile (k + 1) & 1 is published, this band's statistics are complete in s_red[k & 1]
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// input gradient of a 3x3 / stride 2 / pad 1 layer: din[ci][iy][ix] = sum_{co,ky,kx} dz[co][oy][ox] w[co][ci][ky][kx] with
// iy = 2 oy + ky - 1, ix = 2 ox + kx - 1.  Output pixels split into four parity classes (iy = 2a + py, ix = 2b + px):
//   py = 0: ky = 1, oy = a;   py = 1: ky = 0, oy = a + 1  and  ky = 2, oy = a        (the same in x)
// so every tap reads dz at (a + dy, b + dx), dy, dx in {0, 1}: UNIT-stride reads, no de-interleaving; a lane holds the px = 0
// and px = 1 results of its b and stores them as one float2 (rows of din are written contiguously).
// D[ci][b] tiles of 16 x 16, K = 16 output channels per LDS chunk x the taps of the class.  Wave w owns a-rows 2w, 2w + 1.
// EPI (the layer's input is the output of a ReLU + BatchNorm whose backward coefficients are known before this kernel runs - see
// bn_sums_from_conv in cnn_kernels.hip): 1 = the epilogue turns the gradient of the BatchNorm output into the gradient of the
// convolution output x in front of it, din <- [x > 0] (A din + m_i (B x + C)), tab[ci] = {A, B, C, 0}, m_i = wts[n] (the arithmetic of
// conv1_wgrad_kernel's BNBWD): ag_relu_bn_bwd_dx's pass over the tensor (read 2, write 1) becomes one extra read here.
// 2 = also emits, per workgroup, sums[ci][6] = {total, row 0, last row, column 0, (0,0), (last row, 0)} of what it stored: the bias
// gradient and the border sums the layer below needs for ITS reductions.
// 3 (second layer only) = the input gradient is not stored at all: the epilogue applies the FIRST layer's ReLU + BatchNorm backward
// (as 1) and feeds the result, through LDS, straight into the first convolution's weight gradient (conv1_wgrad_kernel's MFMA loop on
// the band's 16 rows, the image band staged with the normaliser as there): `c1_part` [grid][16][32] = per workgroup dw1 [16][25], db1
// in column 25.  The 1.9 GB gradient of the first layer's output is neither written nor read.
struct Conv1Side {             // EPI == 3: the first convolution's input side
    const float* img;          // frames [*, 212, 120]
    const long long* index;    // image i of the batch = frame index[i] (NULL: i)
    const float* nmean;        // per-pixel normaliser statistics (NULL: the image is used as it is)
    const float* nstd;
    float* c1_part;
};
template <int CIN, int COUT, int HIN, int WIN, int WAVES, int EPI>
__global__ __launch_bounds__(WAVES * 64) void conv_s2_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wd,
                                                                  float* __restrict__ dx, int bands, const float* __restrict__ bnx,
                                                                  const float* __restrict__ tab, const float* __restrict__ wts,
                                                                  float* __restrict__ sums, const Conv1Side c1) {
    constexpr int HO = (HIN - 1) / 2 + 1, WO = WIN / 2;
    constexpr int NT = WAVES * 64, AROWS = 2 * WAVES, ZR = AROWS + 1;
    constexpr int NBT = (WO + 15) / 16, RT = CIN / 16;
    constexpr int RSZ = 16 * NBT + 1;
    constexpr int PSZ = pad16mod32(ZR * RSZ);
    constexpr int CINP = (CIN == 16) ? 16 : CIN + 16;
    static_assert(CIN % 16 == 0 && COUT % 16 == 0, "shape");
    static_assert(EPI != 3 || (CIN == 16 && HIN == 106 && WIN == 60 && WAVES == 4), "EPI 3: the second layer");
    // EPI 3 reuses the staging buffers after the main loop: image band [35][136] + 8, gradient tile [16][482], table [64]
    constexpr int C1_RS = 136, C1_EO = 68, C1_ROWS = 2 * (2 * AROWS) + 3, C1_PSZ = 482;
    constexpr int MAIN_FLOATS = 16 * PSZ + 32 + 9 * 16 * CINP, C1_FLOATS = C1_ROWS * C1_RS + 8 + 16 * C1_PSZ + 64;
    __shared__ __attribute__((aligned(16))) float s_all[(EPI == 3 && C1_FLOATS > MAIN_FLOATS) ? C1_FLOATS : MAIN_FLOATS];
    float* const s_z = s_all;
    float* const s_w = s_all + 16 * PSZ + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int a0 = band * AROWS;
    const float* zin = dz + (size_t)n * COUT * HO * WO;

    f32x4 acc[2][2][2][RT][NBT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int bt = 0; bt < NBT; ++bt) acc[r][py][px][rt][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging, slot-major: a thread owns one (row, column [pair]) position of the dz band and walks the 16 planes of a chunk.
    // Only positions inside the tensor are staged; the zero column / rows behind them are written once, here.
    constexpr int ZV = (WO % 2 == 0) ? 2 : 1;
    constexpr int SLOTS = ZR * (WO / ZV);
    static_assert(SLOTS <= NT, "one dz slot per thread");
    constexpr int W_UNITS = 9 * 16 * CIN / 4, W_IT = (W_UNITS + NT - 1) / NT;
    for (int u = tid; u < 16 * PSZ + 32; u += NT) s_z[u] = 0.f;
    const int zlr = tid / (WO / ZV), zc = (tid - zlr * (WO / ZV)) * ZV;
    const bool z_act = tid < SLOTS;
    const unsigned zoff = (z_act && a0 + zlr < HO) ? (unsigned)(((a0 + zlr) * WO + zc) * 4) : kOob;
    float* const z_dst = s_z + zlr * RSZ + zc;
    const __amdgpu_buffer_rsrc_t rz = buf_of(zin, COUT * HO * WO * 4), rw = buf_of(wd, 9 * CIN * COUT * 4);
    float vz[16][ZV];
    float4 vw[W_IT];
    unsigned woff[W_IT];
    float* w_dst[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int u = tid + it * NT;
        const int r = u / (CIN / 4), c4 = u - r * (CIN / 4);
        woff[it] = u < W_UNITS ? (unsigned)(u * 16) : kOob;
        w_dst[it] = s_w + r * CINP + 4 * c4;
    }
    auto fetch = [&](int ch) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (ZV == 2) {
                const float2 v = buf_f2(rz, zoff, (ch * 16 + p) * (HO * WO * 4));
                vz[p][0] = v.x;
                vz[p][ZV - 1] = v.y;
            } else {
                vz[p][0] = buf_f1(rz, zoff, (ch * 16 + p) * (HO * WO * 4));
            }
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) vw[it] = buf_f4(rw, woff[it], ch * (9 * 16 * CIN * 4));
    };
    auto stash = [&]() {
        if (z_act) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                z_dst[p * PSZ] = vz[p][0];
                if (ZV == 2) z_dst[p * PSZ + 1] = vz[p][ZV - 1];
            }
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (tid + it * NT < W_UNITS) *reinterpret_cast<float4*>(w_dst[it]) = vw[it];
    };
    // EPI: the values of the layer's output the epilogue needs (one float2 per float2 it stores), loaded behind the LAST chunk's
    // staging so that they land under its MFMA loop - all waves of the workgroup reach the epilogue together and nothing else would
    // hide the latency there (one workgroup per CU)
    constexpr bool PRE = EPI == 1 || EPI == 2;
    float2 xpre[PRE ? 2 : 1][PRE ? 2 : 1][PRE ? NBT : 1][PRE ? RT : 1][PRE ? 4 : 1];
    auto prefetch_x = [&]() {
        const __amdgpu_buffer_rsrc_t rx = buf_of(bnx + (size_t)n * CIN * HIN * WIN, CIN * HIN * WIN * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int iy = 2 * (a0 + 2 * wave + r) + py;
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    const int b = 16 * bt + m;
                    const unsigned off = (iy < HIN && b < WO) ? (unsigned)(((4 * q * HIN + iy) * WIN + 2 * b) * 4) : kOob;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) xpre[r][py][bt][rt][i] = buf_f2(rx, off, (16 * rt + i) * (HIN * WIN * 4));
                }
            }
    };
    // EPI 3: the image band of the first convolution's weight gradient (and the normaliser's statistics at the same pixels), loaded
    // in the same place
    constexpr int IM_IT = EPI == 3 ? (C1_ROWS * 60 + NT - 1) / NT : 1;
    float2 im[IM_IT], imu[IM_IT], isd[IM_IT];
    auto prefetch_img = [&]() {
        const float* xim = c1.img + (size_t)(c1.index ? c1.index[n] : (long long)n) * (212 * 120);
        const __amdgpu_buffer_rsrc_t ri = buf_of(xim, 212 * 120 * 4);
        const __amdgpu_buffer_rsrc_t rm = buf_of(c1.nmean ? c1.nmean : xim, 212 * 120 * 4);
        const __amdgpu_buffer_rsrc_t rs = buf_of(c1.nstd ? c1.nstd : xim, 212 * 120 * 4);
#pragma unroll
        for (int it = 0; it < IM_IT; ++it) {
            const int u = tid + it * NT;
            const int row = u / 60, jj = u - row * 60;
            const int iy = 4 * a0 - 2 + row;
            const unsigned off = (u < C1_ROWS * 60 && iy >= 0 && iy < 212) ? (unsigned)((iy * 120 + 2 * jj) * 4) : kOob;
            im[it] = buf_f2(ri, off);
            if (c1.nmean) {
                imu[it] = buf_f2(rm, off);
                isd[it] = buf_f2(rs, off);
            }
        }
    };
    fetch(0);
    for (int ch = 0; ch < COUT / 16; ++ch) {
        __syncthreads();
        stash();
        __syncthreads();
        if (ch + 1 < COUT / 16) fetch(ch + 1);
        else if (PRE) prefetch_x();
        else if (EPI == 3) prefetch_img();
        __builtin_amdgcn_sched_barrier(0);      // the loads stay ahead of the MFMA loop
        if (a0 + 2 * wave >= HO) continue;       // both rows of this wave lie below the tensor (last band): staging only
        const float* zb = s_z + q * PSZ + (2 * wave) * RSZ + m;
        const float* ab = s_w + q * CINP + m;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float B[3][2][NBT];
#pragma unroll
            for (int ro = 0; ro < 3; ++ro)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int bt = 0; bt < NBT; ++bt) B[ro][d][bt] = zb[(4 * s) * PSZ + ro * RSZ + 16 * bt + d];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int py = (ky != 1), dy = (ky == 0), px = (kx != 1), dxx = (kx == 0);
                float a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = ab[(tap * 16 + 4 * s) * CINP + 16 * rt];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int bt = 0; bt < NBT; ++bt)
                            acc[r][py][px][rt][bt] = AG_MFMA4(a[rt], B[r + dy][dxx][bt], acc[r][py][px][rt][bt]);
            }
        }
    }
    if constexpr (EPI == 3) {
        float* const s_in = s_all;
        float* const s_dz = s_all + C1_ROWS * C1_RS + 8;
        float* const s_tab = s_dz + 16 * C1_PSZ;
        // the layer's own output at this lane's positions, one half (r) at a time: loaded ahead of use
        const __amdgpu_buffer_rsrc_t rx1 = buf_of(bnx + (size_t)n * CIN * HIN * WIN, CIN * HIN * WIN * 4);
        float2 xq[2][2][NBT][4];
        auto load_x = [&](int r) {
            const int a = a0 + 2 * wave + r;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    const int b = 16 * bt + m;
                    const unsigned off = (a < HO && b < WO) ? (unsigned)(((4 * q * HIN + 2 * a + py) * WIN + 2 * b) * 4) : kOob;
#pragma unroll
                    for (int i = 0; i < 4; ++i) xq[r][py][bt][i] = buf_f2(rx1, off, i * (HIN * WIN * 4));
                }
        };
        load_x(0);
        __syncthreads();                    // every wave is done with the main loop's buffers
        if (tid < 64) s_tab[tid] = tab[tid];
        // the image band of the 16 gradient rows (rows 4 a0 - 2 .. + 34), columns de-interleaved by parity as in conv1_wgrad_kernel:
        // E[k] = column 2k - 2 at [k], O[k] = column 2k - 1 at [EO + k]; the pad positions (columns -2, -1, 120 ...) are zero
        for (int u = tid; u < C1_ROWS * 16; u += NT) {
            const int row = u >> 4, k = u & 15;
            s_in[row * C1_RS + (k == 0 ? 0 : (k < 9 ? 60 + k : 120 + k))] = 0.f;
        }
#pragma unroll
        for (int it = 0; it < IM_IT; ++it) {
            const int u = tid + it * NT;
            if (u >= C1_ROWS * 60) continue;
            const int row = u / 60, jj = u - row * 60;
            const int iy = 4 * a0 - 2 + row;
            float2 v = im[it];
            if (c1.nmean && iy >= 0 && iy < 212) {
                v.x = fminf(fmaxf((v.x - imu[it].x) * __builtin_amdgcn_rcpf(isd[it].x), -5.f), 5.f);
                v.y = fminf(fmaxf((v.y - imu[it].y) * __builtin_amdgcn_rcpf(isd[it].y), -5.f), 5.f);
            }
            s_in[row * C1_RS + jj + 1] = v.x;
            s_in[row * C1_RS + C1_EO + jj + 1] = v.y;
        }
        const float wi = wts ? wts[n] : 1.0f;
        int boff[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tap = m + 16 * t, kyy = tap / 5, kxx = tap - 5 * kyy;
            boff[t] = (tap < 25) ? kyy * C1_RS + (kxx & 1) * C1_EO + (kxx >> 1) + q : 0;
        }
        const bool t1_data = (m + 16) < 25;
        const float t1_const = (m + 16 == 25) ? 1.f : 0.f;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
        float tA[4], tB[4], tC[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t4 = *reinterpret_cast<const float4*>(tab + 4 * (4 * q + i));
            tA[i] = t4.x;
            tB[i] = wi * t4.y;
            tC[i] = wi * t4.z;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int a = a0 + 2 * wave + r;
            const bool live = a < HO;           // wave-uniform: a-rows past the tensor (last band) contribute nothing
            if (live) {
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int bt = 0; bt < NBT; ++bt) {
                        const int b = 16 * bt + m;
                        if (b >= WO) continue;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float2 xv = xq[r][py][bt][i];
                            float2 v = make_float2(acc[r][py][0][0][bt][i], acc[r][py][1][0][bt][i]);
                            v.x = xv.x > 0.f ? fmaf(v.x, tA[i], fmaf(xv.x, tB[i], tC[i])) : 0.f;
                            v.y = xv.y > 0.f ? fmaf(v.y, tA[i], fmaf(xv.y, tB[i], tC[i])) : 0.f;
                            *reinterpret_cast<float2*>(s_dz + (4 * q + i) * C1_PSZ + (2 * wave + py) * 60 + 2 * b) = v;
                        }
                    }
            }
            if (r == 0) load_x(1);
            __syncthreads();        // the gradient rows of this half (and, the first time, the image band) are in LDS
            if (live) {
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int lr = 2 * (2 * wave + r) + py;         // gradient row within the band's 16
                    const float* ap = s_dz + m * C1_PSZ + (2 * wave + py) * 60 + q;
                    const float* bp = s_in + (2 * lr) * C1_RS;
#pragma unroll 5
                    for (int j = 0; j < 15; ++j) {
                        const float av = ap[4 * j];
                        const float b0 = bp[boff[0] + 4 * j];
                        float b1 = bp[boff[1] + 4 * j];
                        b1 = t1_data ? b1 : t1_const;
                        w0 = AG_MFMA4(av, b0, w0);
                        w1 = AG_MFMA4(av, b1, w1);
                    }
                }
            }
            __syncthreads();        // before the next half overwrites the tile / the reduction reuses it
        }
        float* red = s_dz;          // [4 waves][16 co][32 taps]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[(wave * 16 + 4 * q + i) * 32 + m] = w0[i];
            red[(wave * 16 + 4 * q + i) * 32 + 16 + m] = w1[i];
        }
        __syncthreads();
        for (int u = tid; u < 512; u += NT) c1.c1_part[(size_t)blockIdx.x * 512 + u] = red[u] + red[512 + u] + red[1024 + u] + red[1536 + u];
        return;
    }
    float* dout = dx + (size_t)n * CIN * HIN * WIN;
    float tA[RT][4], tB[RT][4], tC[RT][4];
    float tot[RT][4], r0[RT][4], rl[RT][4], c0[RT][4], k00[RT][4], kl0[RT][4];
    if (EPI) {
        const float wi = wts ? wts[n] : 1.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 t4 = *reinterpret_cast<const float4*>(tab + 4 * (16 * rt + 4 * q + i));
                tA[rt][i] = t4.x;
                tB[rt][i] = wi * t4.y;
                tC[rt][i] = wi * t4.z;
                tot[rt][i] = r0[rt][i] = rl[rt][i] = c0[rt][i] = k00[rt][i] = kl0[rt][i] = 0.f;
            }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int a = a0 + 2 * wave + r;
        if (a >= HO) continue;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int iy = 2 * a + py;
            if (iy >= HIN) continue;
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) {
                const int b = 16 * bt + m;
                if (b >= WO) continue;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ci = 16 * rt + 4 * q + i;
                        const size_t at = ((size_t)ci * HIN + iy) * WIN + 2 * b;
                        float2 v = make_float2(acc[r][py][0][rt][bt][i], acc[r][py][1][rt][bt][i]);
                        if (EPI) {
                            const float2 xv = xpre[r][py][bt][rt][i];
                            v.x = xv.x > 0.f ? fmaf(v.x, tA[rt][i], fmaf(xv.x, tB[rt][i], tC[rt][i])) : 0.f;
                            v.y = xv.y > 0.f ? fmaf(v.y, tA[rt][i], fmaf(xv.y, tB[rt][i], tC[rt][i])) : 0.f;
                        }
                        if (EPI == 2) {
                            const float both = v.x + v.y;
                            tot[rt][i] += both;
                            if (iy == 0) r0[rt][i] += both;                 // wave-uniform conditions
                            if (iy == HIN - 1) rl[rt][i] += both;
                            if (b == 0) {
                                c0[rt][i] += v.x;
                                if (iy == 0) k00[rt][i] += v.x;
                                if (iy == HIN - 1) kl0[rt][i] += v.x;
                            }
                        }
                        *reinterpret_cast<float2*>(dout + at) = v;
                    }
            }
        }
    }
    if (EPI == 2) {
        // per-wave sums over the 16 columns of a row group, then over the waves through LDS (s_z is free after this barrier)
        __syncthreads();
        float* red = s_z;       // [WAVES][CIN][6]
        static_assert(WAVES * CIN * 6 <= 16 * PSZ, "reduction scratch fits the staging buffer");
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = row_sum16(tot[rt][i]), u0 = row_sum16(r0[rt][i]), ul = row_sum16(rl[rt][i]);
                if (m == 0) {
                    float* o = red + (wave * CIN + 16 * rt + 4 * q + i) * 6;
                    o[0] = t;
                    o[1] = u0;
                    o[2] = ul;
                    o[3] = c0[rt][i];
                    o[4] = k00[rt][i];
                    o[5] = kl0[rt][i];
                }
            }
        __syncthreads();
        for (int u = tid; u < CIN * 6; u += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w * CIN * 6 + u];
            sums[(size_t)blockIdx.x * (CIN * 6) + u] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// weight gradient of a 3x3 / stride 2 / pad 1 layer: dw[co][ci][ky][kx] = sum_{n,oy,ox} dz[co][oy][ox] in[ci][2oy+ky-1][2ox+kx-1],
// db[co] = sum dz.  D[co][ci] tiles of 16 x 16 per tap, K = output pixels (4 per instruction, one per lane quarter).  Twelve waves:
// wave w owns kernel row ky = w >> 2 (its three kx taps), half of the co tiles and one ci tile (conv3) or half of the pixel
// steps (conv2); a band is 4 output rows.  Workgroups are persistent: each accumulates its (image, band) items in registers and writes ONE
// partial [COUT*CIN*9 + COUT]; the caller sums the partials (fixed order -> deterministic).
// Pixel <-> quarter map (chosen for the bank rule): WO > 16 (30 wide): quarter = (row & 1, half) - rows 2r'+(q>>1), ox = j + 16 (q&1);
// WO <= 16 (15 wide): quarter = row, ox = j.  Pixels ox >= WO have dz = 0 in LDS (and finite input values).
// BIAS = false: db is not computed (the fused trunk takes it from the per-plane sums of dz that the ReLU + BatchNorm backward kernel
// producing dz emits for free; the constant-1 MFMA column costs the kernel-row-0 waves a third more matrix work: 1.30 -> 1.09 ms).
template <int CIN, int COUT, int HIN, int WIN, bool APPLY, bool BIAS>
__global__ __launch_bounds__(768) void conv_s2_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ partials, int items, int bands) {
    constexpr int HO = (HIN - 1) / 2 + 1, WO = WIN / 2;
    constexpr int NT = 768, R = 4, IN_ROWS = 2 * R + 1;
    constexpr bool WIDE = (WO > 16);
    constexpr int RT = COUT / 16, CT = CIN / 16, RTW = RT / 2;
    static_assert(CT == 1 || CT == 2, "the fourth wave-id bit is the ci tile (CT = 2) or the K half (CT = 1)");
    constexpr int RSZ = WIDE ? 32 : 16, PSZ = R * RSZ + 1;
    constexpr int EO = WIDE ? 33 : 17, RS = WIDE ? 65 : 40;
    constexpr int PS = make_odd(IN_ROWS * RS);
    constexpr int W2 = WIN / 2;
    constexpr int PLEN = COUT * CIN * 9 + COUT;
    static_assert(RT % 2 == 0, "row tiles split over two wave groups");
    __shared__ __attribute__((aligned(16))) float s_in[CIN * PS + 32];
    __shared__ __attribute__((aligned(16))) float s_z[COUT * PSZ + 32];
    __shared__ float s_ss[2 * CIN];

    // twelve waves, three per SIMD: kernel row ky (3) x half of the co tiles rg (2) x { ci tile (CT = 2) | half of the pixel steps
    // (CT = 1: the two halves write separate partials) } - every wave has the same number of MFMAs and a workgroup fills the CU evenly
    // (six waves were 2,2,1,1 per SIMD: the barrier waited for the doubly loaded ones)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int ky = wave >> 2, rg = (wave >> 1) & 1, sel = wave & 1;
    const int cg = (CT == 2) ? sel : 0;             // this wave's ci tile
    const int kh = (CT == 1) ? sel : 0;             // this wave's K half (WIDE layers: the row pair rp = kh)
    static_assert((CT == 1) == WIDE, "K halves are the two row pairs of a wide band; a narrow band is one row group");

    for (int u = tid; u < CIN * PS + 32; u += NT) s_in[u] = 0.f;
    for (int u = tid; u < COUT * PSZ + 32; u += NT) s_z[u] = 0.f;

    f32x4 acc[RTW][3], accb[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
        accb[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[rt][kx] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int aq = WIDE ? (q >> 1) * RSZ + 16 * (q & 1) : q * RSZ;
    const int bq = WIDE ? (q >> 1) * 2 * RS + 16 * (q & 1) : q * 2 * RS;
    const float* ab = s_z + (16 * rg * RTW + m) * PSZ + aq + (CT == 1 ? kh * 2 * RSZ : 0);
    const float* bb = s_in + (16 * cg + m) * PS + bq + ky * RS + (CT == 1 ? kh * 4 * RS : 0);

    // Staging, slot-major: a thread owns ONE position of a plane (input: (row, column pair) of the band; dz: (row, column) of the
    // band) and walks the planes - every address is (per-thread base, computed once per item) + (plane step: a constant scalar
    // offset of the load / an immediate offset of the LDS store), so staging costs no address arithmetic per element.
    constexpr int SLOTS_IN = IN_ROWS * W2;               // 270 / 135 float2 per input plane
    constexpr int PG = WIDE ? 2 : 4;                     // plane groups walking in parallel (540 of the 768 threads)
    static_assert(PG * SLOTS_IN <= NT, "plane groups fit the workgroup");
    constexpr int IN_IT = CIN / PG;
    static_assert(CIN % PG == 0, "planes split evenly over the plane groups");
    const int ipg = tid / SLOTS_IN, islot = tid - ipg * SLOTS_IN;
    const int irow = islot / W2, ij = islot - irow * W2;
    const bool in_act = tid < PG * SLOTS_IN;
    float* const in_dst = s_in + ipg * PS + irow * RS + ij;
    constexpr int ZV = WIDE ? 2 : 1;                     // dz row width 30: float2 units; 15: single floats (rows are 4-byte aligned)
    constexpr int SLOTS_Z = R * WO / ZV;                 // 60 per dz plane
    constexpr int ZG = NT / 64;                          // 12 plane groups of 64 threads (60 active)
    constexpr int Z_IT = (COUT + ZG - 1) / ZG;
    static_assert(SLOTS_Z <= 64, "one dz slot per lane");
    const int zg = tid >> 6, zslot = tid & 63;
    const int zlr = zslot / (WO / ZV), zc = (zslot - zlr * (WO / ZV)) * ZV;
    float* const z_dst = s_z + zg * PSZ + zlr * RSZ + zc;
    float2 vin[IN_IT];
    float vz[Z_IT][ZV];
    auto fetch = [&](int item) {
        const int n = item / bands, band = item - n * bands;
        const int oy0 = band * R;
        const __amdgpu_buffer_rsrc_t rz = buf_of(dz + (size_t)n * COUT * HO * WO, COUT * HO * WO * 4);
        const __amdgpu_buffer_rsrc_t rx = buf_of(x + (size_t)n * CIN * HIN * WIN, CIN * HIN * WIN * 4);
        const int iy = 2 * oy0 - 1 + irow;
        const unsigned xoff = (in_act && iy >= 0 && iy < HIN) ? (unsigned)(((ipg * HIN + iy) * WIN + 2 * ij) * 4) : kOob;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) vin[it] = buf_f2(rx, xoff, it * PG * HIN * WIN * 4);
        const unsigned zoff = (zslot < SLOTS_Z && oy0 + zlr < HO) ? (unsigned)(((zg * HO + oy0 + zlr) * WO + zc) * 4) : kOob;
#pragma unroll
        for (int it = 0; it < Z_IT; ++it) {
            const unsigned o = (it * ZG + ZG <= COUT || zg + it * ZG < COUT) ? zoff : kOob;
            if (ZV == 2) {
                const float2 v = buf_f2(rz, o, it * ZG * HO * WO * 4);
                vz[it][0] = v.x;
                vz[it][ZV - 1] = v.y;
            } else {
                vz[it][0] = buf_f1(rz, o, it * ZG * HO * WO * 4);
            }
        }
    };
    auto stash = [&](int item) {
        const int band = item % bands;
        const int oy0 = band * R;
        if (in_act) {
            const int iy = 2 * oy0 - 1 + irow;
            const bool inside = iy >= 0 && iy < HIN;
#pragma unroll
            for (int it = 0; it < IN_IT; ++it) {
                float2 v = vin[it];
                if (APPLY) {
                    if (inside) {
                        const float sc = s_ss[ipg + it * PG], sh = s_ss[CIN + ipg + it * PG];
                        v.x = fmaxf(v.x, 0.f) * sc + sh;
                        v.y = fmaxf(v.y, 0.f) * sc + sh;
                    }
                }
                in_dst[it * PG * PS + EO] = v.x;
                in_dst[it * PG * PS + 1] = v.y;
            }
        }
        if (zslot < SLOTS_Z) {
#pragma unroll
            for (int it = 0; it < Z_IT; ++it) {
                if (it * ZG + ZG <= COUT || zg + it * ZG < COUT) {
                    z_dst[it * ZG * PSZ] = vz[it][0];
                    if (ZV == 2) z_dst[it * ZG * PSZ + 1] = vz[it][ZV - 1];
                }
            }
        }
    };
    if (APPLY) {
        for (int u = tid; u < CIN; u += NT) { s_ss[u] = scale[u]; s_ss[CIN + u] = shift[u]; }
    }
    fetch(blockIdx.x);
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        __syncthreads();
        stash(item);
        __syncthreads();
        if (item + (int)gridDim.x < items) fetch(item + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay ahead of the MFMA loop
        // the wave's pixel steps: the whole band (CT = 2, narrow: one row group) or its row pair (CT = 1; folded into ab / bb)
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            float a[RTW];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) a[rt] = ab[(16 * rt) * PSZ + j];
            float b[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) b[kx] = bb[(kx == 1 ? EO : 0) + (kx == 2 ? 1 : 0) + j];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[rt][kx] = AG_MFMA4(a[rt], b[kx], acc[rt][kx]);
                if (BIAS && ky == 0 && cg == 0) accb[rt] = AG_MFMA4(a[rt], 1.0f, accb[rt]);
            }
        }
    }
    // one partial per workgroup - two (the K halves) where CT = 1
    float* part = partials + ((size_t)blockIdx.x * (CT == 1 ? 2 : 1) + kh) * PLEN;
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * (rg * RTW + rt) + 4 * q + i;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) part[((size_t)(co * CIN + 16 * cg + m) * 3 + ky) * 3 + kx] = acc[rt][kx][i];
            if (BIAS && ky == 0 && cg == 0 && m == 0) part[COUT * CIN * 9 + co] = accb[rt][i];
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// first layer, forward: Conv2d(1, 16, 5, stride 2, pad 2) on (212, 120) -> (16, 106, 60).  One lane per output pixel of a row
// (60 of 64 lanes), 16 accumulators per row, two rows per wave; the 400 weights are wave-uniform (scalar loads, [tap][co]).
// Bound: the 1.93 GB of output per 4 750 images.
// NORM: the image normaliser of the policy (clamp((x - mean) / std, -5, 5) with per-pixel statistics - the division as v_rcp_f32
// (1 ulp) and a multiply: the IEEE division sequence was a tenth of this kernel's vector-ALU work -, model
// a2c_continuous_logstd_model.py:106-114 / running_mean_std.py:78-79) applied while the band is staged: the kernels take the RAW
// image and the normalised copy is never written (mean / std are 100 KB each and stay in L2).
template <bool NORM>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ nmean,
                                                       const float* __restrict__ nstd, const float* __restrict__ w1,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ stats, const long long* __restrict__ index) {
    // One workgroup per image walks its 14 bands of 8 output rows (the next band's rows are loaded while this one is convolved);
    // with `stats`, every lane keeps running sums of relu(y), relu(y)^2 per channel and the workgroup reduces them ONCE at the end:
    // stats[n][16][2], the batch statistics of the following ReLU + BatchNorm without another pass over the 1.9 GB output.
    constexpr int HIN = 212, WIN = 120, HO = 106, WO = 60, ROWS = 8, IN_ROWS = 2 * ROWS + 3, EO = 66, RS = 132;
    constexpr int BANDS = (HO + ROWS - 1) / ROWS;
    __shared__ __attribute__((aligned(16))) float s_in[IN_ROWS * RS + 8];
    __shared__ float s_red[4][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    const float* xin = x + (size_t)(index ? index[n] : (long long)n) * HIN * WIN;      // index: image n lives at row index[n] of x
    constexpr int IN_UNITS = IN_ROWS * (WIN / 2), IN_IT = (IN_UNITS + 255) / 256;
    float2 vin[IN_IT], vm[NORM ? IN_IT : 1], vs[NORM ? IN_IT : 1];
    const __amdgpu_buffer_rsrc_t rx = buf_of(xin, HIN * WIN * 4);
    const __amdgpu_buffer_rsrc_t rm = buf_of(NORM ? nmean : x, HIN * WIN * 4), rs = buf_of(NORM ? nstd : x, HIN * WIN * 4);
    const __amdgpu_buffer_rsrc_t ry = buf_of(y + (size_t)n * 16 * HO * WO, 16 * HO * WO * 4);
    auto fetch = [&](int band) {
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int u = tid + it * 256;
            const int row = u / (WIN / 2), j = u - row * (WIN / 2);
            const int iy = 2 * band * ROWS - 2 + row;
            const unsigned off = (u < IN_UNITS && iy >= 0 && iy < HIN) ? (unsigned)((iy * WIN + 2 * j) * 4) : kOob;
            vin[it] = buf_f2(rx, off);
            if (NORM) { vm[it] = buf_f2(rm, off); vs[it] = buf_f2(rs, off); }
        }
    };
    auto stash = [&](int band) {
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int u = tid + it * 256;
            if (u >= IN_UNITS) continue;
            const int row = u / (WIN / 2), j = u - row * (WIN / 2);
            float* rowp = s_in + row * RS;
            float2 v = vin[it];
            if (NORM) {
                const int iy = 2 * band * ROWS - 2 + row;
                if (iy >= 0 && iy < HIN) {
                    v.x = fminf(fmaxf((v.x - vm[it].x) * __builtin_amdgcn_rcpf(vs[it].x), -5.f), 5.f);
                    v.y = fminf(fmaxf((v.y - vm[it].y) * __builtin_amdgcn_rcpf(vs[it].y), -5.f), 5.f);
                }
            }
            rowp[j + 1] = v.x;            // c = ix + 2 = 2j + 2
            rowp[EO + j + 1] = v.y;       // c = 2j + 3
            if (j == 0) { rowp[0] = 0.f; rowp[EO] = 0.f; rowp[WIN / 2 + 1] = 0.f; rowp[EO + WIN / 2 + 1] = 0.f; }
        }
    };
    float ssum[16], ssq[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) ssum[co] = ssq[co] = 0.f;
    const float* p0 = s_in + (4 * wave) * RS + lane;
    fetch(0);
    for (int band = 0; band < BANDS; ++band) {
        __syncthreads();
        stash(band);
        __syncthreads();
        if (band + 1 < BANDS) fetch(band + 1);
        __builtin_amdgcn_sched_barrier(0);
        float acc[2][16];
#pragma unroll
        for (int co = 0; co < 16; ++co) { acc[0][co] = bias[co]; acc[1][co] = acc[0][co]; }
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky)            // not unrolled: 80 weights (scalar registers) live per kernel row, not 400
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int off = (kx & 1) ? EO + (kx >> 1) : (kx >> 1);
                const float v0 = p0[ky * RS + off], v1 = p0[(2 + ky) * RS + off];
#pragma unroll
                for (int co = 0; co < 16; ++co) {
                    const float wv = w1[(ky * 5 + kx) * 16 + co];
                    acc[0][co] = fmaf(wv, v0, acc[0][co]);
                    acc[1][co] = fmaf(wv, v1, acc[1][co]);
                }
            }
        if (lane < WO) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = band * ROWS + 2 * wave + r;
                if (oy >= HO) continue;
                const unsigned yoff = (unsigned)((oy * WO + lane) * 4);
#pragma unroll
                for (int co = 0; co < 16; ++co) {
                    // buffer store: one per-lane offset for the row, the channel plane as a scalar offset - no address arithmetic
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[r][co]), ry, yoff, co * (HO * WO * 4), 0);
                    const float rl = fmaxf(acc[r][co], 0.f);
                    ssum[co] += rl;
                    ssq[co] = fmaf(rl, rl, ssq[co]);
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int co = 0; co < 16; ++co) {
            const float a = row_sum16(ssum[co]), b = row_sum16(ssq[co]);
            if ((lane & 15) == 0) {
                s_red[wave][lane >> 4][2 * co] = a;
                s_red[wave][lane >> 4][2 * co + 1] = b;
            }
        }
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += (&s_red[0][0][0])[k * 32 + tid];
            stats[(size_t)n * 32 + tid] = t;
        }
    }
}

// first layer, weight gradient: dw[co][tap] = sum_{n,oy,ox} dz[co][oy][ox] in[2oy+ky-2][2ox+kx-2]; D[co][tap] = two 16 x 16 tiles
// (taps 0-15, 16-24; column 25 of the second tile is the constant 1 -> db[co]), K = pixels: ox = 4j + quarter.
// Four waves split the 8 rows of a band; persistent workgroups, one partial [16][32] each.
// BNBWD: dz is not read but formed while staging from the gradient dy of the layer's ReLU + BatchNorm output and the layer's own
// output x1 (ag_relu_bn_bwd_dx's arithmetic folded per channel: dz = [x1 > 0] (A dy + m_i (B x1 + C)), tab[c] = {A, B, C, 0},
// m_i = image multiplicity) - the 1.9 GB gradient of the first convolution's output is never written.  NORM: see conv1_fwd_kernel.
template <bool NORM, bool BNBWD>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x1,
                                                         const float* __restrict__ tab, const float* __restrict__ wts,
                                                         const float* __restrict__ x, const float* __restrict__ nmean,
                                                         const float* __restrict__ nstd, float* __restrict__ partials,
                                                         const long long* __restrict__ index, int items, int bands) {
    constexpr int HIN = 212, WIN = 120, HO = 106, WO = 60, ROWS = 8, IN_ROWS = 2 * ROWS + 3;
    constexpr int EO = 68, RS = 136, PSZ = 482;     // RS = 8, EO = 4, PSZ = 2 (mod 32): conflict-free operand reads
    __shared__ __attribute__((aligned(16))) float s_in[IN_ROWS * RS + 8];
    __shared__ __attribute__((aligned(16))) float s_z[16 * PSZ + 8];
    __shared__ float s_tab[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    for (int u = tid; u < IN_ROWS * RS + 8; u += 256) s_in[u] = 0.f;
    for (int u = tid; u < 16 * PSZ + 8; u += 256) s_z[u] = 0.f;
    int boff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tap = m + 16 * t, kyy = tap / 5, kxx = tap - 5 * kyy;
        boff[t] = (tap < 25) ? kyy * RS + (kxx & 1) * EO + (kxx >> 1) + q : 0;
    }
    const bool t1_data = (m + 16) < 25;
    const float t1_const = (m + 16 == 25) ? 1.f : 0.f;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // staging: dz slot-major (a thread owns one (row, column pair) of the band and walks the 16 channel planes); the input band is
    // one plane of 19 rows, five column pairs per thread
    constexpr int Z_SLOTS = ROWS * (WO / 2);
    static_assert(Z_SLOTS <= 256, "one dz slot per thread");
    constexpr int IN_UNITS = IN_ROWS * (WIN / 2), IN_IT = (IN_UNITS + 255) / 256;
    const int zlr = tid / (WO / 2), zj = tid - zlr * (WO / 2);
    const bool z_act = tid < Z_SLOTS;
    float* const z_dst = s_z + zlr * WO + 2 * zj;
    float2 vz[16], vx[BNBWD ? 16 : 1], vin[IN_IT], vm[NORM ? IN_IT : 1], vs[NORM ? IN_IT : 1];
    const __amdgpu_buffer_rsrc_t rm = buf_of(NORM ? nmean : x, HIN * WIN * 4), rs = buf_of(NORM ? nstd : x, HIN * WIN * 4);
    if (BNBWD) {
        if (tid < 64) s_tab[tid] = tab[tid];
    }
    auto fetch = [&](int item) {
        const int n = item / bands, band = item - n * bands;
        const int oy0 = band * ROWS;
        const __amdgpu_buffer_rsrc_t rz = buf_of(dz + (size_t)n * 16 * HO * WO, 16 * HO * WO * 4);
        const __amdgpu_buffer_rsrc_t rz1 = buf_of((BNBWD ? x1 : dz) + (size_t)n * 16 * HO * WO, 16 * HO * WO * 4);
        const __amdgpu_buffer_rsrc_t rx = buf_of(x + (size_t)(index ? index[n] : (long long)n) * HIN * WIN, HIN * WIN * 4);
        const unsigned zoff = (z_act && oy0 + zlr < HO) ? (unsigned)(((oy0 + zlr) * WO + 2 * zj) * 4) : kOob;
#pragma unroll
        for (int co = 0; co < 16; ++co) {
            vz[co] = buf_f2(rz, zoff, co * (HO * WO * 4));
            if (BNBWD) vx[co] = buf_f2(rz1, zoff, co * (HO * WO * 4));
        }
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int u = tid + it * 256;
            const int row = u / (WIN / 2), j = u - row * (WIN / 2);
            const int iy = 2 * oy0 - 2 + row;
            const unsigned off = (u < IN_UNITS && iy >= 0 && iy < HIN) ? (unsigned)((iy * WIN + 2 * j) * 4) : kOob;
            vin[it] = buf_f2(rx, off);
            if (NORM) { vm[it] = buf_f2(rm, off); vs[it] = buf_f2(rs, off); }
        }
    };
    auto stash = [&](int item) {
        const int n = item / bands, band = item - n * bands;
        const int oy0 = band * ROWS;
        if (z_act) {
            const float wi = (BNBWD && wts) ? wts[n] : 1.0f;
#pragma unroll
            for (int co = 0; co < 16; ++co) {
                float2 v = vz[co];
                if (BNBWD) {
                    const float a = s_tab[co * 4 + 0], b = s_tab[co * 4 + 1], c = s_tab[co * 4 + 2];
                    const float2 xv = vx[co];
                    v.x = xv.x > 0.f ? fmaf(v.x, a, wi * fmaf(xv.x, b, c)) : 0.f;
                    v.y = xv.y > 0.f ? fmaf(v.y, a, wi * fmaf(xv.y, b, c)) : 0.f;
                }
                *reinterpret_cast<float2*>(z_dst + co * PSZ) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int u = tid + it * 256;
            if (u >= IN_UNITS) continue;
            const int row = u / (WIN / 2), j = u - row * (WIN / 2);
            float2 v = vin[it];
            if (NORM) {
                const int iy = 2 * oy0 - 2 + row;
                if (iy >= 0 && iy < HIN) {
                    v.x = fminf(fmaxf((v.x - vm[it].x) * __builtin_amdgcn_rcpf(vs[it].x), -5.f), 5.f);
                    v.y = fminf(fmaxf((v.y - vm[it].y) * __builtin_amdgcn_rcpf(vs[it].y), -5.f), 5.f);
                }
            }
            s_in[row * RS + j + 1] = v.x;
            s_in[row * RS + EO + j + 1] = v.y;
        }
    };
    fetch(blockIdx.x);
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        __syncthreads();
        stash(item);
        __syncthreads();
        if (item + (int)gridDim.x < items) fetch(item + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay ahead of the MFMA loop
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int lr = 2 * wave + rr;
            const float* ap = s_z + m * PSZ + lr * WO + q;
            const float* bp = s_in + (2 * lr) * RS;
#pragma unroll 5
            for (int j = 0; j < 15; ++j) {
                const float a = ap[4 * j];
                const float b0 = bp[boff[0] + 4 * j];
                float b1 = bp[boff[1] + 4 * j];
                b1 = t1_data ? b1 : t1_const;
                acc0 = AG_MFMA4(a, b0, acc0);
                acc1 = AG_MFMA4(a, b1, acc1);
            }
        }
    }
    // cross-wave sum through LDS (s_z is free after the last barrier below)
    __syncthreads();
    float* red = s_z;     // [4 waves][16 co][32 taps]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[(wave * 16 + 4 * q + i) * 32 + m] = acc0[i];
        red[(wave * 16 + 4 * q + i) * 32 + 16 + m] = acc1[i];
    }
    __syncthreads();
    for (int u = tid; u < 512; u += 256)
        partials[(size_t)blockIdx.x * 512 + u] = red[u] + red[512 + u] + red[1024 + u] + red[1536 + u];
}

// ------------------------------------------------------------------------------------------------------------------------
struct Shape { int cin, cout, hin, win; };
constexpr Shape kL2 = {16, 32, 106, 60}, kL3 = {32, 64, 53, 30};
inline int layer_of(int cin, int cout, int hin, int win) {
    if (cin == kL2.cin && cout == kL2.cout && hin == kL2.hin && win == kL2.win) return 2;
    if (cin == kL3.cin && cout == kL3.cout && hin == kL3.hin && win == kL3.win) return 3;
    return 0;
}
// output rows per band = 2 x waves: 53 = 7 bands of 8 (-3), 27 = 2 bands of 14 (-1); 2 / 7 and 4 / 5 waves measured slower
constexpr int kL2Waves = 4, kL3Waves = 7;
constexpr int kWgradWorkgroups = 512;              // persistent 12-wave workgroups: two per CU where registers allow (conv2), else they queue

}  // namespace

#define AG_CONV_LAUNCH_OK() (hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP)

extern "C" int ag_cnn_conv_workspace_floats(int cin, int cout) {
    if (cin == 1 && cout == 16) return 400;
    return 2 * 9 * cin * cout;      // (the split forward's fragment image: three bf16 planes = 6 bytes per weight)
}

extern "C" int ag_cnn_conv1_fwd(const float* x_dev, const long long* index_dev, const float* norm_mean_dev, const float* norm_std_dev,
                                const float* w_dev, const float* b_dev, float* y_dev, float* stats_dev, int n, float* workspace_dev,
                                void* stream) {
    if (!x_dev || !w_dev || !b_dev || !y_dev || !workspace_dev || n <= 0 || (!norm_mean_dev) != (!norm_std_dev)) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pack_conv1_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w_dev, workspace_dev);
    if (norm_mean_dev)
        hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3(n), dim3(256), 0, (hipStream_t)stream, x_dev, norm_mean_dev, norm_std_dev,
                           workspace_dev, b_dev, y_dev, stats_dev, index_dev);
    else
        hipLaunchKernelGGL(conv1_fwd_kernel<false>, dim3(n), dim3(256), 0, (hipStream_t)stream, x_dev, norm_mean_dev, norm_std_dev,
                           workspace_dev, b_dev, y_dev, stats_dev, index_dev);
    return AG_CONV_LAUNCH_OK();
}

extern "C" int ag_cnn_conv1_wgrad_partials(int n) {
    const long long items = (long long)n * ((106 + 7) / 8);
    return (int)(items < 768 ? items : 768);       // three workgroups per CU
}

extern "C" int ag_cnn_conv1_wgrad(const float* dz_dev, const float* bn_x_dev, const float* bn_tab_dev, const float* weights_dev,
                                  const float* x_dev, const long long* index_dev, const float* norm_mean_dev, const float* norm_std_dev,
                                  float* partials_dev, int n, void* stream) {
    if (!dz_dev || !x_dev || !partials_dev || n <= 0 || (!norm_mean_dev) != (!norm_std_dev) || (!bn_x_dev) != (!bn_tab_dev))
        return AG_ERR_INVALID_ARG;
    const int bands = (106 + 7) / 8;
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    const dim3 grid(ag_cnn_conv1_wgrad_partials(n)), block(256);
    const bool norm = norm_mean_dev != nullptr, bn = bn_x_dev != nullptr;
#define AG_C1W(NORM_, BN_)                                                                                                        \
    hipLaunchKernelGGL((conv1_wgrad_kernel<NORM_, BN_>), grid, block, 0, (hipStream_t)stream, dz_dev, bn_x_dev, bn_tab_dev, weights_dev, \
                       x_dev, norm_mean_dev, norm_std_dev, partials_dev, index_dev, n * bands, bands)
    if (norm && bn) AG_C1W(true, true);
    else if (norm) AG_C1W(true, false);
    else if (bn) AG_C1W(false, true);
    else AG_C1W(false, false);
#undef AG_C1W
    return AG_CONV_LAUNCH_OK();
}

extern "C" int ag_cnn_conv_supported(int cin, int cout, int hin, int win) { return layer_of(cin, cout, hin, win) != 0; }

extern "C" int ag_cnn_conv_fwd_bands(int cin, int cout, int hin, int win) {
    const int layer = layer_of(cin, cout, hin, win);
    if (layer == 2) return (53 + 2 * kL2Waves - 1) / (2 * kL2Waves);
    if (layer == 3) return (27 + 2 * kL3Waves - 1) / (2 * kL3Waves);
    return AG_ERR_UNSUPPORTED;
}

extern "C" int ag_cnn_conv_fwd(const float* x_dev, const float* scale_dev, const float* shift_dev, const float* w_dev,
                               const float* b_dev, float* y_dev, float* stats_dev, int n, int cin, int cout, int hin, int win,
                               float* workspace_dev, void* stream) {
    if (!x_dev || !w_dev || !b_dev || !y_dev || !workspace_dev || n <= 0 || (!scale_dev) != (!shift_dev)) return AG_ERR_INVALID_ARG;
    const int layer = layer_of(cin, cout, hin, win);
    if (!layer) return AG_ERR_UNSUPPORTED;
    const int tot = 9 * cin * cout;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_dev, workspace_dev, cin, cout);
    const bool apply = scale_dev != nullptr;
    const int bands = ag_cnn_conv_fwd_bands(cin, cout, hin, win);
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
#define AG_CF(CIN_, COUT_, HIN_, WIN_, WAVES_, APPLY_)                                                                           \
    hipLaunchKernelGGL((conv_s2_fwd_kernel<CIN_, COUT_, HIN_, WIN_, WAVES_, APPLY_>), dim3(n * bands), dim3(WAVES_ * 64), 0,      \
                       (hipStream_t)stream, x_dev, workspace_dev, b_dev, scale_dev, shift_dev, y_dev, stats_dev, bands)
    if (layer == 2) {
        if (apply) AG_CF(16, 32, 106, 60, kL2Waves, true);
        else AG_CF(16, 32, 106, 60, kL2Waves, false);
    } else {
        if (apply) AG_CF(32, 64, 53, 30, kL3Waves, true);
        else AG_CF(32, 64, 53, 30, kL3Waves, false);
    }
#undef AG_CF
    return AG_CONV_LAUNCH_OK();
}

// The same layer on the bf16 matrix cores at float32 accuracy (conv_s2_fwd_split_kernel): bands of 4 output rows, persistent workgroups.
extern "C" int ag_cnn_conv_fwd_split_bands(int cin, int cout, int hin, int win) {
    const int layer = layer_of(cin, cout, hin, win);
    if (layer == 2) return (53 + 3) / 4;
    if (layer == 3) return (27 + 3) / 4;
    return AG_ERR_UNSUPPORTED;
}

extern "C" int ag_cnn_conv_fwd_split(const float* x_dev, const float* scale_dev, const float* shift_dev, const float* w_dev,
                                     const float* b_dev, float* y_dev, float* stats_dev, int n, int cin, int cout, int hin, int win,
                                     float* workspace_dev, void* stream) {
    if (!x_dev || !w_dev || !b_dev || !y_dev || !workspace_dev || n <= 0 || (!scale_dev) != (!shift_dev)) return AG_ERR_INVALID_ARG;
    if ((uintptr_t)workspace_dev & 15) return AG_ERR_INVALID_ARG;
    const int layer = layer_of(cin, cout, hin, win);
    if (!layer) return AG_ERR_UNSUPPORTED;
    const int units = (cout / 32) * 9 * (cin / 16) * 64;
    hipLaunchKernelGGL(pack_fwd_split_kernel, dim3((units + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_dev, (uint4*)workspace_dev,
                       cin, cout);
    const bool apply = scale_dev != nullptr;
    const int bands = ag_cnn_conv_fwd_split_bands(cin, cout, hin, win);
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    static int cus_of[64] = {0};          // CU count per device ordinal
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus_of[dev] == 0)
        cus_of[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    const int cus = cus_of[dev];
#define AG_CFS(CIN_, COUT_, HIN_, WIN_, APPLY_, PER_CU_)                                                                          \
    do {                                                                                                                          \
        const long long g = (long long)cus * (PER_CU_);                                                                           \
        const bool im = (long long)n >= 8 * g;                                                                                    \
        const long long work = im ? (long long)n : (long long)n * bands;                                                          \
        hipLaunchKernelGGL((conv_s2_fwd_split_kernel<CIN_, COUT_, HIN_, WIN_, APPLY_>), dim3((unsigned)(work < g ? work : g)),        \
                           dim3(SplitFwdShape<CIN_, COUT_, WIN_>::NT), 0, (hipStream_t)stream, x_dev, (const uint4*)workspace_dev,  \
                           b_dev, scale_dev, shift_dev, y_dev, stats_dev, n, im ? 1 : 0);                                        \
    } while (0)
    if (layer == 2) {
        if (apply) AG_CFS(16, 32, 106, 60, true, 1);
        else AG_CFS(16, 32, 106, 60, false, 1);
    } else {
        if (apply) AG_CFS(32, 64, 53, 30, true, 1);
        else AG_CFS(32, 64, 53, 30, false, 1);
    }
#undef AG_CFS
    return AG_CONV_LAUNCH_OK();
}

extern "C" int ag_cnn_conv_dgrad(const float* dz_dev, const float* w_dev, float* dx_dev, int n, int cin, int cout, int hin, int win,
                                 float* workspace_dev, void* stream) {
    if (!dz_dev || !w_dev || !dx_dev || !workspace_dev || n <= 0) return AG_ERR_INVALID_ARG;
    const int layer = layer_of(cin, cout, hin, win);
    if (!layer) return AG_ERR_UNSUPPORTED;
    const int tot = 9 * cin * cout;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_dev, workspace_dev, cin, cout);
    const float* none = nullptr;
    if (layer == 2) {
        const int bands = (53 + 2 * kL2Waves - 1) / (2 * kL2Waves);
        if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv_s2_dgrad_kernel<16, 32, 106, 60, kL2Waves, 0>), dim3(n * bands), dim3(kL2Waves * 64), 0,
                           (hipStream_t)stream, dz_dev, workspace_dev, dx_dev, bands, none, none, none, (float*)nullptr, Conv1Side{});
    } else {
        const int bands = (27 + 2 * kL3Waves - 1) / (2 * kL3Waves);
        if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv_s2_dgrad_kernel<32, 64, 53, 30, kL3Waves, 0>), dim3(n * bands), dim3(kL3Waves * 64), 0,
                           (hipStream_t)stream, dz_dev, workspace_dev, dx_dev, bands, none, none, none, (float*)nullptr, Conv1Side{});
    }
    return AG_CONV_LAUNCH_OK();
}

// Workgroups (= rows of `sums`) of ag_cnn_conv_dgrad_bn for n images.
extern "C" int ag_cnn_conv_dgrad_bn_rows(int n, int cin, int cout, int hin, int win) {
    if (layer_of(cin, cout, hin, win) != 3) return AG_ERR_UNSUPPORTED;
    const long long rows = (long long)n * ((27 + 2 * kL3Waves - 1) / (2 * kL3Waves));
    return rows > 0x7fffffffLL ? AG_ERR_UNSUPPORTED : (int)rows;
}

// ag_cnn_conv_dgrad followed by the backward of the ReLU + BatchNorm in front of the layer, in the kernel's epilogue (the third
// layer only: the second layer's input gradient is consumed by ag_cnn_conv1_wgrad, which folds the same arithmetic into its staging).
// dx <- [bn_x > 0] (A g + m_i (B bn_x + C)) with g the convolution's input gradient, bn_tab [cin][4] = {A, B, C, 0} as
// ag_bn_bwd_prep(mode 1) writes it, weights [n] the image multiplicities or NULL.  sums (NULL = not wanted):
// [ag_cnn_conv_dgrad_bn_rows][cin][6] = per workgroup {total, row 0, last row, column 0, (0,0), (last row, 0)} of dx.
extern "C" int ag_cnn_conv_dgrad_bn(const float* dz_dev, const float* w_dev, const float* bn_x_dev, const float* bn_tab_dev,
                                    const float* weights_dev, float* dx_dev, float* sums_dev, int n, int cin, int cout, int hin, int win,
                                    float* workspace_dev, void* stream) {
    if (!dz_dev || !w_dev || !bn_x_dev || !bn_tab_dev || !dx_dev || !workspace_dev || n <= 0) return AG_ERR_INVALID_ARG;
    if (layer_of(cin, cout, hin, win) != 3) return AG_ERR_UNSUPPORTED;
    const int bands = (27 + 2 * kL3Waves - 1) / (2 * kL3Waves);
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    const int tot = 9 * cin * cout;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_dev, workspace_dev, cin, cout);
    if (sums_dev)
        hipLaunchKernelGGL((conv_s2_dgrad_kernel<32, 64, 53, 30, kL3Waves, 2>), dim3(n * bands), dim3(kL3Waves * 64), 0,
                           (hipStream_t)stream, dz_dev, workspace_dev, dx_dev, bands, bn_x_dev, bn_tab_dev, weights_dev, sums_dev, Conv1Side{});
    else
        hipLaunchKernelGGL((conv_s2_dgrad_kernel<32, 64, 53, 30, kL3Waves, 1>), dim3(n * bands), dim3(kL3Waves * 64), 0,
                           (hipStream_t)stream, dz_dev, workspace_dev, dx_dev, bands, bn_x_dev, bn_tab_dev, weights_dev, sums_dev, Conv1Side{});
    return AG_CONV_LAUNCH_OK();
}

// Rows of `partials` ag_cnn_conv_dgrad_conv1_wgrad writes for n images (one [16][32] block per workgroup).
extern "C" int ag_cnn_conv_dgrad_conv1_wgrad_partials(int n) {
    const long long rows = (long long)n * ((53 + 2 * kL2Waves - 1) / (2 * kL2Waves));
    return (n <= 0 || rows > 0x7fffffffLL) ? AG_ERR_UNSUPPORTED : (int)rows;
}

// The second convolution's input gradient, the first layer's ReLU + BatchNorm backward and the first convolution's weight gradient in
// one kernel (ag_cnn_conv_dgrad(16, 32, 106, 60) + ag_cnn_conv1_wgrad with bn_x / bn_tab, without the 1.9 GB tensor between them):
// partials [ag_cnn_conv_dgrad_conv1_wgrad_partials(n)][16][32], columns 0-24 = dw1 [16][5][5], column 25 = db1; the caller sums the rows.
extern "C" int ag_cnn_conv_dgrad_conv1_wgrad(const float* dz_dev, const float* w_dev, const float* bn_x_dev, const float* bn_tab_dev,
                                             const float* weights_dev, const float* x_dev, const long long* index_dev,
                                             const float* norm_mean_dev, const float* norm_std_dev, float* partials_dev, int n,
                                             float* workspace_dev, void* stream) {
    if (!dz_dev || !w_dev || !bn_x_dev || !bn_tab_dev || !x_dev || !partials_dev || !workspace_dev || n <= 0 ||
        (!norm_mean_dev) != (!norm_std_dev))
        return AG_ERR_INVALID_ARG;
    const int bands = (53 + 2 * kL2Waves - 1) / (2 * kL2Waves);
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3((9 * 16 * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_dev, workspace_dev, 16, 32);
    const Conv1Side c1{x_dev, index_dev, norm_mean_dev, norm_std_dev, partials_dev};
    hipLaunchKernelGGL((conv_s2_dgrad_kernel<16, 32, 106, 60, kL2Waves, 3>), dim3(n * bands), dim3(kL2Waves * 64), 0, (hipStream_t)stream,
                       dz_dev, workspace_dev, (float*)nullptr, bands, bn_x_dev, bn_tab_dev, weights_dev, (float*)nullptr, c1);
    return AG_CONV_LAUNCH_OK();
}

static int wgrad_grid(int n, int hin) {
    const int ho = (hin - 1) / 2 + 1;
    const long long items = (long long)n * ((ho + 3) / 4);
    return (int)(items < kWgradWorkgroups ? items : kWgradWorkgroups);
}

extern "C" int ag_cnn_conv_wgrad_partials(int n, int cin, int cout, int hin, int win) {
    const int layer = layer_of(cin, cout, hin, win);
    if (!layer) return AG_ERR_UNSUPPORTED;
    return wgrad_grid(n, hin) * (cin == 16 ? 2 : 1);        // one partial per workgroup; two where the waves split the pixel steps
}

extern "C" int ag_cnn_conv_wgrad(const float* dz_dev, const float* x_dev, const float* scale_dev, const float* shift_dev,
                                 float* partials_dev, int with_bias, int n, int cin, int cout, int hin, int win, void* stream) {
    if (!dz_dev || !x_dev || !partials_dev || n <= 0 || (!scale_dev) != (!shift_dev)) return AG_ERR_INVALID_ARG;
    const int layer = layer_of(cin, cout, hin, win);
    if (!layer) return AG_ERR_UNSUPPORTED;
    const int ho = (hin - 1) / 2 + 1, bands = (ho + 3) / 4;
    if ((long long)n * bands > 0x7fffffffLL) return AG_ERR_UNSUPPORTED;
    const int g = wgrad_grid(n, hin);
    const bool apply = scale_dev != nullptr;
    const dim3 grid(g), block(768);
#define AG_CW(CIN_, COUT_, HIN_, WIN_, APPLY_, BIAS_)                                                                          \
    hipLaunchKernelGGL((conv_s2_wgrad_kernel<CIN_, COUT_, HIN_, WIN_, APPLY_, BIAS_>), grid, block, 0, (hipStream_t)stream, dz_dev, \
                       x_dev, scale_dev, shift_dev, partials_dev, n * bands, bands)
#define AG_CW2(CIN_, COUT_, HIN_, WIN_)                                                \
    do {                                                                               \
        if (apply && with_bias) AG_CW(CIN_, COUT_, HIN_, WIN_, true, true);            \
        else if (apply) AG_CW(CIN_, COUT_, HIN_, WIN_, true, false);                   \
        else if (with_bias) AG_CW(CIN_, COUT_, HIN_, WIN_, false, true);               \
        else AG_CW(CIN_, COUT_, HIN_, WIN_, false, false);                             \
    } while (0)
    if (layer == 2) AG_CW2(16, 32, 106, 60);
    else AG_CW2(32, 64, 53, 30);
#undef AG_CW2
#undef AG_CW
    return AG_CONV_LAUNCH_OK();
}
