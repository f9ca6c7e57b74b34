// split_wgrad.hip - the weight gradient of the 256 x 256 hidden layer on the bf16 matrix cores of gfx950 at float32 accuracy
// (autograd of lib/network/mlp.py:36-39 inside calc_gradients, lib/agent/a2c_continuous.py:299-369):
//
//     dW[co, ci] = sum_m dZ[m, co] * X[m, ci]         dZ, X : [M, 256] f32 row-major (M = minibatch rows, 196 608 at the bench)
//
// Same arithmetic as split_gemm.hip (every f32 operand split EXACTLY into three bf16 pieces, six of the nine cross products
// accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest first); what differs is that the contraction runs over the ROWS of
// both operands, i.e. over their strided dimension, and that both operands are activations (nothing can be pre-split).
//
//   * Work split: K = M is cut into `slices` contiguous row ranges (one workgroup each, one per CU); a workgroup owns the WHOLE
//     256 x 256 output for its rows, so each operand is read from HBM exactly once (402 MB at M = 196 608; the library's
//     128 x 128 macro-tile kernel reads both twice).  Slice s writes partial [s, 256, 256]; the caller sums the slices in a
//     fixed order (ag_sum_rows_multi): deterministic.
//   * Transposition happens in REGISTERS, for free: a wave fetches 4 consecutive rows x 256 columns as 4 fully coalesced 1 KiB
//     row reads (one float4 = 4 columns per lane per row); the 4 values a lane then holds for one column ARE four consecutive-k
//     elements of that column's 8-element fragment unit, so two split_pair() calls yield half (8 bytes) of each of the three
//     bf16x8 units the MFMA wants; the wave that loaded the other 4 rows writes the other half.
//   * LDS image per 16-row chunk, per operand: [plane 3][k-half 2][unit 256] x 16 B with column c at unit (c & 3) * 64 + (c >> 2):
//     the four half-units a lane produces land 64 units apart, so every ds_write_b64 instruction covers 64 consecutive units,
//     and fragment reads are 32 consecutive units (conflict-free) exactly as in split_gemm.hip.  The column
//     permutation is undone for free in the epilogue: wave (wm, wn) takes as its four column tiles the four (c & 3) classes of
//     logical columns 128 wn .. 128 wn + 127, so each lane ends up with four CONSECUTIVE columns per row = one 16-byte store;
//     the row permutation only selects which output row a register belongs to.
//   * 512 threads = 8 waves as 4 (rows of dW) x 2 (columns): 64 x 128 outputs per wave = eight 32 x 32 accumulator tiles (128
//     registers), two waves per SIMD.  Stages are double-buffered (2 x 48 KB); in every chunk each of the 8 waves stages one
//     (operand, k-half, row-half) piece of the NEXT chunk, its loads issued one chunk earlier, right after the registers they
//     fill were drained into LDS - identical staging work in every wave and every chunk, spread between the chunk's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "split_common.hpp"

namespace {

constexpr int WN = 256;                      // layer width (both dims of dW)
constexpr int WBK = 16;                      // rows per chunk
constexpr int OP_UNITS = 3 * 2 * WN;         // 16-byte units per operand per stage
constexpr int WSTAGE_UNITS = 2 * OP_UNITS;   // A (dZ) then B (X)
constexpr size_t kWgradLds = (size_t)2 * WSTAGE_UNITS * 16;      // 96 KB

template <bool ORDERED>
__global__ __launch_bounds__(512, 2) void split_wgrad_kernel(const float* __restrict__ dZ, const float* __restrict__ X,
                                                              float* __restrict__ partials, int M, int chunks_per_slice) {
    extern __shared__ uint4 lds[];           // [2 stages][A: OP_UNITS | B: OP_UNITS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: scalar branches below
    const int total_chunks = (M + WBK - 1) / WBK;
    const int c_begin = min((int)blockIdx.x * chunks_per_slice, total_chunks);
    const int n = min(c_begin + chunks_per_slice, total_chunks) - c_begin;      // chunks of this slice (may be 0)

    // ---- loader role (every wave, every chunk): operand, k-half, row half -> 4 rows x 256 columns = 4 coalesced 1 KiB row
    //      reads; the lane's 4 values of a column are k = 4 lg .. 4 lg + 3 of that column's 8-element fragment unit, i.e. its
    //      low or high 8 bytes (ds_write_b64).  All waves do the same amount of staging work in every chunk, so the compiler
    //      (and the issue-order directives below) can spread it between the chunk's MFMAs instead of behind them.
    const int lop = wave >> 2, lh = (wave >> 1) & 1, lg = wave & 1;
    const float* __restrict__ src = lop ? X : dZ;
    float4 va[4], vb[4];                     // chunk c is staged through set (c & 1): va = even chunks, vb = odd chunks
    int nva = 4, nvb = 4;                    // rows of a set that exist (uniform); < 4 only in the last chunk of the matrix
#define AG_WG_LOAD(t_, v, nvalid)                                                                    \
    do {                                                                                             \
        const int row0_ = (c_begin + (t_)) * WBK + 8 * lh + 4 * lg;                                  \
        nvalid = min(max(M - row0_, 0), 4);                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
            const int row_ = min(row0_ + r, M - 1);                                                  \
            v[r] = reinterpret_cast<const float4*>(src + (size_t)row_ * WN)[lane];                   \
        }                                                                                            \
    } while (0)
    // column q of this lane = logical column 4 lane + q -> unit q * 64 + lane; 4 rows -> half a unit (two packed words per plane)
#define AG_WG_HALF(x0, x1, x2, x3, q_)                                                               \
    do {                                                                                             \
        uint2 h1_, h2_, h3_;                                                                         \
        split_pair(x0, x1, h1_.x, h2_.x, h3_.x);                                                     \
        split_pair(x2, x3, h1_.y, h2_.y, h3_.y);                                                     \
        dst_[(0 * 2 * WN + (q_) * 64) * 2] = h1_;                                                    \
        dst_[(1 * 2 * WN + (q_) * 64) * 2] = h2_;                                                    \
        dst_[(2 * 2 * WN + (q_) * 64) * 2] = h3_;                                                    \
    } while (0)
#define AG_WG_WRITE(stage_, tail_, v, nvalid)                                                        \
    do {                                                                                             \
        if ((tail_) && nvalid < 4) {                                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                            \
                if (r >= nvalid) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);                             \
        }                                                                                            \
        uint2* dst_ = reinterpret_cast<uint2*>(lds + (stage_) * WSTAGE_UNITS + lop * OP_UNITS + lh * WN + lane) + lg; \
        AG_WG_HALF(v[0].x, v[1].x, v[2].x, v[3].x, 0);                                               \
        AG_WG_HALF(v[0].y, v[1].y, v[2].y, v[3].y, 1);                                               \
        AG_WG_HALF(v[0].z, v[1].z, v[2].z, v[3].z, 2);                                               \
        AG_WG_HALF(v[0].w, v[1].w, v[2].w, v[3].w, 3);                                               \
    } while (0)

    // ---- compute role: wave (wm, wn) owns unit rows 64 wm .. +63 of the A image and the four 32-unit groups
    //      q * 64 + 32 wn .. +31 (q = 0..3) of the B image
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // products smallest first (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1), the two row tiles of a column tile interleaved so that
    // consecutive MFMAs never share an accumulator
#define AG_WG_COMPUTE(stage_)                                                                          \
    do {                                                                                               \
        const uint4* sa_ = lds + (stage_) * WSTAGE_UNITS;                                              \
        const uint4* sb_ = sa_ + OP_UNITS;                                                             \
        bf16x8 a_[2][3];                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
                const uint4 u_ = sa_[(p * 2 + khalf) * WN + wm * 64 + i * 32 + l31];                   \
                a_[i][p] = *reinterpret_cast<const bf16x8*>(&u_);                                      \
            }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
            const uint4 ub0_ = sb_[(0 * 2 + khalf) * WN + j * 64 + wn * 32 + l31];                     \
            const uint4 ub1_ = sb_[(1 * 2 + khalf) * WN + j * 64 + wn * 32 + l31];                     \
            const uint4 ub2_ = sb_[(2 * 2 + khalf) * WN + j * 64 + wn * 32 + l31];                     \
            const bf16x8 b0_ = *reinterpret_cast<const bf16x8*>(&ub0_);                                \
            const bf16x8 b1_ = *reinterpret_cast<const bf16x8*>(&ub1_);                                \
            const bf16x8 b2_ = *reinterpret_cast<const bf16x8*>(&ub2_);                                \
            f32x16& d0_ = acc[j];                                                                      \
            f32x16& d1_ = acc[4 + j];                                                                  \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][2], b0_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][2], b0_, d1_, 0, 0, 0);                \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][0], b2_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][0], b2_, d1_, 0, 0, 0);                \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][1], b1_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][1], b1_, d1_, 0, 0, 0);                \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][1], b0_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][1], b0_, d1_, 0, 0, 0);                \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][0], b1_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][0], b1_, d1_, 0, 0, 0);                \
            d0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0][0], b0_, d0_, 0, 0, 0);                \
            d1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1][0], b0_, d1_, 0, 0, 0);                \
        }                                                                                              \
    } while (0)

    // issue order of one chunk: the 18 fragment reads lead (reads of the next column tile one tile ahead), every MFMA is
    // followed by two of the ~90 VALU operations of the split and every fourth by one of the 12 LDS stores; the global loads
    // of chunk t + 2 go last, right after the registers they fill were drained
    // issue order of one chunk: the 4 row loads of chunk t + 2 first (a whole chunk to land), the 9 fragment reads of the first
    // column tile, then every MFMA is followed by two of the ~100 VALU operations of the split, every fourth by one of the 12
    // LDS stores, and the fragment reads of the next column tile run one tile ahead
#define AG_WG_GROUP(reads_)                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, reads_, 0);                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0)
#define AG_WG_ORDER()                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);                                             \
        AG_WG_GROUP(3); AG_WG_GROUP(0); AG_WG_GROUP(0);                                                \
        AG_WG_GROUP(3); AG_WG_GROUP(0); AG_WG_GROUP(0);                                                \
        AG_WG_GROUP(3); AG_WG_GROUP(0); AG_WG_GROUP(0);                                                \
        AG_WG_GROUP(0); AG_WG_GROUP(0); AG_WG_GROUP(0);                                                \
    } while (0)

#ifdef AG_SPLIT_SETPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // static priority for the later-dispatched half (two waves per SIMD)
#endif
    // ---- prologue: chunks 0 and 1 requested, chunk 0 into stage 0
    if (n > 0) AG_WG_LOAD(0, va, nva);
    if (n > 1) AG_WG_LOAD(1, vb, nvb);
    if (n > 0) AG_WG_WRITE(0, 1, va, nva);
    __syncthreads();

    // ---- main loop, two chunks per trip (so that the register set of a chunk is a compile-time choice): at trip t the loads of
    //      chunk t + 2 are issued into the set chunk t was staged from, stage t & 1 is multiplied and chunk t + 1 (requested a
    //      whole chunk ago) is split into the other stage.  Branch-free bodies (one basic block each: the issue order needs
    //      that); only the last chunk of the whole matrix can be ragged, and it is staged by a peeled trip behind the loop.
    int t = 0;
#pragma unroll 1
    for (; t + 3 < n; t += 2) {
        AG_WG_LOAD(t + 2, va, nva);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay at the top: sunk to the end of the chunk (where LLVM puts them
        AG_WG_COMPUTE(0);                       // to shorten live ranges) the next chunk opens waiting for HBM
        AG_WG_WRITE(1, 0, vb, nvb);
        if (ORDERED) AG_WG_ORDER();
        __syncthreads();
        AG_WG_LOAD(t + 3, vb, nvb);
        __builtin_amdgcn_sched_barrier(0);
        AG_WG_COMPUTE(1);
        AG_WG_WRITE(0, 0, va, nva);
        if (ORDERED) AG_WG_ORDER();
        __syncthreads();
    }
    // peeled trips (t even here): at most three chunks are left; rows past M are zeroed where a chunk is staged
#pragma unroll 1
    for (; t < n; t += 2) {
        if (t + 2 < n) AG_WG_LOAD(t + 2, va, nva);
        AG_WG_COMPUTE(0);
        if (t + 1 < n) AG_WG_WRITE(1, 1, vb, nvb);
        __syncthreads();
        if (t + 1 < n) {
            AG_WG_COMPUTE(1);
            if (t + 2 < n) AG_WG_WRITE(0, 1, va, nva);
            __syncthreads();
        }
    }
#undef AG_WG_ORDER
#undef AG_WG_GROUP
#undef AG_WG_COMPUTE
#undef AG_WG_WRITE
#undef AG_WG_HALF
#undef AG_WG_LOAD

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // A-image unit u holds logical row co = 4 (u & 63) + (u >> 6); column tile j holds logical columns 4 (32 wn + l31) + j.
    float* __restrict__ out = partials + (size_t)blockIdx.x * WN * WN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int co = 4 * (u & 63) + (u >> 6);
            const float4 o = make_float4(acc[i * 4 + 0][r], acc[i * 4 + 1][r], acc[i * 4 + 2][r], acc[i * 4 + 3][r]);
            reinterpret_cast<float4*>(out + (size_t)co * WN)[wn * 32 + l31] = o;
        }
    }
}

int device_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

}  // namespace

static int g_wgrad_ordered = 1;
#ifdef AG_EXPERIMENTS
extern "C" int ag_debug_split_wgrad_ordered(int on) { g_wgrad_ordered = on ? 1 : 0; return AG_OK; }
#endif

// one workgroup per CU (each owns the whole 256 x 256 output for its rows), never more slices than 16-row chunks
extern "C" int ag_split_wgrad_slices(int M) {
    if (M <= 0) return 0;
    const int chunks = (M + WBK - 1) / WBK;
    const int s = device_cus();
    return chunks < s ? chunks : s;
}

extern "C" int ag_split_wgrad(const float* dZ_dev, const float* X_dev, float* partials_dev, int M, int n, int k, int slices,
                              void* stream) {
    if (!dZ_dev || !X_dev || !partials_dev || M <= 0 || slices <= 0) return AG_ERR_INVALID_ARG;
    if (n != WN || k != WN) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)dZ_dev | (uintptr_t)X_dev | (uintptr_t)partials_dev) & 15) return AG_ERR_INVALID_ARG;
    static bool attr_set[64] = {};      // per device ordinal: the dynamic-LDS limit is an attribute of (function, device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AG_ERR_HIP;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(split_wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kWgradLds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(split_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kWgradLds) != hipSuccess)
            return AG_ERR_HIP;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int chunks = (M + WBK - 1) / WBK;
    const int cps = (chunks + slices - 1) / slices;
    if (g_wgrad_ordered)
        hipLaunchKernelGGL(split_wgrad_kernel<true>, dim3(slices), dim3(512), kWgradLds, (hipStream_t)stream, dZ_dev, X_dev,
                           partials_dev, M, cps);
    else
        hipLaunchKernelGGL(split_wgrad_kernel<false>, dim3(slices), dim3(512), kWgradLds, (hipStream_t)stream, dZ_dev, X_dev,
                           partials_dev, M, cps);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
