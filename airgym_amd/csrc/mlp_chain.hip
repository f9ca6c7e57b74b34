// mlp_chain.hip - the actor-critic MLP [D -> 256 -> 256 -> (A + 1)] of the PPO loop as ONE kernel whose activations never leave
// the registers (gfx950 / MI355X).  Forward: ModelA2CContinuousLogStd.forward / MLP (lib/model/a2c_continuous_logstd_model.py:
// 80-193, lib/network/mlp.py:36-39) for the shared-trunk ELU network with a fixed sigma: input normaliser -> Linear + ELU ->
// Linear + ELU -> mu | value heads.  Replaces, for the rollout, the launch pair ag_mlp_input_layer + ag_split_gemm_elu_heads
// (the [M, 256] activations h1 and z2 went through HBM between and behind them: 3 x 67 MB per rollout step at M = 65 536).
//
// Arithmetic: exactly that of split_gemm.hip - every float32 operand split into three bf16 pieces, six of the nine cross
// products on v_mfma_f32_32x32x16_bf16 with float32 accumulation, smallest first (float32-accurate) - for ALL three products
// here, the 18-wide first layer and the heads included.
//
// Layout ("lane = batch row"; the index algebra is stated and checked in tests/test_chain_layout_model.py):
//   * a wavefront owns 32 batch rows and computes the TRANSPOSED products  H^T [feature, row] = W [feature, k] . X^T [k, row]:
//     weights are the MFMA's A operand (M = output feature), activations its B operand (N = batch row).  In the C / D layout
//     a lane then holds ONE batch row (n = lane & 31) and, in registers r of output tile t, the features
//     32 t + (r & 3) + 8 (r >> 2) + 4 (lane >> 5): registers 8 s .. 8 s + 7 ARE the next product's B fragment for K step
//     2 t + s - in the permuted K order  k = 16 c + perm(h, i),  perm(h, i) = (i & 3) + 8 (i >> 2) + 4 h  - so they are split
//     into bf16 pieces where they are and fed back to the matrix cores: no LDS round trip, no transposition, ever.
//   * the weights are prepared once per policy version (ag_mlp_chain_prepare) into images in exactly that K order:
//       stream image = [KP1 / 16 blocks of W1 | b1 (natural K order; the bias is column D against an all-ones input column)]
//                      [16 blocks of W2 (chain K order)],  one block = one K step = [plane 3][k-half 2][feature 256] x 16 B = 24 KB
//       head image   = 16 blocks of [plane 3][k-half 2][head 32 (A + 1 used, rest zero)] x 16 B = 48 KB, resident in LDS
//     A block is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), one block
//     ahead of the one being multiplied, two stages; one workgroup barrier per K step.
//   * 256 threads = 4 waves = 128 rows per workgroup iteration, ONE wave per SIMD (the three accumulator sets need ~350 of the
//     512 registers); persistent workgroups, one per CU, stride over the row tiles; the weight stream runs on across tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "split_common.hpp"

namespace {

constexpr int CF = 256;                       // layer width
constexpr int CBLK = 3 * 2 * CF;              // 16-byte units of one streamed weight block (24 KB)
constexpr int CWH = 16 * 3 * 2 * 32;          // units of the resident head image (48 KB)
constexpr int CROWS = 128;                    // batch rows per workgroup iteration (4 waves x 32)

typedef __attribute__((address_space(1))) const void* cg_ptr;
typedef __attribute__((address_space(3))) void* cl_ptr;

__device__ __forceinline__ int chain_perm(int h, int i) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// LDS-DMA of one 1 KiB piece: lane l's 16 bytes at `src` (per-lane global address) land at LDS byte address lds_base + 16 l
// (lds_base wave-uniform, in M0).  Written as inline assembly on purpose: hipcc treats the builtin's destination as "may alias
// any LDS read" and puts an `s_waitcnt vmcnt(0)` in front of the first ds_read behind it - the copy of the NEXT block would be
// waited for at the top of every K step.  Hidden from the compiler, the copies are drained by the explicit vmcnt(0) in front
// of the K step's barrier instead (chain_dma_wait), a whole K step after they were issued.
__device__ __forceinline__ void chain_dma16(const uint4* src, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void chain_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
__device__ __forceinline__ uint32_t chain_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(cl_ptr)(p);
}

__device__ __forceinline__ float chain_elu(float z) {      // same form as split_gemm.hip sg_elu / ppo_kernels.hip elu1
    return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f;
}

// ---- weight images -------------------------------------------------------------------------------------------------------
// one thread = one (block, k-half, row) unit triple: 8 K elements of one weight row, split, to the three planes
__global__ __launch_bounds__(256) void chain_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D, int nb1,
                                                            const float* __restrict__ W2, const float* __restrict__ Wh, int A1,
                                                            uint4* __restrict__ stream_img, uint4* __restrict__ wh_img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n_stream = (nb1 + 16) * 2 * CF;
    const int n_wh = 16 * 2 * 32;
    float v[8];
    uint4* dst;
    int stride;
    if (idx < n_stream) {
        const int m = idx % CF, h = (idx / CF) & 1, blk = idx / (2 * CF);
        if (blk < nb1) {                      // first layer, natural K order, bias in column D
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = 16 * blk + 8 * h + i;
                v[i] = kk < D ? W1[(size_t)m * D + kk] : (kk == D ? b1[m] : 0.0f);
            }
        } else {                              // second layer, chain K order
            const int c = blk - nb1;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = W2[(size_t)m * CF + 16 * c + chain_perm(h, i)];
        }
        dst = stream_img + (size_t)blk * CBLK + h * CF + m;
        stride = 2 * CF;
    } else if (idx < n_stream + n_wh) {
        const int j = idx - n_stream;
        const int m = j % 32, h = (j / 32) & 1, c = j / 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = m < A1 ? Wh[(size_t)m * CF + 16 * c + chain_perm(h, i)] : 0.0f;
        dst = wh_img + (size_t)c * (3 * 2 * 32) + h * 32 + m;
        stride = 2 * 32;
    } else {
        return;
    }
    uint4 p1, p2, p3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
    dst[0] = p1;
    dst[stride] = p2;
    dst[2 * stride] = p3;
}

struct ChainFwdArgs {
    const float* obs;          // [M, D] raw observations
    const double* mean;        // [D] or null (no input normaliser)
    const double* var;         // [D]
    float eps, clip;
    const uint4* stream_img;   // [(KP1 / 16 + 16) blocks][CBLK]
    const uint4* wh_img;       // [CWH]
    const float* b2;           // [256]
    const float* bh;           // [A1]
    float* heads;              // [M, A1]
    float* xn;                 // [M, D] or null: the normalised inputs
    float* h1;                 // [M, 256] or null: first-layer activations
    float* h2;                 // [M, 256] or null: second-layer activations
    int M, D;
    int debug_skip;            // experiments build only: bit0 no weight DMA behind the first block, bit1 no MFMA, bit2 no barriers
};
// (a compile-time constant inside the kernel: a run-time test in front of every MFMA group would change the schedule it measures)
#define AG_CHAIN_DBG(bit_) (DBG & (bit_))

#define AG_CHAIN_MFMA6(acc_, a_, b_)                                                                   \
    do {                                                                                               \
        if (AG_CHAIN_DBG(2)) { acc_[0] += __builtin_bit_cast(float, (int)a_[0][0] ^ (int)b_[0][0] ^ (int)a_[1][1] ^ (int)b_[1][1] ^ (int)a_[2][2] ^ (int)b_[2][2]); break; } \
        if (kPlanes == 3) {                                                                            \
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], b_[0], acc_, 0, 0, 0);               \
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[2], acc_, 0, 0, 0);               \
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[1], acc_, 0, 0, 0);               \
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[0], acc_, 0, 0, 0);               \
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[1], acc_, 0, 0, 0);               \
        }                                                                                              \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[0], acc_, 0, 0, 0);                   \
    } while (0)

// the same six products for TWO accumulator tiles that share the B fragment, interleaved so that consecutive MFMAs never wait
// for each other's result (a dependent pair issued back to back is free, but any other instruction between them costs ~43
// cycles - MI355X_MICROARCH.md - and the compiler does put fillers there)
#define AG_CHAIN_MFMA6x2(acc0_, acc1_, a0_, a1_, b_)                                                   \
    do {                                                                                               \
        if (AG_CHAIN_DBG(2)) {                                                                         \
            acc0_[0] += __builtin_bit_cast(float, (int)a0_[0][0] ^ (int)b_[0][0] ^ (int)a0_[1][1] ^ (int)b_[1][1] ^ (int)a0_[2][2] ^ (int)b_[2][2]); \
            acc1_[0] += __builtin_bit_cast(float, (int)a1_[0][0] ^ (int)a1_[1][1] ^ (int)a1_[2][2]);   \
            break;                                                                                     \
        }                                                                                              \
        if (kPlanes == 3) {                                                                            \
            acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[2], b_[0], acc0_, 0, 0, 0);            \
            acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[2], b_[0], acc1_, 0, 0, 0);            \
            acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[0], b_[2], acc0_, 0, 0, 0);            \
            acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[0], b_[2], acc1_, 0, 0, 0);            \
            acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[1], b_[1], acc0_, 0, 0, 0);            \
            acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[1], b_[1], acc1_, 0, 0, 0);            \
            acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[1], b_[0], acc0_, 0, 0, 0);            \
            acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[1], b_[0], acc1_, 0, 0, 0);            \
            acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[0], b_[1], acc0_, 0, 0, 0);            \
            acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[0], b_[1], acc1_, 0, 0, 0);            \
        }                                                                                              \
        acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_[0], b_[0], acc0_, 0, 0, 0);                \
        acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_[0], b_[0], acc1_, 0, 0, 0);                \
    } while (0)

// registers 8 s .. 8 s + 7 of an accumulator tile -> the three bf16x8 pieces of a B (or A) fragment
__device__ __forceinline__ void chain_split_regs(const f32x16& acc, int s, bf16x8 (&out)[3]) {
    uint4 p1, p2, p3;
    if (s == 0)
        split8(make_float4(acc[0], acc[1], acc[2], acc[3]), make_float4(acc[4], acc[5], acc[6], acc[7]), p1, p2, p3);
    else
        split8(make_float4(acc[8], acc[9], acc[10], acc[11]), make_float4(acc[12], acc[13], acc[14], acc[15]), p1, p2, p3);
    out[0] = *reinterpret_cast<const bf16x8*>(&p1);
    out[1] = *reinterpret_cast<const bf16x8*>(&p2);
    out[2] = *reinterpret_cast<const bf16x8*>(&p3);
}

template <int KP1, int A1, bool STORE, int DBG>
__global__ __launch_bounds__(256, 1) void mlp_chain_fwd_kernel(const ChainFwdArgs a) {
    constexpr int NB1 = KP1 / 16, NB = NB1 + 16;
    static_assert((NB & 1) == 0, "an even number of blocks per tile keeps the stage parity across tiles");
    extern __shared__ uint4 lds[];
    uint4* stage = lds;                                        // [2][CBLK]
    uint4* whres = lds + 2 * CBLK;                             // [CWH]
    float* fconst = reinterpret_cast<float*>(whres + CWH);     // b2 [256] | mean [64] | sd [64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int ntiles = (a.M + CROWS - 1) / CROWS;
    const bool normalize = a.mean != nullptr;

    // The weight stream: block after block of the stream image, wrapping at the end of a tile - a RUNNING scalar pointer (were
    // the source address a function of the unrolled K-step index, LLVM would hoist all 6 x NB per-lane 64-bit addresses out of
    // the tile loop: 216 registers, spilled).  One call issues the next block into stage `stg_`.
    const uint4* dma_src = a.stream_img;
    int dma_blk = 0;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    int dma_left = my_tiles * NB;                      // blocks this workgroup still has to request
    const int lane_unit = wave * 64 + lane;
    const uint32_t stage_addr = __builtin_amdgcn_readfirstlane(chain_lds_addr(stage));
#define AG_CHAIN_ISSUE_NEXT(stg_)                                                                      \
    do {                                                                                               \
        if (dma_left > 0 && !AG_CHAIN_DBG(1)) {                                                        \
            const uint4* src_ = dma_src;                                                               \
            const uint32_t dst_ = stage_addr + ((stg_) * CBLK + wave * 64) * 16;                       \
            _Pragma("unroll") for (int it = 0; it < CBLK / 256; ++it)                                  \
                chain_dma16(src_ + it * 256 + lane_unit, dst_ + it * 256 * 16);                        \
        }                                                                                              \
        --dma_left;                                                                                    \
        dma_src += CBLK;                                                                               \
        if (++dma_blk == NB) { dma_blk = 0; dma_src = a.stream_img; }                                  \
    } while (0)
    // A fragments (three planes) of output tiles 2 p and 2 p + 1 out of a landed block
#define AG_CHAIN_READ_PAIR(w0_, w1_, st_, pair_)                                                       \
    do {                                                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                \
            const uint4 u0_ = (st_)[(p * 2 + h) * CF + 64 * (pair_) + l31];                            \
            const uint4 u1_ = (st_)[(p * 2 + h) * CF + 64 * (pair_) + 32 + l31];                       \
            w0_[p] = *reinterpret_cast<const bf16x8*>(&u0_);                                           \
            w1_[p] = *reinterpret_cast<const bf16x8*>(&u1_);                                           \
        }                                                                                              \
    } while (0)
    // issue order of one tile pair's twelve MFMAs: every MFMA is followed by up to three vector-ALU instructions (the next K
    // step's ELU / split work), the fragment reads of the next pair lead
#define AG_CHAIN_ORDER_PAIR()                                                                          \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                         \
        }                                                                                              \
    } while (0)

    // ---- prologue: resident head image, constants, the first two blocks
    {
        const uint32_t wh_addr = __builtin_amdgcn_readfirstlane(chain_lds_addr(whres)) + wave * 64 * 16;
#pragma unroll
        for (int it = 0; it < CWH / 256; ++it) chain_dma16(a.wh_img + it * 256 + lane_unit, wh_addr + it * 256 * 16);
    }
    fconst[tid] = a.b2[tid];
    if (tid < 64) {
        const bool in = normalize && tid < a.D;
        fconst[256 + tid] = in ? (float)a.mean[tid] : 0.0f;
        fconst[320 + tid] = in ? sqrtf((float)a.var[tid] + a.eps) : 1.0f;
    }
    AG_CHAIN_ISSUE_NEXT(0);
    AG_CHAIN_ISSUE_NEXT(1);
    chain_dma_wait();
    __syncthreads();              // head image, constants, blocks 0 and 1 are in LDS for every wave
    // A fragments of the pair that is multiplied next; read one pair ahead, across K steps and tiles
    bf16x8 wc0[3], wc1[3], wn0[3], wn1[3];
    AG_CHAIN_READ_PAIR(wc0, wc1, stage, 0);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row_raw = tile * CROWS + wave * 32 + l31;
        const bool row_ok = row_raw < a.M;
        const int row = row_ok ? row_raw : a.M - 1;            // rows past M are computed on a copy of the last row, never stored
        const bool next_tile = tile + (int)gridDim.x < ntiles;

        // this lane's input row, k = 16 g + 8 h + i: all loads issued together; columns past D: the all-ones bias column at D,
        // zeros behind it
        float xv[NB1][8];
        const float* xrow = a.obs + (size_t)row * a.D;
#ifdef AG_CHAIN_SCALAR_LOADS                   // (A/B switch: the round-4 form, one dword load per column)
        const bool vec2 = false;
#else
        const bool vec2 = (a.D & 1) == 0;      // even width: rows are 8-byte aligned, a lane's 8 columns of a K step are four 8-byte loads
#endif
#pragma unroll
        for (int g = 0; g < NB1; ++g) {
            const int k0 = 16 * g + 8 * h;
            if (vec2 && k0 + 8 <= a.D) {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const float2 v2 = reinterpret_cast<const float2*>(xrow + k0)[i2];
                    xv[g][2 * i2] = v2.x;
                    xv[g][2 * i2 + 1] = v2.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[g][i] = xrow[min(k0 + i, a.D - 1)];
            }
        }
#pragma unroll
        for (int g = 0; g < NB1; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = 16 * g + 8 * h + i;
                float v = xv[g][i];
                if (normalize) {
                    const int kc = min(kk, 63);
                    v = (v - fconst[256 + kc]) / fconst[320 + kc];
                    v = fminf(fmaxf(v, -a.clip), a.clip);
                    if (STORE && a.xn != nullptr && row_ok && kk < a.D) a.xn[(size_t)row * a.D + kk] = v;
                }
                xv[g][i] = kk < a.D ? v : (kk == a.D ? 1.0f : 0.0f);
            }

        f32x16 acc1[8], acc2[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[t][r] = 0.0f; acc2[t][r] = 0.0f; }

        // The B fragment of K step g + 1 is formed DURING K step g (its inputs - the input row, or first-layer registers that
        // were final long ago - do not depend on this step's products), a QUARTER of it under each tile pair's MFMAs.
        bf16x8 bcur[3];
        uint4 qn[3];                       // the next step's fragment under construction (planes 1..3)
        {
            uint4 q1, q2, q3;
            split8(make_float4(xv[0][0], xv[0][1], xv[0][2], xv[0][3]), make_float4(xv[0][4], xv[0][5], xv[0][6], xv[0][7]), q1, q2, q3);
            bcur[0] = *reinterpret_cast<const bf16x8*>(&q1);
            bcur[1] = *reinterpret_cast<const bf16x8*>(&q2);
            bcur[2] = *reinterpret_cast<const bf16x8*>(&q3);
        }
        // quarter `q_` of the work that turns first-layer tile `tn_`'s registers into the B fragment of K step (tn_, sn_):
        //   sn_ = 0: ELU of all 16 registers (4 per quarter; registers 8..15 are used by the FOLLOWING step), the fragment's low
        //            half in quarter 2, its high half in quarter 3;   sn_ = 1: the two halves in quarters 0 and 1
#define AG_CHAIN_FRAG_QUARTER(tn_, sn_, q_)                                                            \
        do {                                                                                           \
            if ((sn_) == 0) {                                                                          \
                _Pragma("unroll") for (int r = 4 * (q_); r < 4 * (q_) + 4; ++r) acc1[tn_][r] = chain_elu(acc1[tn_][r]); \
                if (STORE && a.h1 != nullptr && row_ok)                                                \
                    *reinterpret_cast<float4*>(a.h1 + (size_t)row * CF + 32 * (tn_) + 8 * (q_) + 4 * h) = \
                        make_float4(acc1[tn_][4 * (q_)], acc1[tn_][4 * (q_) + 1], acc1[tn_][4 * (q_) + 2], acc1[tn_][4 * (q_) + 3]); \
                if ((q_) == 2) {                                                                       \
                    split_pair(acc1[tn_][0], acc1[tn_][1], qn[0].x, qn[1].x, qn[2].x);                 \
                    split_pair(acc1[tn_][2], acc1[tn_][3], qn[0].y, qn[1].y, qn[2].y);                 \
                } else if ((q_) == 3) {                                                                \
                    split_pair(acc1[tn_][4], acc1[tn_][5], qn[0].z, qn[1].z, qn[2].z);                 \
                    split_pair(acc1[tn_][6], acc1[tn_][7], qn[0].w, qn[1].w, qn[2].w);                 \
                }                                                                                      \
            } else if ((q_) == 0) {                                                                    \
                split_pair(acc1[tn_][8], acc1[tn_][9], qn[0].x, qn[1].x, qn[2].x);                     \
                split_pair(acc1[tn_][10], acc1[tn_][11], qn[0].y, qn[1].y, qn[2].y);                   \
            } else if ((q_) == 1) {                                                                    \
                split_pair(acc1[tn_][12], acc1[tn_][13], qn[0].z, qn[1].z, qn[2].z);                   \
                split_pair(acc1[tn_][14], acc1[tn_][15], qn[0].w, qn[1].w, qn[2].w);                   \
            }                                                                                          \
        } while (0)
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            // K step g multiplies block g (stage g & 1), whose first pair of fragments is already in wc0 / wc1.  The step's ONE
            // barrier stands in front of its LAST tile pair: by then this wave has read all of block g, so after the barrier the
            // stage is free for block g + 2 (requested at once: a whole K step to land) and block g + 1 (requested a K step ago,
            // drained by chain_dma_wait) can be read - its first fragments arrive under the last pair's MFMAs.
            const uint4* st = stage + (g & 1) * CBLK;
            const uint4* stn = stage + ((g + 1) & 1) * CBLK;
            if (g + 1 < NB1) {              // first layer: the next K step's fragment is the next 16 input columns
                split8(make_float4(xv[g + 1][0], xv[g + 1][1], xv[g + 1][2], xv[g + 1][3]),
                       make_float4(xv[g + 1][4], xv[g + 1][5], xv[g + 1][6], xv[g + 1][7]), qn[0], qn[1], qn[2]);
            }
            // ---- this step's products: weights = A operand (two output tiles at a time), bcur = B operand.  Every tile pair is
            //      its own scheduling region (sched_barrier): fragment reads of the NEXT pair first, then twelve MFMAs with the
            //      quarter of vector-ALU work spread between them.
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                if (jp < 3) {
                    if (!AG_CHAIN_DBG(16)) AG_CHAIN_READ_PAIR(wn0, wn1, st, jp + 1);
                } else if (g + 1 < NB || next_tile) {
                    chain_dma_wait();
                    if (!AG_CHAIN_DBG(4)) __syncthreads();
                    AG_CHAIN_ISSUE_NEXT(g & 1);
                    if (!AG_CHAIN_DBG(16)) AG_CHAIN_READ_PAIR(wn0, wn1, stn, 0);
                }
                if (g + 1 < NB && g + 1 > NB1 && !AG_CHAIN_DBG(8)) {
                    const int cn = g + 1 - NB1;      // second layer, K step cn >= 1: first-layer tile cn >> 1, half cn & 1
                    AG_CHAIN_FRAG_QUARTER(cn >> 1, cn & 1, jp);
                }
                if (g < NB1) AG_CHAIN_MFMA6x2(acc1[2 * jp], acc1[2 * jp + 1], wc0, wc1, bcur);
                else AG_CHAIN_MFMA6x2(acc2[2 * jp], acc2[2 * jp + 1], wc0, wc1, bcur);
                AG_CHAIN_ORDER_PAIR();
                __builtin_amdgcn_sched_barrier(0);
                if (!AG_CHAIN_DBG(16)) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) { wc0[p] = wn0[p]; wc1[p] = wn1[p]; }
                }
            }
            if (g + 1 == NB1) {
                // the first layer has just ended: the first B fragment of the second layer (tile 0, registers 0..7) - the only
                // one that cannot be formed a step ahead
                AG_CHAIN_FRAG_QUARTER(0, 0, 0); AG_CHAIN_FRAG_QUARTER(0, 0, 1); AG_CHAIN_FRAG_QUARTER(0, 0, 2); AG_CHAIN_FRAG_QUARTER(0, 0, 3);
            }
            if (g + 1 < NB) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bcur[p] = *reinterpret_cast<const bf16x8*>(&qn[p]);
            }
        }
#undef AG_CHAIN_FRAG_QUARTER

        // ---- bias + ELU of the second layer (its registers hold features 32 j + (r & 3) + 8 (r >> 2) + 4 h), then the heads:
        //      one more transposed product against the resident head image, B fragments again straight from the registers.
        //      Two accumulators (even / odd K steps), fragment c + 1 formed while step c multiplies.
#define AG_CHAIN_H2_HALF(t_, s_)                                                                       \
        do {                                                                                           \
            _Pragma("unroll") for (int r = 8 * (s_); r < 8 * (s_) + 8; r += 4) {                       \
                const float4 bb = *reinterpret_cast<const float4*>(fconst + 32 * (t_) + 8 * (r >> 2) + 4 * h); \
                acc2[t_][r] = chain_elu(acc2[t_][r] + bb.x);                                           \
                acc2[t_][r + 1] = chain_elu(acc2[t_][r + 1] + bb.y);                                   \
                acc2[t_][r + 2] = chain_elu(acc2[t_][r + 2] + bb.z);                                   \
                acc2[t_][r + 3] = chain_elu(acc2[t_][r + 3] + bb.w);                                   \
                if (STORE && a.h2 != nullptr && row_ok)                                                \
                    *reinterpret_cast<float4*>(a.h2 + (size_t)row * CF + 32 * (t_) + 8 * (r >> 2) + 4 * h) = \
                        make_float4(acc2[t_][r], acc2[t_][r + 1], acc2[t_][r + 2], acc2[t_][r + 3]);   \
            }                                                                                          \
        } while (0)
        f32x16 acch, acch1;
        bf16x8 bnext[3];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acch[r] = 0.0f; acch1[r] = 0.0f; }
        AG_CHAIN_H2_HALF(0, 0);
        chain_split_regs(acc2[0], 0, bcur);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c + 1 < 16) {
                const int tn = (c + 1) >> 1, sn = (c + 1) & 1;
                if (sn == 0) AG_CHAIN_H2_HALF(tn, 0); else AG_CHAIN_H2_HALF(tn, 1);
                chain_split_regs(acc2[tn], sn, bnext);
            }
            bf16x8 w[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 u = whres[c * (3 * 2 * 32) + (p * 2 + h) * 32 + l31];
                w[p] = *reinterpret_cast<const bf16x8*>(&u);
            }
            if (c & 1) AG_CHAIN_MFMA6(acch1, w, bcur); else AG_CHAIN_MFMA6(acch, w, bcur);
            if (c + 1 < 16) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bcur[p] = bnext[p];
            }
        }
#undef AG_CHAIN_H2_HALF
#pragma unroll
        for (int r = 0; r < 4; ++r) acch[r] += acch1[r];
        // heads of batch row n: head index (r & 3) + 8 (r >> 2) + 4 h -> lane n holds heads 0..3 (registers 0..3), lane n + 32
        // heads 4..7 (registers 0..3 again)
        if (row_ok) {
            float* out = a.heads + (size_t)row * A1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hd = 4 * h + r;
                if (hd < A1) out[hd] = acch[r] + a.bh[hd];
            }
        }
    }
#undef AG_CHAIN_ISSUE_NEXT
#undef AG_CHAIN_READ_PAIR
#undef AG_CHAIN_ORDER_PAIR
}

constexpr size_t chain_fwd_lds_bytes() { return (size_t)(2 * CBLK + CWH) * 16 + (256 + 64 + 64) * 4; }

int chain_kp1(int D) { return D + 1 <= 32 ? 32 : (D + 1 <= 64 ? 64 : 0); }

int chain_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

template <int KP1, int A1, bool STORE, int DBG = 0>
int launch_chain_fwd(const ChainFwdArgs& a, void* stream) {
    static bool attr_set[64] = {};
    auto* fn = mlp_chain_fwd_kernel<KP1, A1, STORE, DBG>;
    constexpr size_t lds_bytes = chain_fwd_lds_bytes();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AG_ERR_HIP;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AG_ERR_HIP;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int ntiles = (a.M + CROWS - 1) / CROWS;
    const int grid = ntiles < chain_cus() ? ntiles : chain_cus();
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

}  // namespace

#ifdef AG_EXPERIMENTS
static int g_chain_debug_skip = 0;
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_debug_chain_skip(int mask) { g_chain_debug_skip = mask; return AG_OK; }
#endif
#else
constexpr int g_chain_debug_skip = 0;
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_mlp_chain_supported(int D, int C, int A1) {
    return (C == CF && chain_kp1(D) != 0 && (A1 == 5 || A1 == 6)) ? 1 : 0;
}
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" long long ag_mlp_chain_image_bytes(int D) {
    const int kp1 = chain_kp1(D);
    if (kp1 == 0) return 0;
    return ((long long)(kp1 / 16 + 16) * CBLK + CWH) * 16;
}
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_mlp_chain_prepare(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, const float* Wh_dev, int A1,
                                    void* image_dev, void* stream) {
    if (!W1_dev || !b1_dev || !W2_dev || !Wh_dev || !image_dev) return AG_ERR_INVALID_ARG;
    if (!ag_mlp_chain_supported(D, CF, A1)) return AG_ERR_UNSUPPORTED;
    if ((uintptr_t)image_dev & 15) return AG_ERR_INVALID_ARG;
    const int nb1 = chain_kp1(D) / 16;
    uint4* stream_img = (uint4*)image_dev;
    uint4* wh_img = stream_img + (size_t)(nb1 + 16) * CBLK;
    const int threads = (nb1 + 16) * 2 * CF + 16 * 2 * 32;
    hipLaunchKernelGGL(chain_prepare_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1_dev, b1_dev, D, nb1,
                       W2_dev, Wh_dev, A1, stream_img, wh_img);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
#endif

extern "C" int AG_PREC(ag_mlp_chain_forward)(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
                                    const void* image_dev, const float* b2_dev, const float* bh_dev, float* heads_dev, float* xn_dev,
                                    float* h1_dev, float* h2_dev, int M, int D, int A1, void* stream) {
    if (!obs_dev || !image_dev || !b2_dev || !bh_dev || !heads_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if ((mean_dev == nullptr) != (var_dev == nullptr)) return AG_ERR_INVALID_ARG;
    if (!ag_mlp_chain_supported(D, CF, A1)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)image_dev & 15) || (h1_dev && ((uintptr_t)h1_dev & 15)) || (h2_dev && ((uintptr_t)h2_dev & 15))) return AG_ERR_INVALID_ARG;
    const int nb1 = chain_kp1(D) / 16;
    ChainFwdArgs a;
    a.obs = obs_dev; a.mean = mean_dev; a.var = var_dev; a.eps = eps; a.clip = clip;
    a.stream_img = (const uint4*)image_dev;
    a.wh_img = a.stream_img + (size_t)(nb1 + 16) * CBLK;
    a.b2 = b2_dev; a.bh = bh_dev; a.heads = heads_dev; a.xn = xn_dev; a.h1 = h1_dev; a.h2 = h2_dev; a.M = M; a.D = D;
    a.debug_skip = g_chain_debug_skip;
    const bool store = xn_dev || h1_dev || h2_dev;
#if defined(AG_EXPERIMENTS) && defined(AG_CHAIN_ABLATIONS)      // (nine more instantiations: ~15 min of compile time; opt-in)
    if (g_chain_debug_skip != 0 && nb1 == 2 && A1 == 5 && !store) {
        switch (g_chain_debug_skip) {
            case 1: return launch_chain_fwd<32, 5, false, 1>(a, stream);
            case 2: return launch_chain_fwd<32, 5, false, 2>(a, stream);
            case 3: return launch_chain_fwd<32, 5, false, 3>(a, stream);
            case 4: return launch_chain_fwd<32, 5, false, 4>(a, stream);
            case 7: return launch_chain_fwd<32, 5, false, 7>(a, stream);
            case 8: return launch_chain_fwd<32, 5, false, 8>(a, stream);
            case 16: return launch_chain_fwd<32, 5, false, 16>(a, stream);
            case 24: return launch_chain_fwd<32, 5, false, 24>(a, stream);
            case 29: return launch_chain_fwd<32, 5, false, 29>(a, stream);
            default: return AG_ERR_INVALID_ARG;
        }
    }
#endif
#define AG_CHAIN_GO(KP, AV) (store ? launch_chain_fwd<KP, AV, true>(a, stream) : launch_chain_fwd<KP, AV, false>(a, stream))
    if (nb1 == 2) return A1 == 5 ? AG_CHAIN_GO(32, 5) : AG_CHAIN_GO(32, 6);
    return A1 == 5 ? AG_CHAIN_GO(64, 5) : AG_CHAIN_GO(64, 6);
#undef AG_CHAIN_GO
}
