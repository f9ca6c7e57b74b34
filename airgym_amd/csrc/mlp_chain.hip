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
};

#define AG_CHAIN_MFMA6(acc_, a_, b_)                                                                   \
    do {                                                                                               \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], b_[0], acc_, 0, 0, 0);                   \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[2], acc_, 0, 0, 0);                   \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[1], acc_, 0, 0, 0);                   \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[0], acc_, 0, 0, 0);                   \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[1], acc_, 0, 0, 0);                   \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[0], acc_, 0, 0, 0);                   \
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

template <int KP1, int A1, bool STORE>
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
    const int lane_unit = wave * 64 + lane;
    const uint32_t stage_addr = __builtin_amdgcn_readfirstlane(chain_lds_addr(stage));
#define AG_CHAIN_ISSUE_NEXT(stg_)                                                                      \
    do {                                                                                               \
        const uint4* src_ = dma_src;                                                                   \
        const uint32_t dst_ = stage_addr + ((stg_) * CBLK + wave * 64) * 16;                           \
        _Pragma("unroll") for (int it = 0; it < CBLK / 256; ++it)                                      \
            chain_dma16(src_ + it * 256 + lane_unit, dst_ + it * 256 * 16);                            \
        dma_src += CBLK;                                                                               \
        if (++dma_blk == NB) { dma_blk = 0; dma_src = a.stream_img; }                                  \
    } while (0)

    // ---- prologue: resident head image, constants, block 0 of the first tile
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
    if ((int)blockIdx.x < ntiles) AG_CHAIN_ISSUE_NEXT(0);
    chain_dma_wait();
    __syncthreads();              // the head image and the constants are visible to every wave

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row_raw = tile * CROWS + wave * 32 + l31;
        const bool row_ok = row_raw < a.M;
        const int row = row_ok ? row_raw : a.M - 1;            // rows past M are computed on a copy of the last row, never stored
        const bool next_tile = tile + (int)gridDim.x < ntiles;

        // this lane's input row, k = 16 g + 8 h + i: all loads issued together (one wait, under the first K step's barrier);
        // columns past D: the all-ones bias column at D, zeros behind it
        float xv[NB1][8];
#pragma unroll
        for (int g = 0; g < NB1; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = 16 * g + 8 * h + i;
                xv[g][i] = a.obs[(size_t)row * a.D + min(kk, a.D - 1)];
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

#pragma unroll
        for (int g = 0; g < NB; ++g) {
            // block g has landed for every wave (each drains its own LDS-DMA pieces, issued a whole K step ago, before it
            // arrives); every wave is done with block g - 1, whose stage block g + 1 goes to
            chain_dma_wait();
            __syncthreads();
            if (g + 1 < NB || next_tile) AG_CHAIN_ISSUE_NEXT((g + 1) & 1);
            const uint4* st = stage + (g & 1) * CBLK;
            if (g < NB1) {
                // ---- first layer, K step g: B fragment = this lane's row of (normalised) inputs, k = 16 g + 8 h + i
                uint4 q1, q2, q3;
                split8(make_float4(xv[g][0], xv[g][1], xv[g][2], xv[g][3]), make_float4(xv[g][4], xv[g][5], xv[g][6], xv[g][7]), q1, q2, q3);
                bf16x8 b[3] = {*reinterpret_cast<const bf16x8*>(&q1), *reinterpret_cast<const bf16x8*>(&q2),
                               *reinterpret_cast<const bf16x8*>(&q3)};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    bf16x8 w[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint4 u = st[(p * 2 + h) * CF + 32 * t + l31];
                        w[p] = *reinterpret_cast<const bf16x8*>(&u);
                    }
                    AG_CHAIN_MFMA6(acc1[t], w, b);
                }
            } else {
                // ---- second layer, K step c: B fragment = registers 8 s .. 8 s + 7 of first-layer tile t, split where they are
                const int c = g - NB1, t = c >> 1, s = c & 1;
                if (s == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[t][r] = chain_elu(acc1[t][r]);
                    if (STORE && a.h1 != nullptr && row_ok) {
#pragma unroll
                        for (int r = 0; r < 16; r += 4)
                            *reinterpret_cast<float4*>(a.h1 + (size_t)row * CF + 32 * t + 8 * (r >> 2) + 4 * h) =
                                make_float4(acc1[t][r], acc1[t][r + 1], acc1[t][r + 2], acc1[t][r + 3]);
                    }
                }
                bf16x8 b[3];
                chain_split_regs(acc1[t], s, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bf16x8 w[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint4 u = st[(p * 2 + h) * CF + 32 * j + l31];
                        w[p] = *reinterpret_cast<const bf16x8*>(&u);
                    }
                    AG_CHAIN_MFMA6(acc2[j], w, b);
                }
            }
        }

        // ---- bias + ELU of the second layer (its registers hold features 32 j + (r & 3) + 8 (r >> 2) + 4 h), then the heads:
        //      one more transposed product against the resident head image, B fragments again straight from the registers
        f32x16 acch;
#pragma unroll
        for (int r = 0; r < 16; ++r) acch[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int t = c >> 1, s = c & 1;
            if (s == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float4 bb = *reinterpret_cast<const float4*>(fconst + 32 * t + 8 * (r >> 2) + 4 * h);
                    acc2[t][r] = chain_elu(acc2[t][r] + bb.x);
                    acc2[t][r + 1] = chain_elu(acc2[t][r + 1] + bb.y);
                    acc2[t][r + 2] = chain_elu(acc2[t][r + 2] + bb.z);
                    acc2[t][r + 3] = chain_elu(acc2[t][r + 3] + bb.w);
                }
                if (STORE && a.h2 != nullptr && row_ok) {
#pragma unroll
                    for (int r = 0; r < 16; r += 4)
                        *reinterpret_cast<float4*>(a.h2 + (size_t)row * CF + 32 * t + 8 * (r >> 2) + 4 * h) =
                            make_float4(acc2[t][r], acc2[t][r + 1], acc2[t][r + 2], acc2[t][r + 3]);
                }
            }
            bf16x8 b[3], w[3];
            chain_split_regs(acc2[t], s, b);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 u = whres[c * (3 * 2 * 32) + (p * 2 + h) * 32 + l31];
                w[p] = *reinterpret_cast<const bf16x8*>(&u);
            }
            AG_CHAIN_MFMA6(acch, w, b);
        }
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

template <int KP1, int A1, bool STORE>
int launch_chain_fwd(const ChainFwdArgs& a, void* stream) {
    static bool attr_set[64] = {};
    auto* fn = mlp_chain_fwd_kernel<KP1, A1, STORE>;
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

extern "C" int ag_mlp_chain_supported(int D, int C, int A1) {
    return (C == CF && chain_kp1(D) != 0 && (A1 == 5 || A1 == 6)) ? 1 : 0;
}

extern "C" long long ag_mlp_chain_image_bytes(int D) {
    const int kp1 = chain_kp1(D);
    if (kp1 == 0) return 0;
    return ((long long)(kp1 / 16 + 16) * CBLK + CWH) * 16;
}

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

extern "C" int ag_mlp_chain_forward(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
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
    const bool store = xn_dev || h1_dev || h2_dev;
#define AG_CHAIN_GO(KP, AV) (store ? launch_chain_fwd<KP, AV, true>(a, stream) : launch_chain_fwd<KP, AV, false>(a, stream))
    if (nb1 == 2) return A1 == 5 ? AG_CHAIN_GO(32, 5) : AG_CHAIN_GO(32, 6);
    return A1 == 5 ? AG_CHAIN_GO(64, 5) : AG_CHAIN_GO(64, 6);
#undef AG_CHAIN_GO
}
