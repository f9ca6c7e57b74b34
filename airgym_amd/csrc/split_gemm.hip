// split_gemm.hip - float32-accurate GEMM on the bf16 matrix cores of gfx950 (MI355X) for the 256-wide hidden layer of the
// actor-critic MLP (lib/network/mlp.py:36-39; forward X W^T and the backward dX = dZ W of a2c_continuous.py:299-369).
//
// Why: CDNA4 runs f32-input MFMA at the f32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 rate) and has no xf32/TF32 form
// (MI355X_MICROARCH.md).  The PPO update is three [196 608 x 256] x [256 x 256] GEMMs per minibatch, i.e. bound by that
// 157 TFLOP/s.  A float32 value splits EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, round to nearest):
//     a = a1 + a2 + a3,   |a2| <= 2^-8 |a|,  |a3| <= 2^-16 |a|       (likewise b)
// and every product ai*bj of two 8-bit significands is exact in the MFMA's f32 accumulate path.  Keeping the six terms
//     a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1
// drops only a2b3 + a3b2 + a3b3 <= (2^-24 + 2^-24 + 2^-32) |ab| < 2^-23 |ab|: at most one float32 ulp per product (an f32
// FMA rounds each step to half an ulp) - so the result differs from an f32 FMA chain by no more than the accumulation-order
// noise every f32 GEMM already has (tests/test_gpu_split_gemm.py measures both against float64).  Six bf16 MFMAs cost
// 6/16 of one f32 MFMA pass: 2.67x the f32-MFMA peak.  (Inf inputs become NaN: inf - inf in the split.)
//
// C[M, 256] = A[M, 256] * B^T, B given as prepared planes (ag_split_gemm_prepare): the weight matrix is split once per
// optimizer step into the exact LDS image the kernel wants, so the B side of the main loop is a straight 16-byte copy.
//   workgroup 256 threads = 4 waves as 2 x 2; block tile 128 rows x 256 columns (all of N); wave (wm, wn) owns rows
//   64 wm .. +63 and columns 128 wn .. +127 as 2 x 4 tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16, 128 accumulator
//   registers); K in chunks of 16, double-buffered in LDS:
//     A stage: [plane 3][k-half 2][row 128] x 16 B   (12 KB)   - the wave splits its f32 rows on the fly
//     B stage: [plane 3][k-half 2][col 256] x 16 B   (24 KB)
//   72 KB per workgroup -> two workgroups per CU, two waves per SIMD: one wave's LDS traffic and f32->bf16 splitting hide
//   under the other's MFMAs.  Fragment reads are ds_read_b128 over 32 consecutive 16-byte units: conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "ppo_loss_math.hpp"
#include "split_common.hpp"

namespace {

constexpr int BN = 256, BK = 16, KDIM = 256;
constexpr int B_UNITS = 3 * 2 * BN;       // 16-byte units per B stage
// row tile: WM waves x 64 rows (WM = 2: 128 rows, 4 waves, two workgroups per CU; WM = 4: 256 rows, 8 waves, one per CU)
constexpr int a_units(int bm) { return 3 * 2 * bm; }          // 16-byte units per A stage
constexpr int stage_units(int bm) { return a_units(bm) + B_UNITS; }

// Weight preparation: W [256, 256] f32 (row-major) -> planes [chunk 16][plane 3][k-half 2][n 256] x 8 bf16, the per-chunk LDS
// image of the main loop.  transpose = 0: B[n][k] = W[n][k] (forward, X W^T); 1: B[n][k] = W[k][n] (backward dX = dZ W).
__global__ __launch_bounds__(256) void split_prepare_kernel(const float* __restrict__ W, uint4* __restrict__ planes, int transpose,
                                                            uint4* __restrict__ planes_t) {
    // one (chunk, k-half, n) unit per thread: 16 * 2 * 256 = 8192 per image; blocks past the first image write the TRANSPOSED
    // image into planes_t (both in one launch: ag_split_gemm_prepare_pair)
    int unit = blockIdx.x * 256 + threadIdx.x;
    if (unit >= 16 * 2 * BN) {
        unit -= 16 * 2 * BN;
        planes = planes_t;
        transpose = !transpose;
    }
    if (unit >= 16 * 2 * BN) return;
    const int n = unit % BN, h = (unit / BN) & 1, c = unit / (2 * BN);
    const int k0 = c * BK + h * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = transpose ? W[(size_t)(k0 + i) * BN + n] : W[(size_t)n * KDIM + k0 + i];
    uint4 p1, p2, p3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
    uint4* chunk = planes + (size_t)c * B_UNITS;
    chunk[(0 * 2 + h) * BN + n] = p1;
    chunk[(1 * 2 + h) * BN + n] = p2;
    chunk[(2 * 2 + h) * BN + n] = p3;
}

// Images of ag_split_gemm_input_prepare (the fused first layer, FIN > 0):
//   [0, kInImageW1Bytes): W1ext [block 8][K step 2][plane 3][h 2][feature 32] x 16 B, W1ext[f][d] = W1[f][d] (d < D), b1[f] (d = D), 0
//   then the forward planes of W2 in the chain's K order, laid out like ag_split_gemm_prepare's
constexpr int kInImageW1Bytes = 8 * 2 * 3 * 2 * 32 * 16;
__global__ __launch_bounds__(256) void split_in_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D,
                                                               const float* __restrict__ W2, uint4* __restrict__ img,
                                                               uint4* __restrict__ planes_t) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 8 * 2 * 2 * 32) {                               // W1ext: one (block, step, h, feature) unit triple per thread
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
    int unit = t - 8 * 2 * 2 * 32;                          // W2 planes: one (chunk, k-half, n) unit triple per thread
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

// A1 > 0: the forward of the LAST hidden layer with the actor/critic heads folded into the epilogue
// (lib/network/mlp.py:36-39 followed by the mu / value Linear of a2c_continuous.py's model): C keeps the bias-free
// pre-activation z (the backward wants it), and heads[m, a] = sum_c ELU(z[m,c] + bias[c]) Wh[a,c] + bh[a] is formed from the
// accumulators while they are still in registers - the separate ELU + head pass re-read all of z (201 MB per minibatch).
// Per lane: 4 columns of 32 rows -> A1 partial dot products per row, summed over the 32 lanes of a half-wave with DPP row
// operations, over the two column halves (waves wn = 0, 1) through LDS.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sg_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
// sum over each 32-lane half of the wave; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float sg_half_sum(float v) {
    v = sg_dpp_add<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v = sg_dpp_add<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v = sg_dpp_add<0x141, 0xF>(v);       // row_half_mirror
    v = sg_dpp_add<0x140, 0xF>(v);       // row_mirror
    v = sg_dpp_add<0x142, 0xA>(v);       // row_bcast15: rows 1, 3 += lane 15 of rows 0, 2
    return v;
}
__device__ __forceinline__ float sg_elu(float z) {      // same form as ppo_kernels.hip elu1
    return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f;
}

//
// DIN > 0: the backward dX GEMM of the SECOND layer with the whole backward of the FIRST layer folded into its epilogue
// (autograd of mlp.py:36-39 for a [DIN -> 256 -> 256] trunk): the accumulators are dh1 = dz2 W2; the epilogue multiplies by
// ELU'(h1) (from the stored activations, h > 0 ? 1 : h + 1), and reduces dW1[c, d] = sum_m dz1[m, c] x[m, d], db1[c] =
// sum_m dz1[m, c] over the tile's 128 rows into one partial per tile.  Neither dh1 nor dz1 is ever written: nothing upstream
// of the first layer needs them, and the separate first-layer kernel re-read both dh1 and h1 (402 MB per minibatch).
// Per lane: 4 columns x 32 rows -> a [4][DIN] register tile (packed v_pk_fma_f32), the input rows broadcast from LDS; the two
// row halves of a wave (lanes l, l + 32) and the two row blocks (waves wm = 0, 1) are summed through LDS in a fixed order.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// First-layer backward on the matrix cores (split_gemm_kernel, DIN > 0).  For one column tile J of a wave:
//     G[d, c] = sum over the wave's 64 rows m of  x[m, d] * dz1[m, c],   dz1 = dh1 * ELU'(h1),   x[m, DIN] = 1 (bias gradient)
// as v_mfma_f32_32x32x16_bf16 products with K = rows.  The contraction index may be enumerated in any order as long as both
// operands agree, so K step ks of row tile i is taken to be exactly the rows held by accumulator registers 8 ks .. 8 ks + 7 of
// a lane (rows (e & 3) + 8 (e >> 2) + 16 ks + 4 h for lane half h): the B operand (dz1) is formed from the accumulators IN
// PLACE, no data movement; the A operand (x, pre-split into the same row order) comes from LDS planes built once per tile.
// Both operands are split three ways like the main product (six MFMAs per step): float32-accurate.
template <int J, int WM, int NDT>
__device__ __forceinline__ void input_wgrad_tile(const f32x16 (&acc)[8], const float* __restrict__ hcol, const uint4* xp_lane,
                                                 float* mine, int m0, int M, int wm, int khalf, int DIN1) {
    constexpr int XS = WM * 2 * 32 * NDT;                    // units per (plane, step) of the x image
    f32x16 g[NDT];                                           // NDT tiles of 32 input columns (Tracking's 48 + bias column: 2)
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = 0.0f;
    // activations of step s + PF are requested before step s is worked on (4 steps per column tile: i = s >> 1, ks = s & 1)
    constexpr int PF = 2;
    float ring[PF][8];
#define AG_IW_ROW(s_, e_) (wm * 64 + ((s_) >> 1) * 32 + ((e_) & 3) + 8 * ((e_) >> 2) + 16 * ((s_) & 1) + 4 * khalf)
#define AG_IW_FETCH(s_)                                                                               \
    _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                     \
        ring[(s_) % PF][e] = hcol[(size_t)min(m0 + AG_IW_ROW(s_, e), M - 1) * BN + 32 * J]
#pragma unroll
    for (int s = 0; s < PF; ++s) AG_IW_FETCH(s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float dz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = ring[s % PF][e];
            dz[e] = acc[(s >> 1) * 4 + J][8 * (s & 1) + e] * (y > 0.0f ? 1.0f : y + 1.0f);
        }
        if (s + PF < 4) AG_IW_FETCH(s + PF);
        uint4 b1, b2, b3;                                   // rows past M need no masking here: their x rows are zero
        split8(make_float4(dz[0], dz[1], dz[2], dz[3]), make_float4(dz[4], dz[5], dz[6], dz[7]), b1, b2, b3);
        const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(&b1), c2 = *reinterpret_cast<const bf16x8*>(&b2),
                     c3 = *reinterpret_cast<const bf16x8*>(&b3);
#pragma unroll
        for (int t = 0; t < NDT; ++t) {                     // the split dz1 fragment feeds every tile of input columns
            const uint4 ua1 = xp_lane[(0 * 4 + s) * XS + 32 * t], ua2 = xp_lane[(1 * 4 + s) * XS + 32 * t],
                        ua3 = xp_lane[(2 * 4 + s) * XS + 32 * t];
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&ua1), a2 = *reinterpret_cast<const bf16x8*>(&ua2),
                         a3 = *reinterpret_cast<const bf16x8*>(&ua3);
            AG_MFMA_SPLIT(g[t], a1, a2, a3, c1, c2, c3);
            asm volatile("" : "+v"(g[t]));                  // this step's products stay in this step (see the head epilogue)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef AG_IW_FETCH
#undef AG_IW_ROW
    // g[t]: column c = this lane's column of tile J, rows d = 32 t + (r & 3) + 8 (r >> 2) + 4 h; keep d <= DIN (d = DIN: bias
    // gradient).  NDT == 1: `mine` is this lane's row of the whole-tile buffer (column tile J at + J * 32 rows); NDT == 2: of the
    // per-column-tile buffer.
    float* dst = mine + (NDT == 1 ? J * 32 * DIN1 : 0);
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (d < DIN1) dst[d] = g[t][r];
        }
}

// LDS layout of the recomputing dX epilogue (RC, DIN > 0): [x image, K = rows order: 3 x 4 x XS units][red: WM x 256 x (DIN + 1)
// floats] over the (idle) stages, then the first-layer image (loaded once, at kernel start) behind whichever is larger
template <int DIN, int WM>
constexpr int rc_w1_offset_units() {
    constexpr int epi = 3 * 4 * (WM * 2 * (DIN + 2)) + (WM * BN * (DIN + 1) * 4 + 15) / 16;
    return epi > 2 * stage_units(WM * 64) ? epi : 2 * stage_units(WM * 64);
}

struct SplitEpilogue {
    const float* bias;         // [256] layer bias: added to C (plain), or inside the ELU only (heads)
    const float* Wh;           // heads: [A1, 256]
    const float* bh;           // heads: [A1]
    float* heads;              // heads: [M, A1]
    const float* h1;           // input wgrad: [M, 256] first-layer activations
    const float* x;            // input wgrad: [M, DIN] (normalised) network inputs
    float* dw_partials;        // input wgrad: [tiles, 256, DIN]
    float* db_partials;        // input wgrad: [tiles, 256]
    // LOSS (ag_split_gemm_loss_heads_bwd): the minibatch rows of the PPO loss and where the head layer's backward goes
    const float* logstd;       // [A]
    const float* actions;      // [M, A]
    const float* old_neglogp;  // [M]
    const float* advantages;   // [M]
    const float* returns;      // [M]
    const float* old_values;   // [M]
    const float* old_mu;       // [M, A]
    const float* old_sigma;    // [M, A]
    float* new_mu;             // [M, A] or null
    float* new_sigma;          // [M, A] or null
    float* loss_partials;      // [tiles, agloss::kNumSums]
    float* dwh_partials;       // [tiles, A1, 256] head weight gradient of the tile's rows
    float* db2_partials;       // [tiles, 256] column sums of dz (bias gradient of this layer)
    agloss::LossParams lp;
    // FIN (ag_split_gemm_input_loss_heads_bwd): the FIRST layer formed on the fly as this GEMM's A operand; A = the raw observations
    const double* in_mean;     // [FIN] running mean / variance of the input normaliser, or null (A is used as it is)
    const double* in_var;
    const uint4* w1img;        // first-layer weight + bias image (ag_split_gemm_input_prepare)
    float* xn_out;             // [M, FIN] normalised inputs (null iff in_mean is null)
    float* h1_out;             // [M, 256] first-layer activations (the backward reads them)
    float in_eps, in_clip;
};

//
// FIN > 0: the first layer of a [FIN -> 256 -> 256] trunk inside this launch (replaces ag_mlp_input_layer in the update): the A
// operand of the main product is PRODUCED, not loaded - see the FIN block in the kernel.  h1 still goes to HBM (the backward reads
// it), but as stores under this kernel's MFMAs instead of a write-bound launch of its own, and the 201 MB read of it is gone.  The
// W2 planes of this launch are in the chain's K order (chunk 2 b + q, k-half h, slot i <-> feature 32 b + 16 q + (i & 3) + 8 (i >> 2)
// + 4 h).  A version that formed the piece on the vector ALU (144 FMAs per thread and chunk) was 0.7 ms per epoch SLOWER.
// RC (round 5: h1 is recomputed, not stored): with FIN > 0 the launch does not write h1_out; with DIN > 0 the epilogue forms
// ELU'(h1) from a natural-orientation recomputation of h1 = ELU(W1ext x_ext) on the matrix cores (rc_input_wgrad below) instead of
// reading the stored activations.
template <bool LOSS, int DIN>
constexpr bool split_persistent() {
#if defined(AG_SPLIT_NOT_PERSISTENT)
    return false;
#elif defined(AG_SPLIT_PERSISTENT_ALL)
    return true;
#else
    return !LOSS;                   // (the loss launches are 1.5 - 3 % slower inside the loop: measured, DESIGN_HISTORY.md)
#endif
}

template <bool HAS_BIAS, int A1, int DIN, int WM, bool LOSS = false, int FIN = 0, bool RC = false>
__global__ __launch_bounds__(WM * 128, 2) void split_gemm_kernel(const float* __restrict__ A, const uint4* __restrict__ Bp,
                                                                  float* __restrict__ C, int M, const SplitEpilogue ep) {
    constexpr int BM = WM * 64, NT = WM * 128;               // rows per tile, threads per workgroup
    constexpr int A_UNITS = a_units(BM), STAGE_UNITS = stage_units(BM);
    constexpr int BPT = B_UNITS / NT;                        // B-plane units copied per thread per chunk (6 or 3)
    static_assert(WM == 2 || WM == 4, "4 or 8 waves");
    static_assert(A1 == 0 || DIN == 0, "one fused epilogue at a time");
    static_assert((DIN & 1) == 0 && DIN <= 62, "input width: even, at most two 32-wide tiles with the bias column");
    static_assert(FIN == 0 || ((FIN & 1) == 0 && FIN >= 16 && FIN <= 20), "fused first layer: widths 16 / 18 / 20");
    static_assert(!RC || FIN > 0 || (DIN > 0 && DIN <= 18), "recomputed h1: the fused forward, or the dX epilogue (widths <= 18)");
    const float* __restrict__ bias = ep.bias;
    const float* __restrict__ Wh = ep.Wh;
    const float* __restrict__ bh = ep.bh;
    float* __restrict__ heads = ep.heads;
    extern __shared__ uint4 lds[];                         // [2 stages][A_UNITS + B_UNITS] 16-byte units
    if constexpr (DIN > 0 && RC) {      // the first-layer image, behind the stages / the epilogue's buffers (visible after the first barrier)
        uint4* w1d = lds + rc_w1_offset_units<DIN, WM>();
        for (int u = threadIdx.x; u < kInImageW1Bytes / 16; u += NT) w1d[u] = ep.w1img[u];
    }
    // Persistent workgroups (late round 5): the launcher starts at most as many workgroups as fit the chip at once and each walks
    // the row tiles blockIdx.x, + gridDim.x, ... - the first-layer image is copied into LDS once per workgroup instead of once per
    // tile, and no workgroup has to be dispatched (LDS allocated, waves started) between a CU's tiles.
    // (`split_persistent` - also the launcher's rule - says which instantiations: not those the loop would make spill)
    constexpr bool PERSIST = split_persistent<LOSS, DIN>();
    const int ntiles_ = (M + BM - 1) / BM;
    for (int tile_it = blockIdx.x; tile_it < ntiles_; tile_it += gridDim.x) {
    int tile = tile_it;
    asm volatile("" : "+s"(tile));      // opaque per iteration: nothing that depends on the tile is hoisted out of the loop
    int tid_op = threadIdx.x;           // ... nor anything that depends on the thread index (it would be live across the epilogue)
    if constexpr (PERSIST) asm volatile("" : "+v"(tid_op));
    const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
    const int m0 = tile * BM;
    const bool first_tile = tile_it == (int)blockIdx.x;
    if (!first_tile) __syncthreads();   // every wave is done with the previous tile's epilogue buffers (they overlay the stages)

    // ---- global -> register staging for one K chunk
    const int a_row = tid >> 1, a_half = tid & 1;                       // BM rows x 2 k-halves: one 32-byte piece per thread
    const int a_grow = min(m0 + a_row, M - 1);                          // rows past M are computed, never stored
    const float4* a_src = reinterpret_cast<const float4*>(A + (size_t)a_grow * KDIM + a_half * 8);
    float4 ra0, ra1;
#ifndef AG_SPLIT_B_STAGED
    // The B planes are the LDS image already, so they go global -> LDS directly (global_load_lds_dwordx4: a wave's 64 units land
    // at M0 + 16 lane): no staging registers, no ds_write (-DAG_SPLIT_B_STAGED restores the copy through registers: epoch +0.4 %).
    // Issued from inline assembly: told about an LDS-DMA, hipcc waits for vmcnt(0) in front of the next ds_read.  Its own vmcnt
    // counts for the A loads stay correct (an unseen younger operation can only make a wait longer).  DMA(c + 1) is issued at the
    // top of chunk c (every wave is past the barrier behind that stage's last reader) and drained by an explicit vmcnt(0) behind
    // the chunk's stores, before the next A loads are issued.
    const uint32_t lds_b0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds +
                                                           (A_UNITS + wave * 64) * 16);
#define AG_SG_DMA(c, stage)                                                            \
    do {                                                                               \
        const uint4* bsrc_ = Bp + (size_t)(c) * B_UNITS + tid;                         \
        _Pragma("unroll") for (int it = 0; it < BPT * kPlanes / 3; ++it)               \
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"   \
                         : : "v"(bsrc_ + it * NT), "s"(lds_b0 + ((stage) * STAGE_UNITS + it * NT) * 16) : "memory"); \
    } while (0)
#define AG_SG_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" : : : "memory")
#define AG_SG_LOAD(c)                                                                  \
    do {                                                                               \
        ra0 = a_src[(c) * (BK / 4)];                                                   \
        ra1 = a_src[(c) * (BK / 4) + 1];                                               \
    } while (0)
#define AG_SG_STORE(stage)                                                             \
    do {                                                                               \
        uint4* sa_ = lds + (stage) * STAGE_UNITS;                                      \
        uint4 p1_, p2_, p3_;                                                           \
        split8(ra0, ra1, p1_, p2_, p3_);                                               \
        sa_[(0 * 2 + a_half) * BM + a_row] = p1_;                                      \
        if (kPlanes == 3) {                                                            \
            sa_[(1 * 2 + a_half) * BM + a_row] = p2_;                                  \
            sa_[(2 * 2 + a_half) * BM + a_row] = p3_;                                  \
        }                                                                              \
    } while (0)
    AG_SG_DMA(0, 0);
#else
    static_assert(FIN == 0, "the fused first layer goes with the LDS-DMA B planes");
    uint4 rb0, rb1, rb2, rb3, rb4, rb5;                                 // (rb3..5: 4-wave tiles only; named, not an array: LLVM
    static_assert(BPT == 3 || BPT == 6, "B copy per thread");           //  left an indexed array in scratch)
#define AG_SG_DMA(c, stage) do { } while (0)
#define AG_SG_DMA_WAIT() do { } while (0)
#define AG_SG_LOAD(c)                                                                  \
    do {                                                                               \
        ra0 = a_src[(c) * (BK / 4)];                                                   \
        ra1 = a_src[(c) * (BK / 4) + 1];                                               \
        const uint4* bsrc_ = Bp + (size_t)(c) * B_UNITS + tid;                         \
        rb0 = bsrc_[0]; rb1 = bsrc_[NT]; rb2 = bsrc_[2 * NT];                          \
        if (BPT == 6) { rb3 = bsrc_[3 * NT]; rb4 = bsrc_[4 * NT]; rb5 = bsrc_[5 * NT]; } \
    } while (0)
#define AG_SG_STORE(stage)                                                             \
    do {                                                                               \
        uint4* sa_ = lds + (stage) * STAGE_UNITS;                                      \
        uint4* sb_ = sa_ + A_UNITS + tid;                                              \
        uint4 p1_, p2_, p3_;                                                           \
        split8(ra0, ra1, p1_, p2_, p3_);                                               \
        sa_[(0 * 2 + a_half) * BM + a_row] = p1_;                                      \
        sa_[(1 * 2 + a_half) * BM + a_row] = p2_;                                      \
        sa_[(2 * 2 + a_half) * BM + a_row] = p3_;                                      \
        sb_[0] = rb0; sb_[NT] = rb1; sb_[2 * NT] = rb2;                                \
        if (BPT == 6) { sb_[3 * NT] = rb3; sb_[4 * NT] = rb4; sb_[5 * NT] = rb5; }     \
    } while (0)
#endif

    // ---- FIN: the first layer on the matrix cores, TRANSPOSED (the chain kernel's first layer, mlp_chain.hip): wave w produces
    // rows 32 w .. + 31 of the tile; per block of 32 features  h1^T[f, row] = W1ext[f, :] . x_ext[row, :]  (K = 32: the FIN inputs, an
    // all-ones column that carries the bias, zeros) as 2 x 6 split MFMAs with the weights as A operand (image in LDS) and the lane's row
    // of inputs as B operand (registers, for the whole tile).  Lane (row, h) then holds features 32 b + (r & 3) + 8 (r >> 2) + 4 h in
    // register r: registers 0..7 and 8..15 ARE one 16-byte A unit each of the main product (chunks 2 b and 2 b + 1, k-half h) in the K
    // order the W2 planes of this launch are prepared in (ag_split_gemm_input_prepare) - bias + ELU + split in registers, no shuffle.
    bf16x8 xq[2][3];
    f32x16 hacc;
    float4 hs0, hs1;
    int hoff = 0;
    uint4* const w1s = lds + 2 * STAGE_UNITS;           // [block 8][K step 2][plane 3][h 2][feature 32] x 16 B (FIN only)
    const int frow = wave * 32 + (lane & 31), fh = lane >> 5;
#define AG_FIN_MFMA(b_)                                                                                \
    do {                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                             \
            bf16x8 wa_[3];                                                                             \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
                const uint4 u_ = w1s[((((b_) * 2 + s_) * 3 + p) * 2 + fh) * 32 + (lane & 31)];         \
                wa_[p] = *reinterpret_cast<const bf16x8*>(&u_);                                        \
            }                                                                                          \
            AG_MFMA_SPLIT(hacc, wa_[0], wa_[1], wa_[2], xq[s_][0], xq[s_][1], xq[s_][2]);                 \
        }                                                                                              \
    } while (0)
    // half q of the block (registers 8 q .. 8 q + 7) -> chunk 2 b + q: ELU, split, the three plane units of (row, k-half h)
#define AG_FIN_EMIT(b_, q_, stage_)                                                                    \
    do {                                                                                               \
        float e_[8];                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) e_[i] = sg_elu(hacc[8 * (q_) + i]);              \
        hs0 = make_float4(e_[0], e_[1], e_[2], e_[3]);                                                 \
        hs1 = make_float4(e_[4], e_[5], e_[6], e_[7]);                                                 \
        hoff = 32 * (b_) + 16 * (q_) + 4 * fh;                                                         \
        uint4* sa_ = lds + (stage_) * STAGE_UNITS;                                                     \
        uint4 p1_, p2_, p3_;                                                                           \
        split8(hs0, hs1, p1_, p2_, p3_);                                                               \
        sa_[(0 * 2 + fh) * BM + frow] = p1_;                                                           \
        if (kPlanes == 3) {                                                                            \
            sa_[(1 * 2 + fh) * BM + frow] = p2_;                                                       \
            sa_[(2 * 2 + fh) * BM + frow] = p3_;                                                       \
        }                                                                                              \
    } while (0)
    // ... and the same values to HBM (the backward reads h1), issued BEHIND the step's vmcnt(0) (which then finds only stores a
    // whole chunk old): features hoff .. + 3 and hoff + 8 .. + 11 of the lane's row
#define AG_FIN_STORE()                                                                                 \
    do {                                                                                               \
        if (!RC && m0 + frow < M) {                                                                           \
            float* d_ = ep.h1_out + (size_t)(m0 + frow) * KDIM + hoff;                                 \
            *reinterpret_cast<float4*>(d_) = hs0;                                                      \
            *reinterpret_cast<float4*>(d_ + 8) = hs1;                                                  \
        }                                                                                              \
    } while (0)
    if constexpr (FIN > 0) {
        if (first_tile) {
            for (int u = tid; u < 8 * 2 * 3 * 2 * 32; u += NT) w1s[u] = ep.w1img[u];
        }
        const bool norm = ep.in_mean != nullptr;
        const int grow = min(m0 + frow, M - 1);
        const float* xrow = A + (size_t)grow * FIN;
        float xv[2][8];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {                    // K step 0: inputs 8 h .. 8 h + 7 (all below FIN >= 16)
            const float2 v2 = reinterpret_cast<const float2*>(xrow + 8 * fh)[i2];
            xv[0][2 * i2] = v2.x;
            xv[0][2 * i2 + 1] = v2.y;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[1][i] = 0.0f;         // K step 1: inputs 16 .. FIN - 1, the bias column at FIN, zeros
        if (fh == 0) {
#pragma unroll
            for (int i2 = 0; i2 < (FIN - 16) / 2; ++i2) {
                const float2 v2 = reinterpret_cast<const float2*>(xrow + 16)[i2];
                xv[1][2 * i2] = v2.x;
                xv[1][2 * i2 + 1] = v2.y;
            }
        }
        if (norm) {      // ag_mlp_input_layer's arithmetic (running_mean_std.py:78-79)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int d = 16 * s_ + 8 * fh + i;
                    if (d < FIN) {
                        float v = (xv[s_][i] - (float)ep.in_mean[d]) / sqrtf((float)ep.in_var[d] + ep.in_eps);
                        v = fminf(fmaxf(v, -ep.in_clip), ep.in_clip);
                        xv[s_][i] = v;
                        if (m0 + frow < M) ep.xn_out[(size_t)(m0 + frow) * FIN + d] = v;
                    }
                }
        }
        if (fh == 0) xv[1][FIN - 16] = 1.0f;                // the all-ones column: W1ext[f, FIN] = b1[f]
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            uint4 q1, q2, q3;
            split8(make_float4(xv[s_][0], xv[s_][1], xv[s_][2], xv[s_][3]), make_float4(xv[s_][4], xv[s_][5], xv[s_][6], xv[s_][7]),
                   q1, q2, q3);
            xq[s_][0] = *reinterpret_cast<const bf16x8*>(&q1);
            xq[s_][1] = *reinterpret_cast<const bf16x8*>(&q2);
            xq[s_][2] = *reinterpret_cast<const bf16x8*>(&q3);
        }
        __syncthreads();                                     // the W1 image is in LDS
        AG_FIN_MFMA(0);
        AG_FIN_EMIT(0, 0, 0);
        AG_SG_DMA_WAIT();
        AG_FIN_STORE();
        __syncthreads();
    } else {
        AG_SG_LOAD(0);
        AG_SG_STORE(0);
        AG_SG_DMA_WAIT();
        int c1 = 1;      // opaque, so that chunk 1's loads stay BEHIND chunk 0's stores and reuse its staging registers (hoisted to
        asm volatile("" : "+s"(c1) : : "memory");      // the top they need a second set: spills, each behind an s_waitcnt vmcnt(0))
        AG_SG_LOAD(c1);
        __syncthreads();
    }

    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    const int l31 = lane & 31, khalf = lane >> 5;
    constexpr int NCHUNK = KDIM / BK;
#ifdef AG_SPLIT_SETPRIO
    // static priority for the later-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md, "two waves per SIMD" item 4)
    if (WM == 4 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    // WM x 2 waves: wave (wm, wn) owns rows 64 wm .. +63 and columns 128 wn .. +127 (2 x 4 tiles of 32 x 32): 18 fragment
    // reads per chunk (every B fragment feeds two row tiles).  Products smallest first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1.
    const int wm = wave >> 1, wn = wave & 1;
#define AG_SG_COMPUTE(stage_)                                                                          \
    do {                                                                                               \
        const uint4* sa_ = lds + (stage_) * STAGE_UNITS;                                               \
        const uint4* sb_ = sa_ + A_UNITS;                                                              \
        bf16x8 a_[2][3];                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
                const uint4 u_ = sa_[(p * 2 + khalf) * BM + wm * 64 + i * 32 + l31];                   \
                a_[i][p] = *reinterpret_cast<const bf16x8*>(&u_);                                      \
            }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
            const uint4 ub0_ = sb_[(0 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];                    \
            const uint4 ub1_ = sb_[(1 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];                    \
            const uint4 ub2_ = sb_[(2 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];                    \
            const bf16x8 b0_ = *reinterpret_cast<const bf16x8*>(&ub0_);                                \
            const bf16x8 b1_ = *reinterpret_cast<const bf16x8*>(&ub1_);                                \
            const bf16x8 b2_ = *reinterpret_cast<const bf16x8*>(&ub2_);                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
                f32x16& d_ = acc[i * 4 + j];                                                           \
                AG_MFMA_SPLIT(d_, a_[i][0], a_[i][1], a_[i][2], b0_, b1_, b2_);                        \
            }                                                                                          \
        }                                                                                              \
    } while (0)
    // Global loads run TWO chunks ahead of the MFMAs and are issued at the END of a chunk, right after the registers they
    // fill were drained into LDS: issued at the top of the chunk that precedes their use, LLVM sinks them into the store block
    // (same condition, only user there) and every chunk ends waiting for HBM; here they have a full chunk to land and cost
    // no extra registers.
    if constexpr (FIN > 0) {
        // chunk pairs: the even chunk's step emits the second half of the current feature block, the odd chunk's step produces the
        // next block (12 MFMAs) and emits its first half
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK / 2 - 1; ++cc) {
            AG_SG_DMA(2 * cc + 1, 1);
            AG_SG_COMPUTE(0);
            AG_FIN_EMIT(cc, 1, 1);
            AG_SG_DMA_WAIT();
            AG_FIN_STORE();
            __syncthreads();
            AG_SG_DMA(2 * cc + 2, 0);
            AG_SG_COMPUTE(1);
            AG_FIN_MFMA(cc + 1);
            AG_FIN_EMIT(cc + 1, 0, 0);
            AG_SG_DMA_WAIT();
            AG_FIN_STORE();
            __syncthreads();
        }
        AG_SG_DMA(NCHUNK - 1, 1);
        AG_SG_COMPUTE(0);
        AG_FIN_EMIT(NCHUNK / 2 - 1, 1, 1);
        AG_SG_DMA_WAIT();
        AG_FIN_STORE();
        __syncthreads();
        AG_SG_COMPUTE(1);
    } else {
#ifdef AG_SPLIT_PINGPONG
        // The two waves of a SIMD (waves w and w + 4 of the 8-wave workgroup) half a chunk out of phase: the first-dispatched half
        // multiplies chunk c and THEN stages chunk c + 1, the second half stages FIRST and then multiplies - so that one wave's
        // staging (split arithmetic, LDS stores, address work) runs beside the other's MFMAs instead of beside its staging.
        const bool stage_first = (WM == 4) && (__builtin_amdgcn_readfirstlane(wave) >= 4);
        if (stage_first) {
#pragma unroll 1
            for (int c = 0; c < NCHUNK - 1; ++c) {
                const int stage = c & 1;
                AG_SG_STORE(stage ^ 1);                     // chunk c + 1 (loaded during chunk c - 1)
                AG_SG_DMA(c + 1, stage ^ 1);                // (behind the store: its wait for the A registers must not cover the DMA)
                const int cn = (c + 2 < NCHUNK) ? c + 2 : NCHUNK - 1;
                AG_SG_LOAD(cn);
#ifndef AG_SPLIT_PINGPONG_LOOSE
                __builtin_amdgcn_sched_barrier(0);
#endif
                AG_SG_COMPUTE(stage);
                AG_SG_DMA_WAIT();
                __syncthreads();
            }
        } else
#endif
        {
#pragma unroll 1
        for (int c = 0; c < NCHUNK - 1; ++c) {
            const int stage = c & 1;
            AG_SG_DMA(c + 1, stage ^ 1);
            AG_SG_COMPUTE(stage);
#if defined(AG_SPLIT_PINGPONG) && !defined(AG_SPLIT_PINGPONG_LOOSE)
            __builtin_amdgcn_sched_barrier(0);
#endif
            AG_SG_STORE(stage ^ 1);                         // chunk c + 1; that stage was last read before the previous barrier
            AG_SG_DMA_WAIT();
            const int cn = (c + 2 < NCHUNK) ? c + 2 : NCHUNK - 1;      // (the last trip reloads a chunk it does not need)
            AG_SG_LOAD(cn);
            __syncthreads();
        }
        }
        AG_SG_COMPUTE((NCHUNK - 1) & 1);
    }
    if (A1 > 0 || DIN > 0) __syncthreads();                 // the fused epilogues reuse the stages
#undef AG_SG_COMPUTE
#undef AG_SG_LOAD
#undef AG_SG_STORE
#undef AG_SG_DMA
#undef AG_SG_DMA_WAIT
#undef AG_FIN_MFMA
#undef AG_FIN_EMIT
#undef AG_FIN_STORE

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (A1 > 0) {
        int late = 0;       // opaque zero: keeps the 4 + 4 A1 per-column constants (tile-invariant, so LLVM would hoist them out
        asm volatile("" : "+s"(late) : : "memory");      // of the persistent loop) out of the main loop's live ranges
        float bcol[4], wcol[A1][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wn * 128 + j * 32 + l31 + late;
            bcol[j] = bias[col];
#pragma unroll
            for (int a = 0; a < A1; ++a) wcol[a][j] = Wh[a * BN + col];
        }
        float* hs = reinterpret_cast<float*>(lds);          // [wn 2][row BM][A1]; the stages are idle after the last barrier
        const int sel = lane & 15;
        float* hs_lane = hs + (wn * BM + wm * 64 + 4 * khalf) * A1 + sel + late;      // + a compile-time row offset per store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const bool live = m0 + rloc < M;
                float part[A1];
#pragma unroll
                for (int a = 0; a < A1; ++a) part[a] = 0.0f;
#ifdef AG_K1_ABL_NO_PASS1       // timing ablation (results wrong): no ELU + head products in the first pass
                if (LOSS) part[0] = acc[i * 4][r];
                else
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[i * 4 + j][r];
                    if (!LOSS && live) C[(size_t)(m0 + rloc) * BN + wn * 128 + j * 32 + l31] = v;
                    const float e = sg_elu(v + bcol[j]);
#pragma unroll
                    for (int a = 0; a < A1; ++a) part[a] = fmaf(e, wcol[a][j], part[a]);
                }
                float out = 0.0f;
#ifdef AG_K1_ABL_NO_ROWSUM      // timing ablation (results wrong): no DPP row sums of the head products
#pragma unroll
                for (int a = 0; a < A1; ++a) out += part[a];
#else
#pragma unroll
                for (int a = 0; a < A1; ++a) {
                    const float sum = sg_half_sum(part[a]);
                    out = (sel == a) ? sum : out;
                }
#endif
                if ((lane & 16) && sel < A1) hs_lane[(i * 32 + (r & 3) + 8 * (r >> 2)) * A1] = out;
                __builtin_amdgcn_sched_barrier(0);      // one row group at a time: interleaving them spills
            }
        }
        __syncthreads();
        if constexpr (!LOSS) {
            if (tid < BM && m0 + tid < M) {
#pragma unroll
                for (int a = 0; a < A1; ++a)
                    heads[(size_t)(m0 + tid) * A1 + a] = hs[tid * A1 + a] + hs[(BM + tid) * A1 + a] + bh[a];
            }
        } else {
            // ---- the PPO loss of the tile's rows (one thread per row; ppo_loss_math.hpp = ag_ppo_loss's arithmetic), then the
            //      head layer's backward from the activations still in the accumulators: dz = (d_heads Wh) * ELU'(h) written where
            //      the pre-activation would have gone, the head weight gradient dWh[a, c] = sum_m d_heads[m, a] h[m, c] and this
            //      layer's bias gradient db[c] = sum_m dz[m, c] as one partial per tile.  Replaces ag_ppo_loss +
            //      ag_heads_bwd_elu_wgrad: heads / d_heads never leave the LDS, z is neither written nor re-read.
            constexpr int AA = A1 - 1;
            float* dhs = hs + 2 * BM * A1;                      // [row BM][A1] d loss / d heads
            float* lred = dhs + BM * A1;                        // [4 waves][kNumSums]
            float* red2 = hs + 6144;                            // [wm WM][khalf 2][col BN][A1 + 1] (floats 6144 ..)
            static_assert((6144 + WM * 2 * BN * (A1 + 1)) * 4 <= 2 * stage_units(BM) * 16, "loss epilogue exceeds the stages");
            float lacc[agloss::kNumSums];
#pragma unroll
            for (int q = 0; q < agloss::kNumSums; ++q) lacc[q] = 0.0f;
            // the constants of the state-independent sigma: once per tile (wave 7, lane 0), through LDS
            float* lcs = lred + 4 * agloss::kNumSums;           // LossConsts<AA> as floats
            static_assert(sizeof(agloss::LossConsts<AA>) % 4 == 0, "LossConsts is an array of floats");
            static_assert(2 * BM * A1 + BM * A1 + 4 * agloss::kNumSums + (int)sizeof(agloss::LossConsts<AA>) / 4 <= 6144, "LDS");
            if (tid == NT - 1) {
                agloss::LossConsts<AA> lc0;
                agloss::loss_consts<AA>(ep.logstd, ep.lp.e_clip, lc0);
                *reinterpret_cast<agloss::LossConsts<AA>*>(lcs) = lc0;
            }
            __syncthreads();
            if (tid < BM) {
                const int row = m0 + tid;
                float dh[A1];
#pragma unroll
                for (int a = 0; a < A1; ++a) dh[a] = 0.0f;
                if (row < M) {
                    const agloss::LossConsts<AA> lc = *reinterpret_cast<const agloss::LossConsts<AA>*>(lcs);
                    float hv[A1], act[AA], om[AA], os[AA];
#pragma unroll
                    for (int a = 0; a < A1; ++a) hv[a] = hs[tid * A1 + a] + hs[(BM + tid) * A1 + a] + bh[a];
#pragma unroll
                    for (int a = 0; a < AA; ++a) {
                        act[a] = ep.actions[(size_t)row * AA + a];
                        om[a] = ep.old_mu[(size_t)row * AA + a];
                        os[a] = ep.old_sigma[(size_t)row * AA + a];
                    }
                    if (heads != nullptr) {
#pragma unroll
                        for (int a = 0; a < A1; ++a) heads[(size_t)row * A1 + a] = hv[a];
                    }
                    if (ep.new_mu != nullptr) {
#pragma unroll
                        for (int a = 0; a < AA; ++a) {
                            ep.new_mu[(size_t)row * AA + a] = hv[a];
                            ep.new_sigma[(size_t)row * AA + a] = lc.sig[a];
                        }
                    }
                    agloss::loss_row<AA, true>(hv, act, ep.old_neglogp[row], ep.advantages[row], ep.returns[row], ep.old_values[row],
                                               om, os, lc, ep.lp, dh, lacc);
                }
#pragma unroll
                for (int a = 0; a < A1; ++a) dhs[tid * A1 + a] = dh[a];      // rows past M: zeros (they add nothing below)
#pragma unroll
                for (int q = 0; q < agloss::kNumSums; ++q) {
                    float x = lacc[q];
                    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
                    if (lane == 0) lred[wave * agloss::kNumSums + q] = x;
                }
            }
            __syncthreads();
            if (tid < agloss::kNumSums) {
                float x = 0.0f;
                for (int w = 0; w < BM / 64; ++w) x += lred[w * agloss::kNumSums + tid];
                ep.loss_partials[(size_t)tile * agloss::kNumSums + tid] = x;
            }
            int late2 = 0;       // opaque zero born HERE: or the 128 row addresses of the dz stores are formed in front of the
            asm volatile("" : "+s"(late2) : : "memory");      // loss phase and spilled across it
            const int m0b = m0 + late2;
            float bcol2[4];      // the bias again, opaque: the activations are RECOMPUTED below (as values common to both passes
#pragma unroll                   // LLVM keeps all 128 of them alive across the loss phase - spilled; writing them over the
                                 // accumulators in pass 1 instead: 58 registers spilled in the epilogue, epoch +0.45 ms)
            for (int j = 0; j < 4; ++j) bcol2[j] = bcol[j] + (float)late2;
            float gw[A1][4], db[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                db[j] = 0.0f;
#pragma unroll
                for (int a = 0; a < A1; ++a) gw[a][j] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rloc = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;      // (M is a multiple of the tile)
                    float d[A1];
#pragma unroll
                    for (int a = 0; a < A1; ++a) d[a] = dhs[rloc * A1 + a];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#ifdef AG_K1_ABL_NO_PASS2       // timing ablation (results wrong): the stores only
                        const float o = acc[i * 4 + j][r] + d[0];
                        gw[0][j] += o;
#else
                        const float e = sg_elu(acc[i * 4 + j][r] + bcol2[j]);
                        float g = 0.0f;
#pragma unroll
                        for (int a = 0; a < A1; ++a) {
                            g = fmaf(d[a], wcol[a][j], g);
                            gw[a][j] = fmaf(d[a], e, gw[a][j]);
                        }
                        const float o = g * (e > 0.0f ? 1.0f : e + 1.0f);
#endif
                        C[(size_t)(m0b + rloc) * BN + wn * 128 + j * 32 + l31] = o;
                        db[j] += o;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the two row halves of a wave and the WM row blocks in a fixed order: deterministic
            {
                float* mine = red2 + (((size_t)wm * 2 + khalf) * BN + wn * 128 + l31) * (A1 + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int a = 0; a < A1; ++a) mine[(j * 32) * (A1 + 1) + a] = gw[a][j];
                    mine[(j * 32) * (A1 + 1) + A1] = db[j];
                }
            }
            __syncthreads();
            for (int idx = tid; idx < BN * (A1 + 1); idx += NT) {
                const int c = idx / (A1 + 1), q = idx - c * (A1 + 1);
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < WM * 2; ++w) v += red2[((size_t)w * BN + c) * (A1 + 1) + q];
                if (q < A1) ep.dwh_partials[((size_t)tile * A1 + q) * BN + c] = v;
                else ep.db2_partials[(size_t)tile * BN + c] = v;
            }
        }
    } else if constexpr (DIN > 0 && RC) {
        // ---- first-layer backward with h1 RECOMPUTED (round 5; h1 is no longer stored).  Per row tile i and column tile J of the
        // wave: z1[row, f] = x_ext[row, :] . W1ext[f, :] as a natural-orientation MFMA tile (K = 32: inputs, ones column = bias,
        // zeros; 2 x 6 split MFMAs; A = the lane's row of x, split once per row tile; B = the first-layer image, the same units the
        // forward reads as its A operand) lands in exactly the accumulator layout of dh1 (lane = column, registers = rows), so
        // dz1 = dh1 * ELU'(z1), ELU'(z) = z > 0 ? 1 : e^z, is formed register by register and feeds the K = rows products of
        // input_wgrad_tile unchanged.  +96 MFMAs per wave and tile; the 128 strided h1 loads per lane (201 MB per launch) are gone.
        constexpr int RW = DIN + 1;                         // reduction row: DIN weight-gradient entries + the bias gradient
        constexpr int DW = RW + 1;                          // units per (wm, h): d < RW real, d = RW zeros (lanes >= RW read that one)
        constexpr int XS = WM * 2 * DW;                     // units per (plane, step)
        uint4* xp = lds;                                    // [plane 3][step 4][wm WM][h 2][d DW] x 16 B: x, split, in K = rows order
        float* red = reinterpret_cast<float*>(lds + 3 * 4 * XS);       // [wm WM][BN][RW]
        const uint4* w1s = lds + rc_w1_offset_units<DIN, WM>();        // [block 8][K step 2][plane 3][h 2][feature 32] x 16 B
        int late = 0;                                       // (opaque zero, as above)
        asm volatile("" : "+s"(late) : : "memory");
        const int m0e = m0 + late;
        // the lane's rows of x for the two row tiles (natural A fragments: lane = row, K step 0 = inputs 8 h .. 8 h + 7, K step 1 =
        // inputs 16 .. DIN - 1, the ones column, zeros - lane half 1 holds zeros there)
        float xr0[2][8], xr1[2][DIN - 16 > 0 ? DIN - 16 : 1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = min(m0e + wm * 64 + i * 32 + l31, M - 1);
            const float* xrow = ep.x + (size_t)row * DIN;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const float2 v2 = reinterpret_cast<const float2*>(xrow + 8 * khalf)[i2];
                xr0[i][2 * i2] = v2.x;
                xr0[i][2 * i2 + 1] = v2.y;
            }
#pragma unroll
            for (int i2 = 0; i2 < (DIN - 16) / 2; ++i2) {
                const float2 v2 = reinterpret_cast<const float2*>(xrow + 16)[i2];
                xr1[i][2 * i2] = v2.x;
                xr1[i][2 * i2 + 1] = v2.y;
            }
        }
        for (int u = tid; u < 4 * XS; u += NT) {
            const int d = u % DW, h = (u / DW) & 1, w2 = (u / (2 * DW)) % WM, st = u / XS;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = m0e + w2 * 64 + (st >> 1) * 32 + (e & 3) + 8 * (e >> 2) + 16 * (st & 1) + 4 * h;
                const float xv = (d < DIN) ? ep.x[(size_t)min(row, M - 1) * DIN + d] : (d == DIN ? 1.0f : 0.0f);
                v[e] = row < M ? xv : 0.0f;
            }
            uint4 p1, p2, p3;
            split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
            xp[(0 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p1;
            if (kPlanes == 3) {
                xp[(1 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p2;
                xp[(2 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p3;
            }
        }
        __syncthreads();
        const uint4* xp_lane = xp + (wm * 2 + khalf) * DW + min(l31, RW);
        float* mine = red + ((size_t)wm * BN + wn * 128 + l31) * RW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 xq[2][3];
            {
                float x1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x1[e] = 0.0f;
#pragma unroll
                for (int e = 0; e < DIN - 16; ++e) x1[e] = khalf == 0 ? xr1[i][e] : 0.0f;
                x1[DIN - 16] = khalf == 0 ? 1.0f : 0.0f;
                uint4 q1, q2, q3;
                split8(make_float4(xr0[i][0], xr0[i][1], xr0[i][2], xr0[i][3]), make_float4(xr0[i][4], xr0[i][5], xr0[i][6], xr0[i][7]),
                       q1, q2, q3);
                xq[0][0] = *reinterpret_cast<const bf16x8*>(&q1);
                xq[0][1] = *reinterpret_cast<const bf16x8*>(&q2);
                xq[0][2] = *reinterpret_cast<const bf16x8*>(&q3);
                split8(make_float4(x1[0], x1[1], x1[2], x1[3]), make_float4(x1[4], x1[5], x1[6], x1[7]), q1, q2, q3);
                xq[1][0] = *reinterpret_cast<const bf16x8*>(&q1);
                xq[1][1] = *reinterpret_cast<const bf16x8*>(&q2);
                xq[1][2] = *reinterpret_cast<const bf16x8*>(&q3);
            }
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const int blk = wn * 4 + J;                 // feature block of this column tile
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf16x8 wb[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint4 u_ = w1s[(((blk * 2 + st) * 3 + p) * 2 + khalf) * 32 + l31];
                        wb[p] = *reinterpret_cast<const bf16x8*>(&u_);
                    }
                    AG_MFMA_SPLIT(z, xq[st][0], xq[st][1], xq[st][2], wb[0], wb[1], wb[2]);
                }
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float dz[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float zz = z[8 * ks + e];
                        dz[e] = acc[i * 4 + J][8 * ks + e] * (zz > 0.0f ? 1.0f : __builtin_amdgcn_exp2f(zz * 1.4426950408889634f));
                    }
                    uint4 b1, b2, b3;                       // rows past M need no masking here: their x rows are zero
                    split8(make_float4(dz[0], dz[1], dz[2], dz[3]), make_float4(dz[4], dz[5], dz[6], dz[7]), b1, b2, b3);
                    const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(&b1), c2 = *reinterpret_cast<const bf16x8*>(&b2),
                                 c3 = *reinterpret_cast<const bf16x8*>(&b3);
                    const int s = 2 * i + ks;
                    const uint4 ua1 = xp_lane[(0 * 4 + s) * XS], ua2 = xp_lane[(1 * 4 + s) * XS], ua3 = xp_lane[(2 * 4 + s) * XS];
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&ua1), a2 = *reinterpret_cast<const bf16x8*>(&ua2),
                                 a3 = *reinterpret_cast<const bf16x8*>(&ua3);
                    AG_MFMA_SPLIT(g, a1, a2, a3, c1, c2, c3);
                }
                // g: column c = this lane's column of tile J, rows d = (r & 3) + 8 (r >> 2) + 4 h; d = DIN: the bias gradient.  The
                // second row tile adds to the first one's entry (lane-private slot, fixed order: deterministic)
                float* dst = mine + J * 32 * RW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (d < RW) dst[d] = (i == 0) ? g[r] : dst[d] + g[r];
                }
                __builtin_amdgcn_sched_barrier(0);      // one column tile at a time
            }
        }
        __syncthreads();
        for (int idx = tid; idx < BN * RW; idx += NT) {     // the WM row blocks in a fixed order
            const int c = idx / RW, d = idx - c * RW;
            float v = red[idx] + red[BN * RW + idx];
            if (WM == 4) v = (v + red[2 * BN * RW + idx]) + red[3 * BN * RW + idx];
            if (d < DIN) ep.dw_partials[((size_t)tile * BN + c) * DIN + d] = v;
            else ep.db_partials[(size_t)tile * BN + c] = v;
        }
    } else if constexpr (DIN > 0) {
        constexpr int RW = DIN + 1;                         // reduction row: DIN weight-gradient entries + the bias gradient
        constexpr int NDT = (RW + 31) / 32;                 // 32-wide tiles of input columns (1: Hovering's 18; 2: Tracking's 48)
        constexpr int DW = 32 * NDT;
        constexpr int XS = WM * 2 * DW;                     // units per (plane, step)
        uint4* xp = lds;                                    // [plane 3][step 4][wm WM][h 2][d DW] x 16 B: x, split, in K order
        float* red = reinterpret_cast<float*>(lds + 3 * 4 * XS);       // NDT 1: [wm WM][BN][RW]; NDT 2: [wm WM][wn 2][32][RW] per column tile
        int late = 0;                                       // (opaque zero, as above)
        asm volatile("" : "+s"(late) : : "memory");
        const int m0e = m0 + late;          // opaque too: or the row addresses are formed before the main loop and spilled
        // one unit = the 8 rows of (wm, step, lane half h) for one input column d, as three bf16x8 pieces; d = DIN is the
        // all-ones column that yields the bias gradient, d > DIN and rows past M are zero (which also masks the tail tile)
        for (int u = tid; u < 4 * XS; u += NT) {
            const int d = u % DW, h = (u / DW) & 1, w2 = (u / (2 * DW)) % WM, st = u / XS;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = m0e + w2 * 64 + (st >> 1) * 32 + (e & 3) + 8 * (e >> 2) + 16 * (st & 1) + 4 * h;
                const float xv = (d < DIN) ? ep.x[(size_t)min(row, M - 1) * DIN + d] : (d == DIN ? 1.0f : 0.0f);
                v[e] = row < M ? xv : 0.0f;
            }
            uint4 p1, p2, p3;
            split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p1, p2, p3);
            xp[(0 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p1;
            if (kPlanes == 3) {
                xp[(1 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p2;
                xp[(2 * 4 + st) * XS + (w2 * 2 + h) * DW + d] = p3;
            }
        }
        __syncthreads();
        const float* hcol = ep.h1 + wn * 128 + l31 + late;
        const uint4* xp_lane = xp + (wm * 2 + khalf) * DW + l31;
        if constexpr (NDT == 1) {
            float* mine = red + ((size_t)wm * BN + wn * 128 + l31) * RW;
            input_wgrad_tile<0, WM, 1>(acc, hcol, xp_lane, mine, m0e, M, wm, khalf, RW);
            input_wgrad_tile<1, WM, 1>(acc, hcol, xp_lane, mine, m0e, M, wm, khalf, RW);
            input_wgrad_tile<2, WM, 1>(acc, hcol, xp_lane, mine, m0e, M, wm, khalf, RW);
            input_wgrad_tile<3, WM, 1>(acc, hcol, xp_lane, mine, m0e, M, wm, khalf, RW);
            __syncthreads();
            // the WM row blocks in a fixed order: deterministic
            for (int idx = tid; idx < BN * RW; idx += NT) {
                const int c = idx / RW, d = idx - c * RW;
                float v = red[idx] + red[BN * RW + idx];
                if (WM == 4) v = (v + red[2 * BN * RW + idx]) + red[3 * BN * RW + idx];
                if (d < DIN) ep.dw_partials[((size_t)tile * BN + c) * DIN + d] = v;
                else ep.db_partials[(size_t)tile * BN + c] = v;
            }
        } else {
            // two tiles of input columns: the whole-tile reduction buffer would not fit beside the x image, so the column
            // tiles are reduced one at a time through a [wm][2 x 32 columns][RW] buffer (same fixed order over the row blocks)
            float* mine = red + ((size_t)wm * 64 + wn * 32 + l31) * RW;
#define AG_IW_COLUMN_TILE(J_)                                                                          \
            do {                                                                                       \
                input_wgrad_tile<J_, WM, NDT>(acc, hcol, xp_lane, mine, m0e, M, wm, khalf, RW);        \
                __syncthreads();                                                                       \
                for (int idx = tid; idx < 64 * RW; idx += NT) {                                        \
                    const int cj = idx / RW, d = idx - cj * RW;                                        \
                    const int c = (cj >> 5) * 128 + (J_) * 32 + (cj & 31);                              \
                    float v = red[idx] + red[64 * RW + idx];                                           \
                    if (WM == 4) v = (v + red[2 * 64 * RW + idx]) + red[3 * 64 * RW + idx];            \
                    if (d < DIN) ep.dw_partials[((size_t)tile * BN + c) * DIN + d] = v;                \
                    else ep.db_partials[(size_t)tile * BN + c] = v;                                    \
                }                                                                                      \
                __syncthreads();                                                                       \
            } while (0)
            AG_IW_COLUMN_TILE(0);
            AG_IW_COLUMN_TILE(1);
            AG_IW_COLUMN_TILE(2);
            AG_IW_COLUMN_TILE(3);
#undef AG_IW_COLUMN_TILE
        }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int col = wn * 128 + (t & 3) * 32 + l31;
        const int row0 = m0 + wm * 64 + (t >> 2) * 32;
        const float bj = HAS_BIAS ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (row < M) C[(size_t)row * BN + col] = acc[t][r] + bj;
        }
    }
    }
    if constexpr (!PERSIST) break;      // one tile per workgroup: no back edge, the body compiles as before
    }      // persistent tile loop
}

}  // namespace

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" long long ag_split_gemm_plane_bytes(void) { return (long long)(KDIM / BK) * B_UNITS * 16; }
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_prepare(const float* W_dev, void* planes_dev, int n, int k, int transpose, void* stream) {
    if (!W_dev || !planes_dev) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM) return AG_ERR_UNSUPPORTED;
    if ((uintptr_t)planes_dev & 15) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(split_prepare_kernel, dim3(16 * 2 * BN / 256), dim3(256), 0, (hipStream_t)stream, W_dev,
                       (uint4*)planes_dev, transpose, (uint4*)nullptr);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_prepare_pair(const float* W_dev, void* planes_dev, void* planes_t_dev, int n, int k, void* stream) {
    if (!W_dev || !planes_dev || !planes_t_dev) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)planes_dev | (uintptr_t)planes_t_dev) & 15) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(split_prepare_kernel, dim3(2 * 16 * 2 * BN / 256), dim3(256), 0, (hipStream_t)stream, W_dev,
                       (uint4*)planes_dev, 0, (uint4*)planes_t_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
#endif

// Row-tile size of the launches: WM = 2 (128 rows, 4 waves, two workgroups per CU) or 4 (256 rows, 8 waves, one per CU).
// Shipped: 4 - every B-plane chunk fetched from L2 feeds twice the MFMAs and a thread copies 3 instead of 6 plane units per
// chunk; in situ (bench.py, same box, interleaved runs) 31.4-31.6 vs 32.2-32.3 ms per epoch (profiles/r03_split_gemm.md).
#define AG_SPLIT_DEFAULT_WM 4
constexpr int kDefaultWM = AG_SPLIT_DEFAULT_WM;
#ifdef AG_EXPERIMENTS
#include <stdlib.h>
static int g_split_wm = [] { const char* e = getenv("AIRGYM_SPLIT_WM"); return (e && atoi(e) == 4) ? 4 : ((e && atoi(e) == 2) ? 2 : kDefaultWM); }();
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_debug_split_gemm_variant(int wm) {      // 2 | 4; -1 = default
    if (wm != -1 && wm != 2 && wm != 4) return AG_ERR_INVALID_ARG;
    g_split_wm = wm < 0 ? kDefaultWM : wm;
    return AG_OK;
}
#endif
#else
constexpr int g_split_wm = kDefaultWM;
#endif

template <int DIN, int WM>
constexpr size_t split_lds_bytes() {
    size_t stages = (size_t)2 * stage_units(WM * 64) * 16;
    // the first-layer-backward epilogue re-uses the stages for the split x image and the per-row-block reduction buffer
    constexpr int ndt = (DIN + 1 + 31) / 32;
    size_t epi = DIN > 0 ? (size_t)3 * 4 * (WM * 2 * 32 * ndt) * 16 + (size_t)WM * (ndt == 1 ? BN : 64) * (DIN + 1) * 4 : 0;
    return stages > epi ? stages : epi;
}

template <bool HAS_BIAS, int A1, int DIN, int WM, bool LOSS = false, int FIN = 0, bool RC = false>
static int launch_split_any(const float* A_dev, const void* planes_dev, float* C_dev, int M, const SplitEpilogue& ep, void* stream) {
    static bool attr_set[64] = {};      // per device ordinal: the dynamic-LDS limit is an attribute of (function, device)
    auto* fn = split_gemm_kernel<HAS_BIAS, A1, DIN, WM, LOSS, FIN, RC>;
    constexpr size_t lds_bytes = (DIN > 0 && RC) ? (size_t)rc_w1_offset_units<DIN, WM>() * 16 + (size_t)kInImageW1Bytes
                                                 : split_lds_bytes<DIN, WM>() + (FIN > 0 ? (size_t)kInImageW1Bytes : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS per workgroup");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AG_ERR_HIP;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return AG_ERR_HIP;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    constexpr int BM = WM * 64;
    const int tiles = (M + BM - 1) / BM;
    // persistent: as many workgroups as are resident at once (LDS decides: one per CU above 80 KB, else two), each walks its tiles
    int grid = tiles;
    if constexpr (split_persistent<LOSS, DIN>()) {
        static int cus[64];
        if (dev >= 0 && dev < 64) {
            if (cus[dev] == 0) {
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                cus[dev] = n;
            }
            const int resident = cus[dev] * (lds_bytes <= 80 * 1024 ? 2 : 1);
            if (grid > resident) grid = resident;
        }
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(WM * 128), lds_bytes, (hipStream_t)stream, A_dev, (const uint4*)planes_dev, C_dev, M, ep);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

// Small minibatches (round 5): with fewer 256-row tiles than CUs half the chip idles (M = 32 768, the reference's minibatch ratio
// at 65 536 envs: 128 tiles on 256 CUs), so the caller may ask for 128-row tiles (4 waves per workgroup) per launch - `tile_rows` of
// ag_loss_epilogue / ag_split_gemm_input_wgrad_recompute; ag_split_gemm_pick_tile_rows(M) is the rule the agent uses.
#if AG_SPLIT_PLANES == 3
extern "C" int ag_split_gemm_pick_tile_rows(int M) {
    if (M <= 0) return 256;
    static int cus_of[64] = {0};          // per device ordinal, as launch_split_any caches it
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus_of[dev] == 0)
        cus_of[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    const int cus = cus_of[dev];
    return ((M + 255) / 256 < cus && M % 128 == 0) ? 128 : 256;
}
#endif

// dispatch on the runtime tile choice (a constant in the shipped build: only that instantiation is compiled)
#ifdef AG_EXPERIMENTS
#define AG_SG_DISPATCH(call2, call4) (g_split_wm == 4 ? (call4) : (call2))
#elif AG_SPLIT_DEFAULT_WM == 4
#define AG_SG_DISPATCH(call2, call4) (call4)
#else
#define AG_SG_DISPATCH(call2, call4) (call2)
#endif

extern "C" int AG_PREC(ag_split_gemm_elu_heads)(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                                       const float* bh_dev, float* Z_dev, float* heads_dev, int M, int n, int k, int A1,
                                       void* stream) {
    if (!A_dev || !planes_dev || !bias_dev || !Wh_dev || !bh_dev || !Z_dev || !heads_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM || (A1 != 5 && A1 != 6)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)planes_dev & 15)) return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.bias = bias_dev; ep.Wh = Wh_dev; ep.bh = bh_dev; ep.heads = heads_dev;
#define AG_SGH(A, W) launch_split_any<true, A, 0, W>(A_dev, planes_dev, Z_dev, M, ep, stream)
    if (A1 == 5) return AG_SG_DISPATCH(AG_SGH(5, 2), AG_SGH(5, 4));
    return AG_SG_DISPATCH(AG_SGH(6, 2), AG_SGH(6, 4));
#undef AG_SGH
}

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_loss_rows(void) { return g_split_wm * 64; }
#endif

extern "C" int AG_PREC(ag_split_gemm_loss_heads_bwd)(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                                            const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* L, int M, int n, int k, int A1,
                                            void* stream) {
    if (!A_dev || !planes_dev || !bias_dev || !Wh_dev || !bh_dev || !dZ_dev || !L || M <= 0) return AG_ERR_INVALID_ARG;
    if (L->struct_size != sizeof(ag_loss_epilogue)) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM || A1 != 5) return AG_ERR_UNSUPPORTED;      // four actions + the value head
    if (L->tile_rows != 0 && L->tile_rows != 128 && L->tile_rows != 256) return AG_ERR_INVALID_ARG;
    const int rows_ = L->tile_rows == 128 ? 128 : (L->tile_rows == 256 ? 256 : g_split_wm * 64);
    if (M % rows_ != 0) return AG_ERR_UNSUPPORTED;                        // whole row tiles only (the dZ stores are unguarded)
    if (L->partial_tiles < 0 || (L->partial_tiles > 0 && M / rows_ > L->partial_tiles)) return AG_ERR_INVALID_ARG;
    if (!L->logstd_dev || !L->actions_dev || !L->old_neglogp_dev || !L->advantages_dev || !L->returns_dev || !L->old_values_dev ||
        !L->old_mu_dev || !L->old_sigma_dev || !L->loss_partials_dev || !L->dwh_partials_dev || !L->db_partials_dev)
        return AG_ERR_INVALID_ARG;
    if ((L->new_mu_dev == nullptr) != (L->new_sigma_dev == nullptr)) return AG_ERR_INVALID_ARG;
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)planes_dev & 15)) return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.bias = bias_dev; ep.Wh = Wh_dev; ep.bh = bh_dev; ep.heads = L->heads_dev;
    ep.logstd = L->logstd_dev; ep.actions = L->actions_dev; ep.old_neglogp = L->old_neglogp_dev; ep.advantages = L->advantages_dev;
    ep.returns = L->returns_dev; ep.old_values = L->old_values_dev; ep.old_mu = L->old_mu_dev; ep.old_sigma = L->old_sigma_dev;
    ep.new_mu = L->new_mu_dev; ep.new_sigma = L->new_sigma_dev; ep.loss_partials = L->loss_partials_dev;
    ep.dwh_partials = L->dwh_partials_dev; ep.db2_partials = L->db_partials_dev;
    ep.lp = agloss::LossParams{L->e_clip, L->critic_coef, L->bounds_loss_coef, 1.0f / (float)M, L->clip_value, L->bound_type};
#define AG_SGL(W) launch_split_any<true, 5, 0, W, true>(A_dev, planes_dev, dZ_dev, M, ep, stream)
    return rows_ == 128 ? AG_SGL(2) : AG_SGL(4);
#undef AG_SGL
}

// widths with the first layer formed inside the forward GEMM (Hovering's 18 and the neighbouring even widths); 256-row tiles only
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_fwd_supported(int D) { return (g_split_wm == 4 && (D == 16 || D == 18 || D == 20)) ? 1 : 0; }
#endif

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" long long ag_split_gemm_input_image_bytes(void) { return (long long)kInImageW1Bytes + (long long)(KDIM / BK) * B_UNITS * 16; }
#endif

static int launch_in_prepare(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev, void* planes_t_dev,
                             void* stream) {
    if (!W1_dev || !b1_dev || !W2_dev || !image_dev) return AG_ERR_INVALID_ARG;
    if (!ag_split_gemm_input_fwd_supported(D)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)image_dev | (uintptr_t)planes_t_dev) & 15) return AG_ERR_INVALID_ARG;
    const int threads = 8 * 2 * 2 * 32 + (planes_t_dev ? 2 : 1) * 16 * 2 * BN;
    hipLaunchKernelGGL(split_in_prepare_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1_dev, b1_dev, D, W2_dev,
                       (uint4*)image_dev, (uint4*)planes_t_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_prepare(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev,
                                           void* stream) {
    return launch_in_prepare(W1_dev, b1_dev, D, W2_dev, image_dev, nullptr, stream);
}
#endif

// ... and, in the same launch, the backward planes of W2 (what ag_split_gemm_prepare(transpose = 1) writes): one weight-image launch per
// optimizer step instead of two
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_prepare_pair(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev,
                                                void* planes_t_dev, void* stream) {
    if (!planes_t_dev) return AG_ERR_INVALID_ARG;
    return launch_in_prepare(W1_dev, b1_dev, D, W2_dev, image_dev, planes_t_dev, stream);
}
#endif

extern "C" int AG_PREC(ag_split_gemm_input_loss_heads_bwd)(const ag_input_layer_args* in, const void* image_dev, const float* bias_dev,
                                                  const float* Wh_dev, const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* L,
                                                  int M, int n, int k, int A1, void* stream) {
    if (!in || !image_dev || !bias_dev || !Wh_dev || !bh_dev || !dZ_dev || !L || M <= 0) return AG_ERR_INVALID_ARG;
    if (in->struct_size != sizeof(ag_input_layer_args) || L->struct_size != sizeof(ag_loss_epilogue)) return AG_ERR_INVALID_ARG;
    if (!in->obs_dev) return AG_ERR_INVALID_ARG;          // h1_dev NULL: h1 is not written (the backward recomputes it)
    const bool norm = in->mean_dev != nullptr;
    if (norm != (in->var_dev != nullptr) || norm != (in->xn_dev != nullptr)) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM || A1 != 5 || !ag_split_gemm_input_fwd_supported(in->D)) return AG_ERR_UNSUPPORTED;
    if (L->tile_rows != 0 && L->tile_rows != 128 && L->tile_rows != 256) return AG_ERR_INVALID_ARG;
    const bool small_ = L->tile_rows == 128;
    if (M % (small_ ? 128 : 256) != 0) return AG_ERR_UNSUPPORTED;         // whole row tiles only
    if (L->partial_tiles < 0 || (L->partial_tiles > 0 && M / (small_ ? 128 : 256) > L->partial_tiles)) return AG_ERR_INVALID_ARG;
    if (!L->logstd_dev || !L->actions_dev || !L->old_neglogp_dev || !L->advantages_dev || !L->returns_dev || !L->old_values_dev ||
        !L->old_mu_dev || !L->old_sigma_dev || !L->loss_partials_dev || !L->dwh_partials_dev || !L->db_partials_dev)
        return AG_ERR_INVALID_ARG;
    if ((L->new_mu_dev == nullptr) != (L->new_sigma_dev == nullptr)) return AG_ERR_INVALID_ARG;
    if (((uintptr_t)in->obs_dev & 7) || ((uintptr_t)in->h1_dev & 15) || ((uintptr_t)image_dev & 15)) return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.bias = bias_dev; ep.Wh = Wh_dev; ep.bh = bh_dev; ep.heads = L->heads_dev;
    ep.logstd = L->logstd_dev; ep.actions = L->actions_dev; ep.old_neglogp = L->old_neglogp_dev; ep.advantages = L->advantages_dev;
    ep.returns = L->returns_dev; ep.old_values = L->old_values_dev; ep.old_mu = L->old_mu_dev; ep.old_sigma = L->old_sigma_dev;
    ep.new_mu = L->new_mu_dev; ep.new_sigma = L->new_sigma_dev; ep.loss_partials = L->loss_partials_dev;
    ep.dwh_partials = L->dwh_partials_dev; ep.db2_partials = L->db_partials_dev;
    ep.lp = agloss::LossParams{L->e_clip, L->critic_coef, L->bounds_loss_coef, 1.0f / (float)M, L->clip_value, L->bound_type};
    ep.in_mean = in->mean_dev; ep.in_var = in->var_dev; ep.w1img = reinterpret_cast<const uint4*>(image_dev); ep.xn_out = in->xn_dev;
    ep.h1_out = in->h1_dev; ep.in_eps = in->eps; ep.in_clip = in->clip;
    const void* planes_dev = reinterpret_cast<const char*>(image_dev) + kInImageW1Bytes;
#define AG_SGF_(F, W) (in->h1_dev ? launch_split_any<true, 5, 0, W, true, F, false>(in->obs_dev, planes_dev, dZ_dev, M, ep, stream) \
                                  : launch_split_any<true, 5, 0, W, true, F, true>(in->obs_dev, planes_dev, dZ_dev, M, ep, stream))
#define AG_SGF(F) (small_ ? AG_SGF_(F, 2) : AG_SGF_(F, 4))
    switch (in->D) {
        case 16: return AG_SGF(16);
        case 18: return AG_SGF(18);
        default: return AG_SGF(20);
    }
#undef AG_SGF
#undef AG_SGF_
}

#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_wgrad_rows(void) { return g_split_wm * 64; }
#endif

// input widths with a fused first-layer backward: Hovering 18 (16 / 20: the neighbouring even widths), Tracking 48
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_wgrad_supported(int D) { return (D == 16 || D == 18 || D == 20 || D == 48) ? 1 : 0; }
#endif

extern "C" int AG_PREC(ag_split_gemm_input_wgrad)(const float* dZ_dev, const void* planes_dev, const float* h1_dev, const float* x_dev,
                                         float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D, void* stream) {
    if (!dZ_dev || !planes_dev || !h1_dev || !x_dev || !dw_partials_dev || !db_partials_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM || !ag_split_gemm_input_wgrad_supported(D)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)dZ_dev & 15) || ((uintptr_t)planes_dev & 15)) return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.h1 = h1_dev; ep.x = x_dev; ep.dw_partials = dw_partials_dev; ep.db_partials = db_partials_dev;
#define AG_SGI(DV, W) launch_split_any<false, 0, DV, W>(dZ_dev, planes_dev, nullptr, M, ep, stream)
    switch (D) {
        case 16: return AG_SG_DISPATCH(AG_SGI(16, 2), AG_SGI(16, 4));
        case 18: return AG_SG_DISPATCH(AG_SGI(18, 2), AG_SGI(18, 4));
        case 20: return AG_SG_DISPATCH(AG_SGI(20, 2), AG_SGI(20, 4));
        default: return AG_SG_DISPATCH(AG_SGI(48, 2), AG_SGI(48, 4));
    }
#undef AG_SGI
}

// widths whose first-layer backward can recompute h1 in the dX epilogue (256-row tiles; the LDS layout holds D + 2 <= 20 columns)
#if AG_SPLIT_PLANES == 3      // (exists once: the one-plane build calls the three-plane build's)
extern "C" int ag_split_gemm_input_wgrad_recompute_supported(int D) { return (g_split_wm == 4 && (D == 16 || D == 18)) ? 1 : 0; }
#endif

extern "C" int AG_PREC(ag_split_gemm_input_wgrad_recompute)(const float* dZ_dev, const void* planes_dev, const void* image_dev, const float* x_dev,
                                                   float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D,
                                                   int tile_rows, void* stream) {
    if (!dZ_dev || !planes_dev || !image_dev || !x_dev || !dw_partials_dev || !db_partials_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM || !ag_split_gemm_input_wgrad_recompute_supported(D)) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)dZ_dev & 15) || ((uintptr_t)planes_dev & 15) || ((uintptr_t)image_dev & 15) || ((uintptr_t)x_dev & 7))
        return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.x = x_dev; ep.dw_partials = dw_partials_dev; ep.db_partials = db_partials_dev;
    ep.w1img = reinterpret_cast<const uint4*>(image_dev);
    if (tile_rows != 0 && tile_rows != 128 && tile_rows != 256) return AG_ERR_INVALID_ARG;
    if (tile_rows == 128) {      // partials: one per 128 rows
        if (D == 16) return launch_split_any<false, 0, 16, 2, false, 0, true>(dZ_dev, planes_dev, nullptr, M, ep, stream);
        return launch_split_any<false, 0, 18, 2, false, 0, true>(dZ_dev, planes_dev, nullptr, M, ep, stream);
    }
    if (D == 16) return launch_split_any<false, 0, 16, 4, false, 0, true>(dZ_dev, planes_dev, nullptr, M, ep, stream);
    return launch_split_any<false, 0, 18, 4, false, 0, true>(dZ_dev, planes_dev, nullptr, M, ep, stream);
}

extern "C" int AG_PREC(ag_split_gemm)(const float* A_dev, const void* planes_dev, const float* bias_dev, float* C_dev, int M, int n, int k,
                             void* stream) {
    if (!A_dev || !planes_dev || !C_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)planes_dev & 15)) return AG_ERR_INVALID_ARG;
    SplitEpilogue ep = {};
    ep.bias = bias_dev;
#define AG_SGP(HB, W) launch_split_any<HB, 0, 0, W>(A_dev, planes_dev, C_dev, M, ep, stream)
    if (bias_dev) return AG_SG_DISPATCH(AG_SGP(true, 2), AG_SGP(true, 4));
    return AG_SG_DISPATCH(AG_SGP(false, 2), AG_SGP(false, 4));
#undef AG_SGP
}
