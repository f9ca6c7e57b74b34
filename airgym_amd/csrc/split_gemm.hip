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
//   workgroup 256 threads = 4 waves; block tile 128 rows x 256 columns (all of N); wave w owns rows 32w..32w+31 as eight
//   32x32 tiles (v_mfma_f32_32x32x16_bf16, 128 accumulator registers); K in chunks of 16, double-buffered in LDS:
//     A stage: [plane 3][k-half 2][row 128] x 16 B   (12 KB)   - the wave splits its f32 rows on the fly
//     B stage: [plane 3][k-half 2][col 256] x 16 B   (24 KB)
//   72 KB per workgroup -> two workgroups per CU, two waves per SIMD: one wave's LDS traffic and f32->bf16 splitting hide
//   under the other's MFMAs.  Fragment reads are ds_read_b128 over 32 consecutive 16-byte units: conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 16, KDIM = 256;
constexpr int A_UNITS = 3 * 2 * BM;       // 16-byte units per A stage
constexpr int B_UNITS = 3 * 2 * BN;       // 16-byte units per B stage
constexpr int STAGE_UNITS = A_UNITS + B_UNITS;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// (x, y) -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (x in the low half)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float x, float y) {
    const f32x2_t v = {x, y};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&b);
}

// Exact 3-way split of two consecutive-k floats into bf16 pieces, one packed word per plane.  a1 = rn_bf16(a), a2 =
// rn_bf16(a - a1), a3 = a - a1 - a2: both differences are exact in f32 and the last one has at most 8 significant bits, so
// a == a1 + a2 + a3 with |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a| (round to nearest; truncation would give 2^-7 / 2^-15).
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    w1 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(w1 << 16), ry = y - __uint_as_float(w1 & 0xFFFF0000u);
    w2 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(w2 << 16), sy = ry - __uint_as_float(w2 & 0xFFFF0000u);
    w3 = cvt_pk_bf16(sx, sy);
}

// 8 consecutive-k floats -> three 16-byte bf16x8 units (one per plane)
__device__ __forceinline__ void split8(const float4 lo, const float4 hi, uint4& p1, uint4& p2, uint4& p3) {
    split_pair(lo.x, lo.y, p1.x, p2.x, p3.x);
    split_pair(lo.z, lo.w, p1.y, p2.y, p3.y);
    split_pair(hi.x, hi.y, p1.z, p2.z, p3.z);
    split_pair(hi.z, hi.w, p1.w, p2.w, p3.w);
}

// Weight preparation: W [256, 256] f32 (row-major) -> planes [chunk 16][plane 3][k-half 2][n 256] x 8 bf16, the per-chunk LDS
// image of the main loop.  transpose = 0: B[n][k] = W[n][k] (forward, X W^T); 1: B[n][k] = W[k][n] (backward dX = dZ W).
__global__ __launch_bounds__(256) void split_prepare_kernel(const float* __restrict__ W, uint4* __restrict__ planes, int transpose) {
    const int unit = blockIdx.x * 256 + threadIdx.x;          // one (chunk, k-half, n) unit per thread: 16 * 2 * 256 = 8192
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

// VAR bit 0: pin the next chunk's global loads to the TOP of the chunk (a whole chunk of MFMAs for them to land) instead of
// letting the scheduler sink them to shorten live ranges.  VAR bit 1: waves as 2 x 2 (64 rows x 128 columns each) instead of
// 4 x 1 (32 rows x 256 columns): fewer LDS fragment reads.
template <bool HAS_BIAS, int VAR>
__global__ __launch_bounds__(256, 2) void split_gemm_kernel(const float* __restrict__ A, const uint4* __restrict__ Bp,
                                                             const float* __restrict__ bias, float* __restrict__ C, int M) {
    extern __shared__ uint4 lds[];                         // [2 stages][A_UNITS + B_UNITS] 16-byte units
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int num_tiles = (M + BM - 1) / BM;
    if (VAR & 4) {
        // persistent workgroups, two per CU.  All workgroups run identical tiles, so without help the two on a CU stay in
        // lockstep and their epilogues (128 KB of C stores each) stall the matrix pipe together.  The second workgroup of
        // a CU (non-zero LDS base, HW_REG_LDS_ALLOC) starts half a tile late: from then on one stores while the other
        // multiplies.
        const unsigned lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);
        if ((lds_alloc & 0xFF) != 0) {
#pragma unroll 1
            for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ((VAR & 4) ? num_tiles : (int)blockIdx.x + 1); tile += (VAR & 4) ? (int)gridDim.x : num_tiles) {
    const int m0 = tile * BM;
    if ((VAR & 4) && tile != (int)blockIdx.x) __syncthreads();      // the previous tile's epilogue is done before LDS is reused

    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // ---- global -> register staging for one K chunk
    const int a_row = tid >> 1, a_half = tid & 1;                       // 128 rows x 2 k-halves: one 32-byte piece per thread
    const int a_grow = min(m0 + a_row, M - 1);                          // rows past M are computed, never stored
    const float4* a_src = reinterpret_cast<const float4*>(A + (size_t)a_grow * KDIM + a_half * 8);
    float4 ra0, ra1;
    uint4 rb0, rb1, rb2, rb3, rb4, rb5;
#define AG_SG_LOAD(c)                                                                  \
    do {                                                                               \
        ra0 = a_src[(c) * (BK / 4)];                                                   \
        ra1 = a_src[(c) * (BK / 4) + 1];                                               \
        const uint4* bsrc_ = Bp + (size_t)(c) * B_UNITS + tid;                         \
        rb0 = bsrc_[0]; rb1 = bsrc_[256]; rb2 = bsrc_[512];                            \
        rb3 = bsrc_[768]; rb4 = bsrc_[1024]; rb5 = bsrc_[1280];                        \
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
        sb_[0] = rb0; sb_[256] = rb1; sb_[512] = rb2;                                  \
        sb_[768] = rb3; sb_[1024] = rb4; sb_[1280] = rb5;                              \
    } while (0)

    AG_SG_LOAD(0);
    AG_SG_STORE(0);
    __syncthreads();

    const int l31 = lane & 31, khalf = lane >> 5;
    constexpr int NCHUNK = KDIM / BK;
    for (int c = 0; c < NCHUNK; ++c) {
        const int stage = c & 1;
        if (c + 1 < NCHUNK) AG_SG_LOAD(c + 1);              // global loads in flight under this chunk's MFMAs
        if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
        const uint4* sa = lds + stage * STAGE_UNITS;
        const uint4* sb = sa + A_UNITS;
        if (VAR & 2) {
            // 2 x 2 waves: wave (wm, wn) owns rows 64 wm .. +63 and columns 128 wn .. +127 (2 x 4 tiles): 18 fragment reads per
            // chunk instead of 27 (every B fragment feeds two row tiles)
            const int wm = wave >> 1, wn = wave & 1;
            bf16x8 a[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const uint4 u = sa[(p * 2 + khalf) * BM + wm * 64 + i * 32 + l31];
                    a[i][p] = *reinterpret_cast<const bf16x8*>(&u);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 ub0 = sb[(0 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];
                const uint4 ub1 = sb[(1 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];
                const uint4 ub2 = sb[(2 * 2 + khalf) * BN + wn * 128 + j * 32 + l31];
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&ub0);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&ub1);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&ub2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x16& d = acc[i * 4 + j];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b0, d, 0, 0, 0);
                }
            }
        } else {
        const uint4 ua0 = sa[(0 * 2 + khalf) * BM + wave * 32 + l31];
        const uint4 ua1 = sa[(1 * 2 + khalf) * BM + wave * 32 + l31];
        const uint4 ua2 = sa[(2 * 2 + khalf) * BM + wave * 32 + l31];
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&ua0);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&ua1);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&ua2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 ub0 = sb[(0 * 2 + khalf) * BN + j * 32 + l31];
            const uint4 ub1 = sb[(1 * 2 + khalf) * BN + j * 32 + l31];
            const uint4 ub2 = sb[(2 * 2 + khalf) * BN + j * 32 + l31];
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&ub0);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&ub1);
            const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&ub2);
            // smallest terms first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
        }
        }
        if (c + 1 < NCHUNK) AG_SG_STORE(stage ^ 1);         // the other stage was last read before the previous barrier
        __syncthreads();
    }
#undef AG_SG_LOAD
#undef AG_SG_STORE

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int col = (VAR & 2) ? ((wave & 1) * 128 + (t & 3) * 32 + l31) : (t * 32 + l31);
        const int row0 = (VAR & 2) ? (m0 + (wave >> 1) * 64 + (t >> 2) * 32) : (m0 + wave * 32);
        const float bj = HAS_BIAS ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (row < M) C[(size_t)row * BN + col] = acc[t][r] + bj;
        }
    }
    }   // tile loop
}

}  // namespace

extern "C" long long ag_split_gemm_plane_bytes(void) { return (long long)(KDIM / BK) * B_UNITS * 16; }

extern "C" int ag_split_gemm_prepare(const float* W_dev, void* planes_dev, int n, int k, int transpose, void* stream) {
    if (!W_dev || !planes_dev) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM) return AG_ERR_UNSUPPORTED;
    if ((uintptr_t)planes_dev & 15) return AG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(split_prepare_kernel, dim3(16 * 2 * BN / 256), dim3(256), 0, (hipStream_t)stream, W_dev,
                       (uint4*)planes_dev, transpose);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

static int g_split_variant = -1;      // -1 = pick by size (measured on MI355X, profiles/r02_split_gemm.md)
extern "C" int ag_debug_split_gemm_variant(int variant) {
    if (variant < -1 || variant > 7) return AG_ERR_INVALID_ARG;
    g_split_variant = variant;
    return AG_OK;
}

template <int VAR>
static int launch_split(const float* A_dev, const void* planes_dev, const float* bias_dev, float* C_dev, int M, void* stream) {
    static bool attr_set = false;
    const size_t lds = (size_t)2 * STAGE_UNITS * 16;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<false, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<true, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return AG_ERR_HIP;
        attr_set = true;
    }
    int tiles = (M + BM - 1) / BM;
    if (VAR & 4) tiles = tiles < 512 ? tiles : 512;        // persistent: 2 workgroups on each of the 256 CUs
    const dim3 grid(tiles), block(256);
    if (bias_dev) hipLaunchKernelGGL((split_gemm_kernel<true, VAR>), grid, block, lds, (hipStream_t)stream, A_dev, (const uint4*)planes_dev, bias_dev, C_dev, M);
    else hipLaunchKernelGGL((split_gemm_kernel<false, VAR>), grid, block, lds, (hipStream_t)stream, A_dev, (const uint4*)planes_dev, bias_dev, C_dev, M);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}

extern "C" int ag_split_gemm(const float* A_dev, const void* planes_dev, const float* bias_dev, float* C_dev, int M, int n, int k,
                             void* stream) {
    if (!A_dev || !planes_dev || !C_dev || M <= 0) return AG_ERR_INVALID_ARG;
    if (n != BN || k != KDIM) return AG_ERR_UNSUPPORTED;
    if (((uintptr_t)A_dev & 15) || ((uintptr_t)planes_dev & 15)) return AG_ERR_INVALID_ARG;
    // more than one round of workgroups (> 2 per CU): persistent, de-phased, 2 x 2 waves (6); a single round: plain 2 x 2 (2)
    const int variant = g_split_variant >= 0 ? g_split_variant : (((M + BM - 1) / BM > 512) ? 6 : 2);
    switch (variant) {
        case 0: return launch_split<0>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 1: return launch_split<1>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 2: return launch_split<2>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 3: return launch_split<3>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 4: return launch_split<4>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 6: return launch_split<6>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        case 7: return launch_split<7>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
        default: return launch_split<5>(A_dev, planes_dev, bias_dev, C_dev, M, stream);
    }
}
