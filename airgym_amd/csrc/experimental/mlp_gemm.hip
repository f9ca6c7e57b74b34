// EXPERIMENTAL - not part of libairgym_hip.so (see README.md in this directory for the measurements and why).
// mlp_gemm.hip - the hidden-layer GEMM of the actor-critic with its consumers fused in (lib/network/mlp.py:36-39,
// lib/model/a2c_continuous_logstd_model.py:130-146):
//
//     Z[M,256]     = X[M,K] W[256,K]^T + b            (pre-activation, kept for the backward pass)
//     heads[M,A1]  = ELU(Z) Wh[A1,256]^T + bh          (mu | value), formed in the epilogue from the accumulators
//
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32, 64 cycles per instruction per SIMD = the f32 vector rate, 157 TFLOP/s
// chip peak).  Because one MFMA occupies the matrix pipe for 64 cycles, a plain LDS-tiled loop with register-staged
// prefetch keeps it fed; there is no need for the 8-phase schedules bf16 GEMMs want.
//
// Tiling: workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); block tile 128 x 256 (ALL output columns, so the head product
// of a row is complete inside one workgroup); wave tile 64 x 128 = 2 x 4 MFMA tiles of 32 x 32 -> 128 accumulator VGPRs;
// BK = 16 per stage, double-buffered in LDS as As[k][row] / Ws[k][col] (fragment reads are 32 consecutive floats per
// half-wave: conflict-free ds_read_b32).  Two workgroups fit a CU (48 KB LDS, < 256 VGPRs), so one wave's epilogue VALU
// work overlaps the other's MFMAs.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../../include/airgym_hip.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 16;
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float elu_fast(float z) { return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

// sum over each 32-lane half of the wave; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);
    v = dpp_add<0x4E, 0xF>(v);
    v = dpp_add<0x141, 0xF>(v);
    v = dpp_add<0x140, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    return v;
}

template <int A1, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bias_heads_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, const float* __restrict__ Wh,
                                                                 const float* __restrict__ bh, float* __restrict__ Z,
                                                                 float* __restrict__ heads, int M, int K) {
    __shared__ float As[2][BK][BM];
    __shared__ float Ws[2][BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int m0 = blockIdx.x * BM;

    v16f acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- global -> register staging: A rows (tid % 128), k-quads {tid/128, tid/128 + 2}; W row tid, k-quads 0..3
    const int a_row = tid & 127, a_q0 = tid >> 7;
    const int a_grow = min(m0 + a_row, M - 1);                  // clamp: rows past M are computed but never stored
    const float4* xa = reinterpret_cast<const float4*>(X + (size_t)a_grow * K);
    const float4* wa = reinterpret_cast<const float4*>(W + (size_t)tid * K);
    float4 ra[2], rw[4];
    auto load_stage = [&](int kc) {
        const int q = kc * (BK / 4);
        ra[0] = xa[q + a_q0];
        ra[1] = xa[q + a_q0 + 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = wa[q + i];
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (a_q0 + 2 * h) * 4;
            As[buf][k + 0][a_row] = ra[h].x; As[buf][k + 1][a_row] = ra[h].y;
            As[buf][k + 2][a_row] = ra[h].z; As[buf][k + 3][a_row] = ra[h].w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Ws[buf][4 * i + 0][tid] = rw[i].x; Ws[buf][4 * i + 1][tid] = rw[i].y;
            Ws[buf][4 * i + 2][tid] = rw[i].z; Ws[buf][4 * i + 3][tid] = rw[i].w;
        }
    };

    const int nk = K / BK;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_stage(kc + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + khalf;
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[buf][k][wm * 64 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Ws[buf][k][wn * 128 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, store Z, head product.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* Hs = &Ws[0][0][0];                       // [2 (wn)][BM][8] partial head sums (reuses the W stage buffers)
    float bcol[4], whc[A1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 128 + j * 32 + l31;
        bcol[j] = bias[col];
#pragma unroll
        for (int a = 0; a < A1; ++a) whc[a][j] = Wh[(size_t)a * BN + col];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row_l = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int row = m0 + row_l;
            float p[A1];
#pragma unroll
            for (int a = 0; a < A1; ++a) p[a] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = acc[i][j][r] + bcol[j];
                if (MODE == 2) {           // diagnostic: one store per accumulator tile only
                    if (r == 0 && row < M) Z[(size_t)row * BN + wn * 128 + j * 32 + l31] = z;
                } else if (row < M) {
                    Z[(size_t)row * BN + wn * 128 + j * 32 + l31] = z;
                }
                if (MODE == 0) {
                    const float hval = elu_fast(z);
#pragma unroll
                    for (int a = 0; a < A1; ++a) p[a] = fmaf(hval, whc[a][j], p[a]);
                }
            }
            if (MODE != 0) continue;
#pragma unroll
            for (int a = 0; a < A1; ++a) p[a] = half_wave_sum(p[a]);
            if (l31 == 31) {
#pragma unroll
                for (int a = 0; a < A1; ++a) Hs[((size_t)wn * BM + row_l) * 8 + a] = p[a];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * A1; idx += 256) {
        const int row_l = idx / A1, a = idx - row_l * A1;
        const int row = m0 + row_l;
        if (row < M) heads[(size_t)row * A1 + a] = (Hs[(size_t)row_l * 8 + a] + Hs[((size_t)BM + row_l) * 8 + a]) + bh[a];
    }
}


// ---------------------------------------------------------------------------------------------------
// Pipelined persistent variant (K = 256, M % 128 == 0): one workgroup per CU walks over row tiles; while the MFMAs of
// tile t run, the epilogue of tile t-1 (bias, Z store, ELU, head product, DPP row sums) is issued in their shadow - an
// f32 MFMA holds the matrix pipe for 64 cycles, ~15 issue slots that would otherwise idle.  Accumulators of two tiles are
// live (256 VGPRs), so one wave per SIMD; only the last tile's epilogue is exposed.
// ---------------------------------------------------------------------------------------------------
template <int A1>
__global__ __launch_bounds__(256, 1) void gemm_heads_pipelined_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                      const float* __restrict__ bias, const float* __restrict__ Wh,
                                                                      const float* __restrict__ bh, float* __restrict__ Z,
                                                                      float* __restrict__ heads, int num_tiles) {
    constexpr int K = 256, NK = K / BK;
    __shared__ float As[2][BK][BM];
    __shared__ float Ws[2][BK][BN];
    __shared__ float Hs[2][BM][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;

    float bcol[4], whc[A1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 128 + j * 32 + l31;
        bcol[j] = bias[col];
#pragma unroll
        for (int a = 0; a < A1; ++a) whc[a][j] = Wh[(size_t)a * BN + col];
    }

    v16f acc[2][4], prev[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[i][j][r] = 0.f;

    const int a_row = tid & 127, a_q0 = tid >> 7;
    float4 ra[2], rw[4];
    const float4* wa = reinterpret_cast<const float4*>(W + (size_t)tid * K);

    // epilogue of ONE accumulator row slot (i, r), column tile j of the pending tile at rows pm0..pm0+127
    float p[A1];
    auto epi = [&](int i, int r, int j, int pm0) {
        const int row_l = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const float z = prev[i][j][r] + bcol[j];
        Z[(size_t)(pm0 + row_l) * BN + wn * 128 + j * 32 + l31] = z;
        const float hval = elu_fast(z);
        if (j == 0) {
#pragma unroll
            for (int a = 0; a < A1; ++a) p[a] = hval * whc[a][0];
        } else {
#pragma unroll
            for (int a = 0; a < A1; ++a) p[a] = fmaf(hval, whc[a][j], p[a]);
        }
        if (j == 3) {
#pragma unroll
            for (int a = 0; a < A1; ++a) p[a] = half_wave_sum(p[a]);
            if (l31 == 31) {
#pragma unroll
                for (int a = 0; a < A1; ++a) Hs[wn][row_l][a] = p[a];
            }
        }
    };
    auto finish_heads = [&](int pm0) {
        for (int idx = tid; idx < BM * A1; idx += 256) {
            const int row_l = idx / A1, a = idx - row_l * A1;
            heads[(size_t)(pm0 + row_l) * A1 + a] = (Hs[0][row_l][a] + Hs[1][row_l][a]) + bh[a];
        }
    };

    int pm0 = blockIdx.x * BM;                 // the first "pending" tile is all zeros: its stores land on this block's own
                                               // first tile and are overwritten by the real epilogue one iteration later
    for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        const float4* xa = reinterpret_cast<const float4*>(X + (size_t)(m0 + a_row) * K);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto load_stage = [&](int kc) {
            const int q = kc * (BK / 4);
            ra[0] = xa[q + a_q0];
            ra[1] = xa[q + a_q0 + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) rw[i] = wa[q + i];
        };
        auto store_stage = [&](int buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = (a_q0 + 2 * h) * 4;
                As[buf][k + 0][a_row] = ra[h].x; As[buf][k + 1][a_row] = ra[h].y;
                As[buf][k + 2][a_row] = ra[h].z; As[buf][k + 3][a_row] = ra[h].w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Ws[buf][4 * i + 0][tid] = rw[i].x; Ws[buf][4 * i + 1][tid] = rw[i].y;
                Ws[buf][4 * i + 2][tid] = rw[i].z; Ws[buf][4 * i + 3][tid] = rw[i].w;
            }
        };
        load_stage(0);
        store_stage(0);
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < NK; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < NK) load_stage(kc + 1);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = 2 * kk + khalf;
                float a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[buf][k][wm * 64 + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Ws[buf][k][wn * 128 + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                // pending tile: k-step (kc, kk) carries row slot s = 2 kc + kk / 4, column tile kk % 4
                const int s2 = 2 * kc + (kk >> 2);
                epi(s2 >> 4, s2 & 15, kk & 3, pm0);
            }
            if (kc + 1 < NK) store_stage(buf ^ 1);
            __syncthreads();
        }
        finish_heads(pm0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) prev[i][j] = acc[i][j];
        pm0 = m0;
    }
    // drain: the last tile's epilogue has no MFMAs to hide behind
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j) epi(s2 >> 4, s2 & 15, j, pm0);
    __syncthreads();
    finish_heads(pm0);
}

}  // namespace

extern "C" int ag_mlp_hidden_heads(const float* X, const float* W, const float* bias, const float* Wh, const float* bh, float* Z,
                                   float* heads, int M, int K, int N, int A1, void* stream) {
    if (!X || !W || !bias || !Wh || !bh || !Z || !heads || M <= 0) return AG_ERR_INVALID_ARG;
    if (N != BN || K <= 0 || (K % BK) != 0) return AG_ERR_UNSUPPORTED;
    const int grid = (M + BM - 1) / BM;
    const char* dbg = getenv("AG_MLP_GEMM_MODE");      // diagnostics only: 1 = no head epilogue, 2 = minimal stores, 3 = plain
    const int mode = dbg ? atoi(dbg) : 0;
    if (mode == 0 && K == 256 && M % BM == 0) {        // pipelined persistent kernel: one workgroup per CU
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int tiles = M / BM;
        const int g = tiles < cus ? tiles : cus;
        if (A1 == 5)
            hipLaunchKernelGGL(gemm_heads_pipelined_kernel<5>, dim3(g), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, tiles);
        else if (A1 == 6)
            hipLaunchKernelGGL(gemm_heads_pipelined_kernel<6>, dim3(g), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, tiles);
        else
            return AG_ERR_UNSUPPORTED;
        return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
    }
    if (A1 == 5 && mode == 1)
        hipLaunchKernelGGL((gemm_bias_heads_kernel<5, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, M, K);
    else if (A1 == 5 && mode == 2)
        hipLaunchKernelGGL((gemm_bias_heads_kernel<5, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, M, K);
    else if (A1 == 5)      // mode 3 or shapes the pipelined kernel does not take
        hipLaunchKernelGGL((gemm_bias_heads_kernel<5, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, M, K);
    else if (A1 == 6)
        hipLaunchKernelGGL((gemm_bias_heads_kernel<6, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, W, bias, Wh, bh, Z, heads, M, K);
    else
        return AG_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
