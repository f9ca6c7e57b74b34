// split_common.hpp - the exact three-way bf16 split of float32 values shared by the split GEMMs (split_gemm.hip: forward and
// dX products; split_wgrad.hip: the weight gradient).  See split_gemm.hip for the error analysis.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// AG_SPLIT_PLANES = 3 (default): the exact 3-way split, six bf16 MFMAs per f32 product - float32-accurate.
// AG_SPLIT_PLANES = 1: `mixed_precision: true` (the reference's torch.cuda.amp.autocast switch, lib/agent/a2c_base.py:236-237,566,582):
// ONE bf16 MFMA per product - operands rounded to bf16 (round to nearest), f32 accumulate, f32 master weights; the same sources
// are compiled a second time with this setting and every compute entry point exported with the suffix _bf16 (csrc/build.py).
// Planes 2 and 3 of the prepared weight images are simply not read; activation planes 2 and 3 are neither formed nor stored.
#ifndef AG_SPLIT_PLANES
#define AG_SPLIT_PLANES 3
#endif
#if AG_SPLIT_PLANES != 1 && AG_SPLIT_PLANES != 3
#error "AG_SPLIT_PLANES: 1 or 3"
#endif
#define AG_CAT2_(a, b) a##b
#define AG_CAT2(a, b) AG_CAT2_(a, b)
#if AG_SPLIT_PLANES == 3
#define AG_PREC(name) name                       /* compute entry points: ag_xyz / ag_xyz_bf16 */
#else
#define AG_PREC(name) AG_CAT2(name, _bf16)       /* (size / capability / weight-image exports exist once, in the 3-plane build) */
#endif

namespace {

constexpr int kPlanes = AG_SPLIT_PLANES;

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

}  // namespace

// d += a * b for split operands (planes 1..3 of each): the six kept cross products, smallest first - or, with one plane, a1 b1
#if AG_SPLIT_PLANES == 3
#define AG_MFMA_SPLIT(d, a1, a2, a3, b1, b2, b3)                                  \
    do {                                                                          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, d, 0, 0, 0);          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, d, 0, 0, 0);          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, d, 0, 0, 0);          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, d, 0, 0, 0);          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d, 0, 0, 0);          \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d, 0, 0, 0);          \
    } while (0)
#else
#define AG_MFMA_SPLIT(d, a1, a2, a3, b1, b2, b3)                                  \
    do {                                                                          \
        (void)(a2); (void)(a3); (void)(b2); (void)(b3);                           \
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d, 0, 0, 0);          \
    } while (0)
#endif
