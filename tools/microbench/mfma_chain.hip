// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 when consecutive instructions write the SAME accumulator (a dependent
// chain) against round-robin over NACC independent accumulators - one wave per SIMD (256 threads, 1 block per CU) or two (512).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libmfma_chain.so tools/microbench/mfma_chain.hip
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void mfma_chain_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.0f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int mfma_chain_launch(int nacc, int threads, int blocks, int iters, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (nacc) {
        case 1: hipLaunchKernelGGL(mfma_chain_kernel<1>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL(mfma_chain_kernel<2>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
        case 3: hipLaunchKernelGGL(mfma_chain_kernel<3>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
        case 4: hipLaunchKernelGGL(mfma_chain_kernel<4>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
        case 6: hipLaunchKernelGGL(mfma_chain_kernel<6>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
