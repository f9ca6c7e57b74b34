// update_tail.hip - the tail of a single-GPU optimizer step as ONE launch (round 6).
//
// After the three matrix-core launches of a PPO minibatch (a2c_continuous.py:299-369) the step still has to (1) finish the
// partial-sum reductions of every gradient (stage 2 of ag_sum_rows_multi), (2) form the global gradient norm, (3) clip, run Adam and
// the KL-adaptive learning-rate rule (trancate_gradients_and_step, a2c_base.py:293-316; schedulers.py:19-32) and (4) re-split the
// updated weights into the bf16 plane images the next step's GEMM launches read.  As four launches that is 6.1 + 4.6 + 9.2 + 4.7 us
// of kernels that are each too small to fill the chip (profiles/r06_bench_kernel_trace.md) - 25 us of a 560 us step at the headline's
// 196 608-sample minibatches and of a 180 us step at the reference's minibatch ratio (240 steps per epoch).  Here the four phases run
// in one launch of kTailBlocks workgroups separated by grid barriers; every phase executes the SAME device body as the separate
// kernels (tail_parts.hpp), so gradients, Adam state, parameters and weight images are bit-identical to the four-launch sequence
// (tests/test_gpu_update_tail.py).
//
// Grid barrier: a monotonically increasing ticket counter in device memory (caller-owned, zero-initialised once).  Every workgroup
// takes a ticket, the barrier opens when the counter reaches the next multiple of the grid size; kTailBlocks is a power of two, so
// the arithmetic survives the 32-bit wrap.  Release / acquire at agent scope (the 8 XCDs' L2s are not coherent with each other for
// plain accesses: the fence pair writes back / invalidates).  All kTailBlocks workgroups are co-resident by construction: 128
// workgroups of 256 threads with 4 KB of LDS on 256 CUs.  Multi-GPU runs keep the separate launches: their gradient all-reduce
// sits between phases (1) and (2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "tail_parts.hpp"

namespace {

constexpr int kTailBlocks = 128;
static_assert((kTailBlocks & (kTailBlocks - 1)) == 0 && kTailBlocks >= kAdamBlocks, "power of two, one workgroup per norm partial");

__device__ __forceinline__ void grid_barrier(unsigned* counter) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // this workgroup's stores first (L2 write-back across XCDs)
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / kTailBlocks + 1u) * kTailBlocks;
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // ... and nobody reads stale lines behind the barrier
    }
    __syncthreads();
}

struct PrepareArgs {
    const float* W1; const float* b1; int D; const float* W2; uint4* img; uint4* planes_t;
};

__global__ __launch_bounds__(256) void update_tail_kernel(const SumJobs jobs, const int stage2_blocks, const AdamArgs adam,
                                                          float* __restrict__ partial, double* __restrict__ state, const PrepareArgs prep,
                                                          unsigned* __restrict__ barrier) {
    __shared__ float4 red[256];
    // (1) stage 2 of the partial-sum reductions: the final gradient slices
    for (int vb = blockIdx.x; vb < stage2_blocks; vb += kTailBlocks) {
        sum_stage2_block(jobs, vb, red);
        __syncthreads();
    }
    grid_barrier(barrier);
    // (2) 64 partial sums of g^2, one per workgroup of the first 64 (the summation order of adam_norm_kernel)
    if (blockIdx.x < kAdamBlocks) {
        const float t = adam_norm_block(adam.g, adam.n, blockIdx.x, reinterpret_cast<float*>(red));
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
    // {lr, step} as the step found them: read by everybody BEFORE the barrier, published by one thread AFTER it
    const double lr = state[0], step_in = state[1];
    grid_barrier(barrier);
    // (3) clip + Adam + the KL rule
    const AdamScalars s = adam_scalars(adam, partial, lr, step_in);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[0] = adam_next_lr(adam, s.lr);
        state[1] = s.step;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < adam.n; i += kTailBlocks * 256) adam_update_element(adam, s, i);
    grid_barrier(barrier);
    // (4) the weight images of the next step
    const int units = in_prepare_threads(prep.planes_t != nullptr);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < units; t += kTailBlocks * 256)
        split_in_prepare_unit(t, prep.W1, prep.b1, prep.D, prep.W2, prep.img, prep.planes_t);
}

}  // namespace

extern "C" int ag_update_tail_barrier_bytes(void) { return 64; }

extern "C" int ag_update_tail(const ag_sum_job* jobs, int njobs, float* scratch, long long scratch_floats, float* param, float* grad,
                              float* exp_avg, float* exp_avg_sq, double* state, int n, float beta1, float beta2, float eps,
                              float weight_decay, float max_grad_norm, float kl_threshold, float min_lr, float max_lr,
                              const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev, void* planes_t_dev,
                              void* barrier_dev, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || n <= 0 || !barrier_dev) return AG_ERR_INVALID_ARG;
    if (!W1_dev || !b1_dev || !W2_dev || !image_dev) return AG_ERR_INVALID_ARG;
    if (!(D == 16 || D == 18 || D == 20)) return AG_ERR_UNSUPPORTED;      // ag_split_gemm_input_fwd_supported
    if ((((uintptr_t)image_dev | (uintptr_t)planes_t_dev) & 15) || ((uintptr_t)barrier_dev & 3)) return AG_ERR_INVALID_ARG;
    SumJobs k;
    int b1 = 0, b2 = 0;
    const int rc = build_sum_jobs(jobs, njobs, scratch, scratch_floats, k, b1, b2);
    if (rc != AG_OK) return rc;
    const AdamArgs a{param, grad, exp_avg, exp_avg_sq, state, n, beta1, beta2, eps, weight_decay, max_grad_norm,
                     kl_threshold, min_lr, max_lr};
    // state_dev layout (ag_adam_state_bytes): double[2] {lr, step} | double[2] (unused here) | 64 float partials
    float* partial = reinterpret_cast<float*>(state + 4);
    const PrepareArgs p{W1_dev, b1_dev, D, W2_dev, (uint4*)image_dev, (uint4*)planes_t_dev};
    hipLaunchKernelGGL(update_tail_kernel, dim3(kTailBlocks), dim3(256), 0, (hipStream_t)stream, k, b2, a, partial, state, p,
                       (unsigned*)barrier_dev);
    return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;
}
