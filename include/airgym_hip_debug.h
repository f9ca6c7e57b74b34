/*
 * airgym_hip_debug.h - benchmark and diagnostic entry points of libairgym_hip.so.
 *
 * NOT part of the drop-in interface (include/airgym_hip.h) and NOT in the shipped libairgym_hip.so: these symbols exist only
 * in libairgym_hip_exp.so (`python airgym_amd/csrc/build.py --experiments`, loaded by tools/ with AIRGYM_EXPERIMENTS=1).
 * Nothing here replaces reference behaviour.  They let tools/ price the launch floor of the env-step kernel, see where the
 * hardware places its waves, pin scheduling variants of the split GEMM and time the Planning render kernel's phases.
 */
#ifndef AIRGYM_HIP_DEBUG_H
#define AIRGYM_HIP_DEBUG_H

#include "airgym_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A kernel with the Hovering/CTBR step's loads and stores and no arithmetic (launch + memory-latency floor). */
int ag_debug_touch(ag_handle h, const float* actions_dev, void* stream);
/* mode 1 = non-temporal stores, 2 = non-temporal loads + stores, 3 = empty kernel (pure dependent-launch boundary) */
int ag_debug_touch_variant(ag_handle h, const float* actions_dev, int mode, void* stream);

/* Where the hardware places the step kernel's waves: launches its geometry (ceil(n/64) workgroups x 2 waves) and writes, per
 * wave, {HW_REG_HW_ID, HW_REG_XCC_ID} into out_dev [ceil(n/64) * 2, 2] u32 (tools/wave_placement.py decodes SIMD / CU / XCC). */
int ag_debug_wave_placement(ag_handle h, unsigned int* out_dev, void* stream);

/* Row-tile size of ag_split_gemm / ag_split_gemm_elu_heads / ag_split_gemm_input_wgrad: 2 = 128 rows per workgroup (4 waves, two
 * workgroups per CU), 4 = 256 rows (8 waves, one per CU), -1 = the shipped default.  Also read once from AIRGYM_SPLIT_WM.  Changes
 * what ag_split_gemm_input_wgrad_rows() returns: set it before a caller sizes its partial buffers. */
int ag_debug_split_gemm_variant(int variant);

/* ag_split_wgrad: 1 (default) = the issue order of a chunk is prescribed (staging work spread between the MFMAs), 0 = left to
 * the compiler. */
int ag_debug_split_wgrad_ordered(int on);

/* ag_mlp_chain_forward with parts of the kernel left out (timing ablations; results are then meaningless): bit0 no weight
 * LDS-DMA behind the first block, bit1 no MFMA, bit2 no workgroup barriers. */
int ag_debug_chain_skip(int mask);

/* Next Planning step renders with parts of the render kernel skipped: bit0 ray-cast, bit1 noise passes, bit2 5x5 pass. */
int ag_debug_planning_render_parts(ag_handle h, int skip_mask);

#ifdef __cplusplus
}
#endif
#endif /* AIRGYM_HIP_DEBUG_H */
