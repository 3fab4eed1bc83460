/*
 * oracle/ref_build/hip_adapter.h -- force-included (-include) ahead of every hipify-perl'ed reference
 * translation unit when oracle/ref_build/build_ref.sh compiles THE REFERENCE ITSELF for gfx950.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/stp_oracle.h).  This header contains no reference code and no
 * rasterizer logic.  It only supplies the few spellings CUDA's headers accept and ROCm 7.2's reject,
 * so that the reference's kernels compile unmodified in meaning:
 *   1. cooperative-groups tile.shfl() of float2/float3/float4 (CUDA shuffles any trivially copyable
 *      type up to 32 bytes; HIP's tile.shfl() forwards to __shfl(T, lane, width), which exists for
 *      scalars only) -> component-wise overloads;
 *   2. __ballot_sync / __shfl_sync called with CUDA's 32-bit warp mask (HIP statically asserts a
 *      64-bit mask).  The reference hard-wires a 32-lane warp (auxiliary.h:26), so the 32-lane meaning
 *      is kept: ballot of the caller's own 32-lane half of the wave64, shuffle of width 32;
 *   3. __trap (auxiliary.h:231).
 * hipcub is included first because the reference's `#define NUM_CHANNELS 3` (config.h:15) collides
 * with a template parameter name inside hipcub's device_histogram.hpp.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <hip/hip_cooperative_groups.h>

__device__ inline float2 __shfl(float2 v, int src, int width = warpSize) {
    return make_float2(__shfl(v.x, src, width), __shfl(v.y, src, width));
}
__device__ inline float3 __shfl(float3 v, int src, int width = warpSize) {
    return make_float3(__shfl(v.x, src, width), __shfl(v.y, src, width), __shfl(v.z, src, width));
}
__device__ inline float4 __shfl(float4 v, int src, int width = warpSize) {
    return make_float4(__shfl(v.x, src, width), __shfl(v.y, src, width), __shfl(v.z, src, width),
                       __shfl(v.w, src, width));
}

__device__ inline unsigned int stp_ref_ballot32(int pred) {
    const unsigned long long b = __ballot(pred);
    return (unsigned int)(b >> (__lane_id() & 32u));
}
template <class T> __device__ inline T stp_ref_shfl32(T v, int src) { return __shfl(v, src, 32); }

#define __ballot_sync(mask, pred) stp_ref_ballot32(pred)
#define __shfl_sync(mask, v, src) stp_ref_shfl32(v, src)
#define __trap() __builtin_trap()
