// stp_debug_viz.hip -- second half of the debug depth visualisation (`render_depth=True`).
//
// Replaces applyDebugVisualization for DebugVisualization::Depth (reference rasterizer_impl.cu:54-109: cub min / max over
// channel 0 of the frame) and FORWARD::render_debug / render_debug_CUDA<DEPTH = true> + colormapTurbo (forward.cu:674-729,
// stopthepop_common.cuh:641-657).  The render kernels have left sum(depth * alpha * T) in channel 0 and the final
// transmittance T in channel 1 of out_color; here the frame's minimum and maximum of channel 0 are reduced on the
// device (no host round trip: the reference copies them back only to feed the viewer's statistics callback) and every
// pixel becomes  turbo( clamp(value + T * max, min, max) / (max - min) )  -- the reference's expression, kept as is.
#include "stp_internal.h"
#include "../../include/stp_turbo_colormap.h"

namespace stp {

namespace {

__device__ __constant__ float c_turbo[STP_TURBO_ENTRIES * 3] = STP_TURBO_TABLE_INITIALIZER;

// order-preserving float <-> uint mapping so that the extrema can be taken with integer atomics
__device__ __forceinline__ uint32_t to_ordered(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered(uint32_t o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

__global__ void __launch_bounds__(256) minmax_kernel(const float* __restrict__ v, int N, uint32_t* __restrict__ mm) // mm[0] = min, mm[1] = max (ordered)
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const uint32_t o = to_ordered(v[i]);
        lo = min(lo, o);
        hi = max(hi, o);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

__global__ void __launch_bounds__(256) depth_colormap_kernel(float* __restrict__ out_color, int N, const uint32_t* __restrict__ mm)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    const float mn = from_ordered(mm[0]), mx = from_ordered(mm[1]);
    const float T = out_color[N + idx];
    // glm::clamp = min(max(x, lo), hi) with glm's `a < b ? b : a` selections: a NaN passes through (an empty frame has
    // min == max, the reference divides 0 by 0 and returns NaN pixels; fminf / fmaxf would turn them into table entry 0)
    auto clamp_glm = [](float v, float lo, float hi) { const float t = (v < lo) ? lo : v; return (hi < t) ? hi : t; };
    const float x = clamp_glm(out_color[idx] + T * mx, mn, mx) / (mx - mn);
    // colormapTurbo: linear interpolation in the 256-entry table, every channel clamped to [0, 1]
    const float interp = clamp_glm(x * 255.0f, 0.0f, 255.0f);
    const int lo = x > 0.0f ? (int)interp : 0;
    const int hi = lo >= 255 ? 255 : lo + 1;
    const float diff = interp - (float)lo;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float a = c_turbo[3 * lo + ch], b = c_turbo[3 * hi + ch];
        out_color[ch * N + idx] = clamp_glm(a + (b - a) * diff, 0.0f, 1.0f);
    }
}

} // namespace

// minmax: two words of the image buffer (ImageState::dbg_minmax)
hipError_t launch_depth_colormap(float* out_color, int N, uint32_t* minmax, hipStream_t st)
{
    if (N <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(minmax, 0xFF, sizeof(uint32_t), st); // running minimum starts at the largest key
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(minmax + 1, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(minmax_kernel, dim3(1024), dim3(256), 0, st, out_color, N, minmax);
    hipLaunchKernelGGL(depth_colormap_kernel, dim3((N + 255) / 256), dim3(256), 0, st, out_color, N, minmax);
    return hipGetLastError();
}

} // namespace stp
