// stp_hbm_probe.hip -- streaming read / write / copy kernels that measure what THIS box's HBM delivers to a plain float4 stream.
// Measurement infrastructure (bench.py: `hbm_measured`): the streaming stages of the path (preprocess, colour, entry gather, per-Gaussian
// backward) are priced against these ceilings, next to the 8 TB/s spec peak the contract's roofline fractions use.  MI355X_MICROARCH.md
// quotes 6.29 TB/s for a copy; a reduction written with a library's generic kernel (torch.sum: 3.9 TB/s) is not a read ceiling.
#include "stp_internal.h"

namespace stp {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HBM_UNROLL = 8; // 16-byte accesses a thread keeps in flight

template <bool NT> __global__ void __launch_bounds__(256) hbm_read_kernel(const f32x4* __restrict__ p, size_t n16, float* __restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    for (; i + (HBM_UNROLL - 1) * stride < n16; i += HBM_UNROLL * stride) {
        f32x4 v[HBM_UNROLL];
#pragma unroll
        for (int u = 0; u < HBM_UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < HBM_UNROLL; u++) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n16; i += stride) { const f32x4 v = p[i]; acc += (v.x + v.y) + (v.z + v.w); }
    if (acc == 1.2345678e-33f) sink[0] = acc; // (never: keeps the loads alive without a store stream)
}

template <bool NT> __global__ void __launch_bounds__(256) hbm_write_kernel(f32x4* __restrict__ p, size_t n16, float value)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x4 z = {value, value, value, value};
    for (; i + (HBM_UNROLL - 1) * stride < n16; i += HBM_UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < HBM_UNROLL; u++) { if (NT) __builtin_nontemporal_store(z, p + i + u * stride); else p[i + u * stride] = z; }
    }
    for (; i < n16; i += stride) p[i] = z;
}

template <bool NT> __global__ void __launch_bounds__(256) hbm_copy_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (HBM_UNROLL - 1) * stride < n16; i += HBM_UNROLL * stride) {
        f32x4 v[HBM_UNROLL];
#pragma unroll
        for (int u = 0; u < HBM_UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < HBM_UNROLL; u++) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

} // namespace
} // namespace stp

extern "C" int stp_hbm_probe(int kind, void* dst, const void* src, size_t bytes, int blocks, void* stream)
{
    using namespace stp;
    if (bytes % 16 != 0 || blocks <= 0 || kind < 0 || kind > 6 || ((kind & 3) != 0 && !dst) || ((kind & 3) != 1 && !src)) return STP_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const size_t n16 = bytes / 16;
    const bool nt = (kind & 4) != 0; // + 4: non-temporal loads / stores
    switch (kind & 3) {
    case 0:
        if (nt) hipLaunchKernelGGL(hbm_read_kernel<true>, dim3(blocks), dim3(256), 0, st, (const f32x4*)src, n16, (float*)(dst ? dst : const_cast<void*>(src)));
        else hipLaunchKernelGGL(hbm_read_kernel<false>, dim3(blocks), dim3(256), 0, st, (const f32x4*)src, n16, (float*)(dst ? dst : const_cast<void*>(src)));
        break;
    case 1:
        if (nt) hipLaunchKernelGGL(hbm_write_kernel<true>, dim3(blocks), dim3(256), 0, st, (f32x4*)dst, n16, 2.0f);
        else hipLaunchKernelGGL(hbm_write_kernel<false>, dim3(blocks), dim3(256), 0, st, (f32x4*)dst, n16, 2.0f);
        break;
    case 2:
        if (nt) hipLaunchKernelGGL(hbm_copy_kernel<true>, dim3(blocks), dim3(256), 0, st, (f32x4*)dst, (const f32x4*)src, n16);
        else hipLaunchKernelGGL(hbm_copy_kernel<false>, dim3(blocks), dim3(256), 0, st, (f32x4*)dst, (const f32x4*)src, n16);
        break;
    default: return STP_ERR_INVALID_ARGUMENT;
    }
    return hipGetLastError() == hipSuccess ? 0 : STP_ERR_HIP;
}
