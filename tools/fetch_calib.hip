// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of tile_sort_gather_kernel.
// MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read (128-byte requests tallied at 64 bytes) and
// says "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The entry gather reads one
// random 64-byte line (four 16-byte loads of ONE thread) and one random 12-byte colour per list entry; bench.py doubles its FETCH_SIZE like
// everybody else's.  Every kernel here moves a KNOWN number of distinct bytes out of a 2 GiB buffer (beyond the 256 MiB Infinity Cache):
//   stream16      every lane 16 consecutive bytes, unit stride                      (the guide's calibration point)
//   line64_thread every lane the four float4 of its own random 64-byte line          (write_entry's gpack read)
//   line64_quad   four lanes share one random 64-byte line, 16 bytes each
//   line128_thread every lane the eight float4 of its own random 128-byte line
//   rgb12         every lane three floats at a random 12-byte-aligned position      (write_entry's colour read)
// Run under `rocprofv3 --pmc FETCH_SIZE` (tools/fetch_calib.sh); the program prints the bytes each kernel asked for.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

namespace stp { // (profiles/pmc_summary.py keeps kernels of this namespace)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void stream16(const float4* __restrict__ src, size_t n, float* __restrict__ sink)
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void line64_thread(const float4* __restrict__ src, uint32_t lines, uint32_t n, float* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* p = src + 4 * (size_t)(hash32(i) % lines);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float acc = a.x + b.y + c.z + d.w;
    if (acc == 12345.678f) *sink = acc;
}
__global__ void line64_quad(const float4* __restrict__ src, uint32_t lines, uint32_t n, float* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * n) return;
    const float4 a = src[4 * (size_t)(hash32(i >> 2) % lines) + (i & 3)];
    if (a.x + a.w == 12345.678f) *sink = a.x;
}
__global__ void line128_thread(const float4* __restrict__ src, uint32_t lines, uint32_t n, float* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* p = src + 8 * (size_t)(hash32(i) % (lines / 2));
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) { const float4 v = p[k]; acc += v.x + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void rgb12(const float* __restrict__ src, uint32_t triples, uint32_t n, float* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = src + 3 * (size_t)(hash32(i) % triples);
    const float acc = p[0] + p[1] + p[2];
    if (acc == 12345.678f) *sink = acc;
}

} // namespace stp
using namespace stp;

int main()
{
    const size_t bytes = 2ull << 30;
    float4* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    const uint32_t lines = (uint32_t)(bytes / 64), n = 8u << 20; // 8 Mi random lines out of 32 Mi: ~12 % picked twice
    const size_t nstream = (1ull << 30) / 16;
    for (int rep = 0; rep < 3; rep++) {
        stream16<<<4096, 256>>>(buf, nstream, sink);
        line64_thread<<<(n + 255) / 256, 256>>>(buf, lines, n, sink);
        line64_quad<<<(4 * n + 255) / 256, 256>>>(buf, lines, n, sink);
        line128_thread<<<(n + 255) / 256, 256>>>(buf, lines, n, sink);
        rgb12<<<(n + 255) / 256, 256>>>((const float*)buf, (uint32_t)(bytes / 12), n, sink);
    }
    CK(hipDeviceSynchronize());
    printf("asked bytes per launch: stream16 %zu  line64_thread %zu  line64_quad %zu  line128_thread %zu  rgb12 %zu (x64-byte lines touched: %zu .. %zu)\n",
           (size_t)1 << 30, (size_t)n * 64, (size_t)n * 64, (size_t)n * 128, (size_t)n * 12, (size_t)n * 64, (size_t)n * 128);
    return 0;
}
