// fill_bench.hip -- zeroing the backward's 64 MB of gradient records: hipMemsetAsync against a plain kernel of 16-byte stores
// (torch's fill kernel takes 23.6 us for it in the frame: profiles/r03_full/kernel_stats.csv).
//   hipcc --offload-arch=gfx950 -O3 tools/fill_bench.hip -o tools/fill_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int UNROLL> __global__ void __launch_bounds__(256) zero_kernel(float4* p, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 z = make_float4(0, 0, 0, 0);
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) p[i + u * stride] = z;
    }
    for (; i < n16; i += stride) p[i] = z;
}
int main(int argc, char** argv)
{
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : (size_t)64 << 20;
    void* p; CK(hipMalloc(&p, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto f) {
        for (int i = 0; i < 5; i++) f();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 30; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = ms < best ? ms : best; }
        printf("%-40s mean %6.1f us  min %6.1f us  (%.2f TB/s at the minimum)\n", name, 1000 * sum / 30, 1000 * best, bytes / (best * 1e-3) / 1e12);
    };
    time("hipMemsetAsync", [&] { CK(hipMemsetAsync(p, 0, bytes, 0)); });
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        char name[64]; snprintf(name, 64, "zero_kernel<4>, %d blocks", blocks);
        time(name, [&] { hipLaunchKernelGGL(zero_kernel<4>, dim3(blocks), dim3(256), 0, 0, (float4*)p, bytes / 16); });
    }
    time("zero_kernel<1>, one store per thread", [&] { hipLaunchKernelGGL(zero_kernel<1>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, (float4*)p, bytes / 16); });
    return 0;
}
