// microbenchmark: throughput of LDS atomics by type / address pattern on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE, int GROUP>  // GROUP = lanes sharing one address
__global__ void __launch_bounds__(256) k(int iters, float* out)
{
    __shared__ float sf[4][9 * 128];
    __shared__ unsigned int su[4][9 * 128];
    __shared__ unsigned long long sl[4][9 * 128];
    __shared__ double sd[4][9 * 128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 9 * 128; i += 64) { sf[w][i] = 0; su[w][i] = 0; sl[w][i] = 0; sd[w][i] = 0; }
    __syncthreads();
    unsigned int rng = blockIdx.x * 977 + w * 131 + 7;
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        rng = rng * 1664525u + 1013904223u;
        const int slot = ((rng >> 8) + (lane / GROUP) * 5) & 127;   // wave-uniform random base, GROUP lanes share a slot
        const float v = (float)(lane + it) * 1e-3f;
#pragma unroll
        for (int kk = 0; kk < 9; kk++) {
            if (MODE == 0) atomicAdd(&sf[w][kk * 128 + slot], v);
            if (MODE == 1) atomicAdd(&su[w][kk * 128 + slot], (unsigned int)(v * 1024.0f));
            if (MODE == 2) atomicAdd(&sl[w][kk * 128 + slot], (unsigned long long)(long long)(v * 1048576.0f));
            if (MODE == 3) { float t = sf[w][kk * 128 + slot]; sf[w][kk * 128 + slot] = t + v; }
            if (MODE == 4) acc += __shfl_xor(v, 1) + v;
            if (MODE == 5) atomicAdd(&sd[w][kk * 128 + slot], (double)v);
            if (MODE == 6) { const double tq = fma((double)v, 1048576.0, 6755399441055744.0); atomicAdd(&sl[w][kk * 128 + slot], (unsigned long long)(__double_as_longlong(tq) - 0x4338000000000000ll)); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sf[0][3] + (float)su[0][3] + (float)sl[0][3] + (float)sd[0][3] + acc;
}
template <int MODE, int GROUP> void run(const char* name)
{
    float* d; hipMalloc(&d, 4096 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 1024;
    hipLaunchKernelGGL((k<MODE, GROUP>), dim3(blocks), dim3(256), 0, 0, 10, d);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, GROUP>), dim3(blocks), dim3(256), 0, 0, iters, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waveops = (double)blocks * 4 * iters * 9;
    printf("%-28s group=%2d : %8.3f ms  %7.2f G wave-ops/s  (%.1f cycles per wave-op per CU @2.3GHz, 256 CUs)\n", name, GROUP, ms,
           waveops / ms / 1e6, ms * 1e-3 * 2.3e9 * 256 / waveops);
    hipFree(d);
}
int main()
{
    run<0, 1>("ds_add_f32"); run<0, 4>("ds_add_f32"); run<0, 16>("ds_add_f32");
    run<1, 1>("ds_add_u32"); run<1, 4>("ds_add_u32"); run<1, 16>("ds_add_u32");
    run<2, 1>("ds_add_u64"); run<2, 4>("ds_add_u64"); run<2, 16>("ds_add_u64");
    run<3, 1>("plain read+write"); run<3, 4>("plain read+write");
    run<4, 1>("valu baseline");
    run<5, 1>("ds_add_f64"); run<5, 4>("ds_add_f64"); run<5, 16>("ds_add_f64");
    run<6, 1>("fixed-point u64 (fma trick)"); run<6, 4>("fixed-point u64 (fma trick)");
    return 0;
}
