// global_atomic_bench.hip -- how should the backward hand its per-Gaussian sums (9 floats) to HBM?
// Each "event" adds nine floats to the gradient of one (random) Gaussian.  Variants:
//   A: one lane, nine atomic instructions into four separate arrays (SoA: colour P x 3, mean2D P x 3, conic P x 4, opacity P)
//   B: nine lanes, one atomic instruction into one 64-byte record per Gaussian (AoS, P x 16 floats)
//   C: as B, four events per instruction (16-lane groups)
//   D: one lane, nine atomic instructions into the 64-byte record (AoS, still nine instructions)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o global_atomic_bench.bin global_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V> __global__ void __launch_bounds__(256) bench(float* c, float* m, float* co, float* o, float* rec, int P, int events_per_wave)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    for (int e = 0; e < events_per_wave; e += (V == 2 ? 4 : 1)) {
        if (V == 0) {
            const int id = hash32(wave * 9973u + e) % P;
            if (lane == 0) {
                atomicAdd(&c[3 * id + 0], 1.0f); atomicAdd(&c[3 * id + 1], 1.0f); atomicAdd(&c[3 * id + 2], 1.0f);
                atomicAdd(&m[3 * id + 0], 1.0f); atomicAdd(&m[3 * id + 1], 1.0f);
                atomicAdd(&co[4 * id + 0], 1.0f); atomicAdd(&co[4 * id + 1], 1.0f); atomicAdd(&co[4 * id + 3], 1.0f);
                atomicAdd(&o[id], 1.0f);
            }
        } else if (V == 1) {
            const int id = hash32(wave * 9973u + e) % P;
            if (lane < 9) atomicAdd(&rec[16 * (size_t)id + lane], 1.0f);
        } else if (V == 2) {
            const int id = hash32(wave * 9973u + e + (lane >> 4)) % P;
            if ((lane & 15) < 9) atomicAdd(&rec[16 * (size_t)id + (lane & 15)], 1.0f);
        } else {
            const int id = hash32(wave * 9973u + e) % P;
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) atomicAdd(&rec[16 * (size_t)id + k], 1.0f);
            }
        }
    }
}

int main()
{
    const int P = 1000000, waves = 32640, events = 128; // ~4.2 M events, 37.6 M float adds
    float *c, *m, *co, *o, *rec;
    hipMalloc(&c, P * 12); hipMalloc(&m, P * 12); hipMalloc(&co, P * 16); hipMalloc(&o, P * 4); hipMalloc(&rec, (size_t)P * 64);
    hipMemset(c, 0, P * 12); hipMemset(m, 0, P * 12); hipMemset(co, 0, P * 16); hipMemset(o, 0, P * 4); hipMemset(rec, 0, (size_t)P * 64);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    const char* names[4] = {"A  SoA, 1 lane x 9 instr", "B  AoS, 9 lanes x 1 instr", "C  AoS, 4 events/instr", "D  AoS, 1 lane x 9 instr"};
    for (int v = 0; v < 4; v++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(t0);
            if (v == 0) bench<0><<<waves / 4, 256>>>(c, m, co, o, rec, P, events);
            if (v == 1) bench<1><<<waves / 4, 256>>>(c, m, co, o, rec, P, events);
            if (v == 2) bench<2><<<waves / 4, 256>>>(c, m, co, o, rec, P, events);
            if (v == 3) bench<3><<<waves / 4, 256>>>(c, m, co, o, rec, P, events);
            hipEventRecord(t1); hipEventSynchronize(t1);
            float ms; hipEventElapsedTime(&ms, t0, t1);
            if (rep == 2) printf("%-28s %8.3f ms  %7.2f M events/ms\n", names[v], ms, (double)waves * events / ms * 1e-6);
        }
    }
    return 0;
}
