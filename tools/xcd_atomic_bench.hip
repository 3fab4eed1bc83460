// xcd_atomic_bench.hip -- what does a per-tile counter cost when it is (a) one device-wide array updated with agent-scope atomics
// (eight XCDs, eight L2s: the atomic has to be resolved behind them) or (b) one array PER XCD, updated with workgroup-scope
// atomics by the workgroups that run on that XCD (an atomic always executes in the L2, and only this XCD's L2 ever holds the
// array)?  N events on T counters, with and without using the returned value (the cursor form), plus the (key, id) scatter the
// cursor form feeds.  Question behind it: can binning by tile counters (STP_SORT=counters) beat the device-wide radix sort?
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_atomic_bench.hip -o tools/xcd_atomic_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF); } // HW_REG_XCC_ID, bits [3:0]

// MODE 0: agent scope, no return; 1: agent scope, returned value used as a slot for a 12-byte scatter
// MODE 2: per-XCD array, workgroup scope, no return; 3: per-XCD array, workgroup scope, slot + scatter
// MODE 4: no atomics at all, scatter to hashed slots (what the writes alone cost)
template <int MODE>
__global__ void __launch_bounds__(256) bench(uint32_t* counters, int T, int per_thread, uint64_t* keys, uint32_t* ids, uint32_t cap, uint32_t* xcd_seen)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const int xcd = xcc_id();
    if (threadIdx.x == 0) atomicOr(&xcd_seen[blockIdx.x & 1023], 1u << xcd);
    uint32_t* const mine = (MODE == 2 || MODE == 3) ? counters + (size_t)xcd * T : counters;
    for (int e = 0; e < per_thread; e++) {
        const uint32_t h = hash32(gid * 31u + e);
        const int t = (int)(h % (uint32_t)T);
        if (MODE == 0) __hip_atomic_fetch_add(&mine[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) __hip_atomic_fetch_add(&mine[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else {
            uint32_t slot;
            if (MODE == 1) slot = __hip_atomic_fetch_add(&mine[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 3) slot = __hip_atomic_fetch_add(&mine[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else slot = h >> 8;
            // tile t's segment starts at t * (cap / T); XCD x's part of it at x / 8 of the segment (MODE 3)
            const uint32_t seg = cap / (uint32_t)T;
            uint32_t pos = (uint32_t)t * seg + (MODE == 3 ? (uint32_t)xcd * (seg / 8) + slot % (seg / 8) : slot % seg);
            keys[pos] = ((uint64_t)t << 32) | h;
            ids[pos] = gid;
        }
    }
}

template <int MODE> static void run(const char* name, int T, size_t N)
{
    const int per_thread = 3, threads = (int)((N + per_thread - 1) / per_thread), blocks = (threads + 255) / 256;
    const uint32_t cap = (uint32_t)(((N * 5 / 4) / T / 8 + 1) * 8 * T);
    uint32_t *counters, *ids, *seen; uint64_t* keys;
    CK(hipMalloc(&counters, (size_t)8 * T * 4)); CK(hipMalloc(&keys, (size_t)cap * 8)); CK(hipMalloc(&ids, (size_t)cap * 4)); CK(hipMalloc(&seen, 1024 * 4));
    CK(hipMemset(seen, 0, 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipMemsetAsync(counters, 0, (size_t)8 * T * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, counters, T, per_thread, keys, ids, cap, seen);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    std::vector<uint32_t> h((size_t)8 * T), hs(1024);
    CK(hipMemcpy(h.data(), counters, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), seen, 4096, hipMemcpyDeviceToHost));
    unsigned long long total = 0; for (uint32_t v : h) total += v;
    int rr = 0; for (int b = 0; b < 1024 && b < blocks; b++) rr += hs[b] == (1u << (b & 7)); // workgroup b ran on XCD b mod 8?
    printf("%-58s mean %7.1f us  min %7.1f us   counted %llu of %zu   round-robin blocks %d / %d\n", name, 1000 * sum / reps, 1000 * best, total,
           (size_t)blocks * 256 * per_thread, rr, blocks < 1024 ? blocks : 1024);
    CK(hipFree(counters)); CK(hipFree(keys)); CK(hipFree(ids)); CK(hipFree(seen));
}

int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8160;
    const size_t N = argc > 2 ? (size_t)atol(argv[2]) : 2620299;
    printf("%zu events on %d counters\n", N, T);
    run<0>("agent scope, one array, fire and forget", T, N);
    run<2>("workgroup scope, array per XCD, fire and forget", T, N);
    run<1>("agent scope, one array, cursor + 12-byte scatter", T, N);
    run<3>("workgroup scope, array per XCD, cursor + 12-byte scatter", T, N);
    run<4>("no atomics, 12-byte scatter to hashed slots", T, N);
    return 0;
}
