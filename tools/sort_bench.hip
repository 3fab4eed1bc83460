// sort_bench.hip -- the device-wide tile-bit sort of the binning stage in isolation: rocPRIM radix_sort_pairs on (u64 key, u32 id)
// pairs, bits [32, 32 + tile bits), with the library's gfx950 default configuration and with other onesweep configurations.
// Keys are laid out as duplicate_kernel leaves them (Gaussian-major, each Gaussian a small rectangle of tiles).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_bench.hip -o tools/sort_bench.bin && tools/sort_bench.bin [R] [gx] [gy]
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class Config>
static float run(const char* name, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t R, unsigned b0, unsigned b1,
                 std::vector<uint64_t>* ref)
{
    size_t bytes = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, R, b0, b1)));
    void* tmp = nullptr;
    CK(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, R, b0, b1)));
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, R, b0, b1)));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(R);
    CK(hipMemcpy(h.data(), kout, R * 8, hipMemcpyDeviceToHost));
    bool same = true;
    if (ref->empty()) *ref = h; else same = memcmp(ref->data(), h.data(), R * 8) == 0;
    printf("%-34s %8.2f us per sort   temp %zu B   %s\n", name, 1000.0f * ms / reps, bytes, same ? "identical" : "DIFFERENT");
    CK(hipFree(tmp));
    return ms / reps;
}

template <unsigned BS, unsigned IPT, unsigned BITS, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::match>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

int main(int argc, char** argv)
{
    const size_t R = argc > 1 ? (size_t)atol(argv[1]) : 2620299;
    const int gx = argc > 2 ? atoi(argv[2]) : 120, gy = argc > 3 ? atoi(argv[3]) : 68;
    unsigned bit = 0;
    while ((1u << bit) <= (unsigned)(gx * gy)) bit++; // (bit above the MSB, like higher_msb)
    std::vector<uint64_t> hk(R);
    std::vector<uint32_t> hv(R);
    std::mt19937 rng(1);
    size_t n = 0;
    uint32_t g = 0;
    while (n < R) { // one Gaussian: a w x h rectangle of tiles, row-major, 2.8 tiles on average
        const int w = 1 + (int)(rng() % 2) + (int)(rng() % 8 == 0), h = 1 + (int)(rng() % 2) + (int)(rng() % 8 == 0);
        const int x0 = (int)(rng() % (unsigned)gx), y0 = (int)(rng() % (unsigned)gy);
        const uint32_t depth = 0x3f000000u + (rng() & 0xffffffu);
        for (int y = y0; y < y0 + h && y < gy; y++)
            for (int x = x0; x < x0 + w && x < gx; x++)
                if (n < R) { hk[n] = ((uint64_t)(y * gx + x) << 32) | depth; hv[n] = g; n++; }
        g++;
    }
    uint64_t *kin, *kout; uint32_t *vin, *vout;
    CK(hipMalloc(&kin, R * 8)); CK(hipMalloc(&kout, R * 8)); CK(hipMalloc(&vin, R * 4)); CK(hipMalloc(&vout, R * 4));
    CK(hipMemcpy(kin, hk.data(), R * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(vin, hv.data(), R * 4, hipMemcpyHostToDevice));
    printf("R = %zu pairs, %d x %d tiles, sort on bits [32, %u)\n", R, gx, gy, 32 + bit);
    std::vector<uint64_t> ref;
    using rocprim::block_radix_rank_algorithm;
    run<rocprim::default_config>("library default (512 x 16, 8 bits)", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<256, 8, 8>>("256 x 8, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<256, 12, 8>>("256 x 12, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<256, 16, 8>>("256 x 16, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 4, 8>>("512 x 4, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 8, 8>>("512 x 8, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 12, 8>>("512 x 12, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<1024, 4, 8>>("1024 x 4, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<1024, 8, 8>>("1024 x 8, 8 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 16, 7>>("512 x 16, 7 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 8, 7>>("512 x 8, 7 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<256, 8, 7>>("256 x 8, 7 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<256, 12, 7>>("256 x 12, 7 bits", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    run<OS<512, 8, 6>>("512 x 8, 6 bits (3 passes)", kin, kout, vin, vout, R, 32, 32 + bit, &ref);
    return 0;
}
