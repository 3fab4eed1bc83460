// numerics_probe.hip -- can cheaper instruction sequences replace IEEE 1/x and expf in the render kernels?
//  (1) 1/x: exhaustive comparison over every positive normal float of  v_rcp_f32 + n Newton steps (fma)  against the
//      correctly rounded quotient.  The ordering-critical depth uses 1/x, so only a bit-exact sequence is acceptable.
//  (2) exp(x), x in [-16, 0]: ulp error histogram of a 6-instruction v_exp_f32 sequence against double-precision exp.
// Build: hipcc --offload-arch=gfx950 -O3 -o numerics_probe.bin numerics_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__global__ void rcp_probe(unsigned long long* bad)
{
    const uint32_t first = 0x00800000u, last = 0x7f7fffffu;
    for (uint64_t b = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float ref = __fdiv_rn(1.0f, x);
        float r = __builtin_amdgcn_rcpf(x);
        if (r != ref) atomicAdd(&bad[0], 1ull);
        float e = fmaf(-x, r, 1.0f);
        r = fmaf(e, r, r);
        if (r != ref) { atomicAdd(&bad[1], 1ull); atomicAdd(&bad[64 + (b >> 23)], 1ull); }
        e = fmaf(-x, r, 1.0f);
        r = fmaf(e, r, r);
        if (r != ref) { atomicAdd(&bad[2], 1ull); if (x > 1e-5f && x < 1e30f) atomicAdd(&bad[3], 1ull); }
    }
}

__device__ __forceinline__ float fast_exp(float x)
{
    // exp(x) = 2^(x*log2e): y = rn(x*log2e), r = exact residual of the product + low part of log2e, first-order fix-up
    const float y = x * 1.44269502162933349609375f;
    float r = fmaf(x, 1.44269502162933349609375f, -y);
    r = fmaf(x, 1.925963033500011e-8f, r);
    const float g = __builtin_amdgcn_exp2f(y);
    return fmaf(g, r * 0.693147182464599609375f, g);
}

__global__ void exp_probe(unsigned long long* hist_fast, unsigned long long* hist_ocml, unsigned long long* hist_plain, unsigned long long* differ)
{
    // every float in [-16, -2^-20]
    const uint32_t first = __float_as_uint(-9.5367431640625e-07f), last = __float_as_uint(-16.0f);
    for (uint64_t b = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const double ref = exp((double)x);
        const float rf = (float)ref;
        const float ulp = __uint_as_float(__float_as_uint(rf) + 1) - rf;
        auto bucket = [&](float v) { const double e = fabs(((double)v - ref) / (double)ulp); return e < 0.5 ? 0 : e < 1.0 ? 1 : e < 2.0 ? 2 : e < 4.0 ? 3 : e < 8.0 ? 4 : 5; };
        const float f = fast_exp(x), o = expf(x), p = __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f);
        atomicAdd(&hist_fast[bucket(f)], 1ull);
        atomicAdd(&hist_ocml[bucket(o)], 1ull);
        atomicAdd(&hist_plain[bucket(p)], 1ull);
        if (f != o) atomicAdd(&differ[0], 1ull);
        if (o != rf) atomicAdd(&differ[1], 1ull);
        if (f != rf) atomicAdd(&differ[2], 1ull);
    }
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 320 * 8); hipMemset(d, 0, 320 * 8);
    rcp_probe<<<4096, 256>>>(d);
    exp_probe<<<4096, 256>>>(d + 8, d + 16, d + 24, d + 32);
    unsigned long long h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("1/x over all positive normal floats (2130706432 values): mismatches vs correctly rounded\n");
    printf("  v_rcp_f32 alone          %llu\n  + 1 Newton step          %llu\n  + 2 Newton steps         %llu   (of which in [1e-5, 1e30]: %llu)\n", h[0], h[1], h[2], h[3]);
    printf("  1 Newton step, mismatches by biased exponent of x:");
    for (int e = 0; e < 256; e++) if (h[64 + e]) printf(" %d:%llu", e, h[64 + e]);
    printf("\n");
    const char* names[3] = {"fast (6 instr)", "ocml expf", "v_exp(x*log2e)"};
    printf("exp(x) for every float in [-16, -9.5e-7]: error vs double exp, in ulps  [<0.5, <1, <2, <4, <8, >=8]\n");
    for (int k = 0; k < 3; k++) { printf("  %-16s", names[k]); for (int i = 0; i < 6; i++) printf(" %llu", h[8 + 8 * k + i]); printf("\n"); }
    printf("  fast != ocml: %llu   ocml != rn(exp): %llu   fast != rn(exp): %llu\n", h[32], h[33], h[34]);
    return 0;
}
