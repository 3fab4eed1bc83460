// tools/mfma_probe.hip -- what v_mfma_f32_4x4x1_16b_f32 computes, lane by lane (run on the GPU box: tools/mfma_probe.bin).
// The head level's candidate evaluation uses it as "entry held by lane r of my quad  x  my pixel's ray component", 16 quads
// at once: D[r] of lane (quad b, pixel j) must be A(lane 4b + r) * B(lane 4b + j) + C[r], with ONE rounding (an fmaf).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, const float* c, float* d)
{
    const int l = threadIdx.x;
    float4v acc = {c[4 * l + 0], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[4 * l + r] = acc[r];
}
int main()
{
    float ha[64], hb[64], hc[256], hd[256];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    int bad_layout = 0, bad_round = 0, denorm_flushed = 0;
    for (int rep = 0; rep < 200; rep++) {
        for (int i = 0; i < 64; i++) { ha[i] = rnd(); hb[i] = rnd(); }
        for (int i = 0; i < 256; i++) hc[i] = rep % 3 == 0 ? 0.0f : rnd() * 1e-3f;
        if (rep == 7) { ha[5] = 1e-30f; hb[6] = 1e-10f; hc[4 * 6 + 1] = 0.0f; } // lane 6 (quad 1, pixel 2), r = 1 (lane 5): a denormal product
        float *da, *db, *dc, *dd;
        hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const float A = ha[4 * (l / 4) + r], B = hb[l];
                const float want = fmaf(A, B, hc[4 * l + r]);
                const float got = hd[4 * l + r];
                if (memcmp(&want, &got, 4) != 0) {
                    if (rep == 7 && l == 6 && r == 1) { denorm_flushed = got == 0.0f; continue; }
                    const float two = A * B + hc[4 * l + r]; // (host compilers may fuse this too; informational)
                    if (fabsf(want - got) > 1e-3f * fmaxf(1.0f, fabsf(want))) bad_layout++; else bad_round++;
                    if (bad_layout + bad_round < 6) printf("rep %d lane %d r %d: want %.9g got %.9g (unfused %.9g)\n", rep, l, r, want, got, two);
                }
            }
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    }
    printf("v_mfma_f32_4x4x1_16b_f32: layout mismatches %d, rounding mismatches vs fmaf %d, denormal product flushed: %s\n", bad_layout, bad_round,
           denorm_flushed ? "yes" : "no");
    return bad_layout || bad_round;
}
