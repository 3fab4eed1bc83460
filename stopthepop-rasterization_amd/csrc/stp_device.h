// stp_device.h -- device-side leaf math and wave64 helpers shared by the gfx950 kernels.
//
// Numerics policy (see DESIGN.md "Numerics"): every quantity that decides a discrete outcome --
// screen rectangles, radii, tile counts, the 64-bit sort keys, view rays and depth-along-ray keys --
// is evaluated with floating-point contraction disabled (or with explicitly written fmaf) in a
// fixed operation order, so tile lists and per-pixel blend orders are reproducible bit-for-bit.
// Colour/alpha arithmetic in the blend loops is left to the compiler (it may fuse).
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

namespace stp {

constexpr float ALPHA_THRESHOLD = 1.0f / 255.0f; // reference auxiliary.h:21-22
constexpr float T_THRESHOLD = 0.0001f;           // reference auxiliary.h:23
constexpr uint32_t INVALID_TILE_ID = 0xFFFFFFFFu; // reference config.h:19

__device__ __constant__ const float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ const float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ const float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                 -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                 0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                                 -0.5900435899266435f};

// ---------------------------------------------------------------- wave64 helpers
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// Intra-wave LDS hand-off: DS operations of one wave execute in issue order, so only the
// compiler must be kept from reordering or caching across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Broadcast lane I (0..3) of every aligned group of 4 lanes (DPP quad_perm, no LDS traffic).
template <int I> __device__ __forceinline__ int quad_bcast_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, I * 0x55, 0xF, 0xF, true);
}
template <int I> __device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(quad_bcast_i<I>(__float_as_int(v)));
}
__device__ __forceinline__ float quad_bcast_dyn(float v, int i)
{
    switch (i) {
    case 0: return quad_bcast<0>(v);
    case 1: return quad_bcast<1>(v);
    case 2: return quad_bcast<2>(v);
    default: return quad_bcast<3>(v);
    }
}

// ---------------------------------------------------------------- small vector helpers
// Arithmetic with a quad-broadcast operand folded into the instruction (VOP2 + DPP quad_perm): bcast<I>(a) OP b in
// ONE instruction instead of a v_mov_dpp plus the operation.  The hot per-pixel evaluation of a head-queue candidate
// consumes fifteen values that one lane of the quad fetched for all four; the compiler's DPP combiner leaves the
// broadcasts as separate moves, so they are written out here.  Same rounding as the plain forms (v_fmac is the fused
// multiply-add).  Callers make sure all four lanes of the quad are active and that `a` was not written by a VALU
// instruction in the two preceding issue slots (DPP read hazard): the callers' sources come from memory loads and
// the first use is preceded by `dpp_hazard_guard()`.
__device__ __forceinline__ void dpp_hazard_guard() { asm volatile("s_nop 1"); }
// The same for a value the caller has just computed with a VALU instruction: the guard takes the register in and hands
// it out again, so the write cannot sink below it and every later read is ordered after it.
__device__ __forceinline__ void dpp_hazard_guard_on(float& v) { asm volatile("s_nop 1" : "+v"(v)); }
#define STP_DPP_OP3(NAME, INSTR)                                                                                              \
    template <int I> __device__ __forceinline__ float NAME(float a, float b)                                                  \
    {                                                                                                                         \
        float r;                                                                                                              \
        if constexpr (I == 0) asm(INSTR " %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b)); \
        else if constexpr (I == 1) asm(INSTR " %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b)); \
        else if constexpr (I == 2) asm(INSTR " %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b)); \
        else asm(INSTR " %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b));  \
        return r;                                                                                                             \
    }
STP_DPP_OP3(quad_mul, "v_mul_f32_dpp") // bcast<I>(a) * b
STP_DPP_OP3(quad_sub, "v_sub_f32_dpp") // bcast<I>(a) - b
#undef STP_DPP_OP3
template <int I> __device__ __forceinline__ float quad_fma(float a, float b, float acc) // fma(bcast<I>(a), b, acc)
{
    if constexpr (I == 0) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (I == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (I == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

// acc += partner(a) * b in one instruction, the partner lane given by a DPP control: 0xB1 = lane ^ 1, 0x4E = lane ^ 2
// (quad_perm), 0x141 = mirror inside the 8-lane half, 0x140 = mirror inside the 16-lane row.  Same caveats as above.
template <int CTRL> __device__ __forceinline__ float partner_fma(float a, float b, float acc)
{
    if constexpr (CTRL == 0xB1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (CTRL == 0x4E) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (CTRL == 0x141) asm("v_fmac_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_fmac_f32_dpp %0, %1, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

struct Mat3 { float m[3][3]; }; // m[c][r]: column c, row r (same storage convention as the reference's vector library)

__device__ __forceinline__ Mat3 mat_mul(const Mat3& a, const Mat3& b)
{
#pragma clang fp contract(off)
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[0][j] * b.m[i][0] + a.m[1][j] * b.m[i][1] + a.m[2][j] * b.m[i][2];
    return r;
}
__device__ __forceinline__ Mat3 mat_transpose(const Mat3& a)
{
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
__device__ __forceinline__ Mat3 mat_diag(float a, float b, float c)
{
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = 0.0f;
    r.m[0][0] = a; r.m[1][1] = b; r.m[2][2] = c;
    return r;
}

// Rotation matrix from quaternion (r,x,y,z), the nine terms laid out as in reference
// forward_common.h:158-169 (first three terms form column 0).
__device__ __forceinline__ Mat3 quat_to_mat(float4 q)
{
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    Mat3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z);       R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y);       R.m[2][1] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } // NaN -> 0

// reference auxiliary.h:66-69 (double arithmetic because of the double literals there)
__device__ __forceinline__ float ndc_to_pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// reference auxiliary.h:91-101 + our tile-row window
__device__ __forceinline__ void get_rect(float2 p, float2 ext, int gx, int gy, int ty0, int ty1, int& x0, int& y0, int& x1, int& y1)
{
#pragma clang fp contract(off)
    x0 = min(gx, max(0, (int)floorf((p.x - ext.x) / 16.0f)));
    y0 = min(gy, max(0, (int)floorf((p.y - ext.y) / 16.0f)));
    x1 = min(gx, max(0, (int)ceilf((p.x + ext.x) / 16.0f)));
    y1 = min(gy, max(0, (int)ceilf((p.y + ext.y) / 16.0f)));
    y0 = max(y0, ty0);
    y1 = min(y1, ty1);
    if (y1 < y0) y1 = y0;
}

// ln(x) rounded to fp32 from the double-precision logarithm: the correctly rounded value (up to double rounding, ~1e-8 of
// the arguments), which is also what the oracle computes -- two fp32 logf implementations differ by an ulp now and then,
// and behind this logarithm sits a ceil() that turns the ulp into a radius (tight_opacity_bounding).  Per Gaussian, not
// per pixel: the cost does not show.
__device__ __forceinline__ float log_rounded(float x) { return (float)log((double)x); }

// 1/x, bit for bit the IEEE quotient the compiler's 12-instruction division sequence returns, in three
// instructions: v_rcp_f32 plus one Newton step is correctly rounded for every 2^-126 <= |x| < 2^126
// (tools/numerics_probe.hip checks all 2.1e9 positive normal floats on the device: the only mismatches are the
// two top binades, where 1/x is subnormal).  Outside that domain (never seen; NaN and 0 included) the whole
// wave takes the division instead, so the result is the same everywhere.
// FAST = true leaves the domain check out: for callers that KNOW 2^-126 <= |x| < 2^126 (the depth keys of a frame
// whose Sigma^-1 entries are all below 1e36 -- preprocess_kernel reports anything else in the status word and stp_forward
// then launches the kernels built with the check).  The check is a compare plus a branch per evaluation, and the branch
// splits the head level's straight-line step into basic blocks: -4 % on the forward without it.
template <bool FAST = false> __device__ __forceinline__ float rcp_ieee(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    if constexpr (!FAST) {
        const float ax = fabsf(x);
        if (__builtin_expect(__any(!(ax >= 1.17549435e-38f && ax < 8.5e37f)), 0)) r = 1.0f / x;
    }
    return r;
}

// exp(x) for the blend weights, x <= 0 (results for x > 0 are discarded by the callers): 2^(x log2 e) with the
// rounding error of the product x*log2(e) fed back to first order -- six instructions instead of the library's
// fourteen.  Measured over every float in [-16, 0) against double precision (tools/numerics_probe.hip):
// 88% of the results within 0.5 ulp, all within 2 ulp (library expf: 92% / all within 1 ulp; the reference's
// CUDA expf is specified to 2 ulp).
__device__ __forceinline__ float exp_blend(float x)
{
    // (written with -y: the error term is then a multiply-add with a literal, a VOP2 instruction; as fma(x, c, -y) it
    // needs the VOP3 form, whose constant sits in an SGPR -- twice the issue cost on gfx950, tools/valu_rate_bench.hip)
    const float ny = x * -1.44269502162933349609375f;
    float r = fmaf(x, 1.44269502162933349609375f, ny);
    r = fmaf(x, 1.925963033500011e-8f, r);
    const float g = __builtin_amdgcn_exp2f(-ny);
    return fmaf(g, r * 0.693147182464599609375f, g);
}

// Depth of the point of maximum contribution along a view ray (reference stopthepop_common.cuh:44-55).
// p0 = [S00 S01 S02], p1 = [S11 S12 S22], p2 = Sigma^-1 (mu - cam).  The reciprocal is the IEEE quotient 1/x in both forms.
#ifndef STP_IEEE_DEPTH
#define STP_IEEE_DEPTH 1 // 1 (the default since round 4): the reference's expression with NO contraction, every product and sum rounded
                         // on its own in the order stopthepop_common.cuh:47-51 writes them -- what the -ffp-contract=off build of the
                         // reference computes.  With it the sort keys, tile lists and per-pixel orders are the reference's bit for bit
                         // (tests/test_reference_pin.py).  0 (`make FMA_DEPTH=1` -> libstp_raster_fma.so): every dot product is
                         // fma(c, z, fma(b, y, a*x)), ten VALU instructions fewer per evaluation (forward render -3.7 % at C2-full);
                         // keys then sit an ulp off the reference's now and then and list neighbours swap.
#endif
template <bool FAST = false> __device__ __forceinline__ float depth_along_ray(float3 p0, float3 p1, float3 p2, float3 v)
{
#if STP_IEEE_DEPTH
    {
#pragma clang fp contract(off)
        const float b0 = (p0.x * v.x + p0.y * v.y) + p0.z * v.z;
        const float b1 = (p0.y * v.x + p1.x * v.y) + p1.y * v.z;
        const float b2 = (p0.z * v.x + p1.y * v.y) + p1.z * v.z;
        const float n = (p2.x * v.x + p2.y * v.y) + p2.z * v.z;
        const float d = (b0 * v.x + b1 * v.y) + b2 * v.z;
        return n * rcp_ieee<FAST>(fmaxf(0.00001f, d));
    }
#endif
    const float a0 = fmaf(p0.z, v.z, fmaf(p0.y, v.y, p0.x * v.x));
    const float a1 = fmaf(p1.y, v.z, fmaf(p1.x, v.y, p0.y * v.x));
    const float a2 = fmaf(p1.z, v.z, fmaf(p1.y, v.y, p0.z * v.x));
    const float num = fmaf(p2.z, v.z, fmaf(p2.y, v.y, p2.x * v.x));
    const float den = fmaf(a2, v.z, fmaf(a1, v.y, a0 * v.x));
    const float rcp = rcp_ieee<FAST>(fmaxf(0.00001f, den));
    return num * rcp;
}

// fminf(0.99f, x) for an x that comes out of inline asm: one v_min_f32 (the compiler, not knowing that x is not a
// signalling NaN, would canonicalise it first with a v_max_f32 x, x).  NaN -> 0.99 like fminf.
__device__ __forceinline__ float min_099(float x)
{
    float r;
    asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(r) : "v"(x));
    return r;
}

// depth_along_ray() for lane I's packed Sigma^-1 rows (c0 = [S00 S01 S02], c1 = [S11 S12 S22], c2 = Sigma^-1 (mu - cam)),
// every product taking its broadcast operand through DPP: the same operations in the same order, bit for bit.
template <int I, bool FAST = false> __device__ __forceinline__ float depth_along_ray_quad(float4 c0, float4 c1, float4 c2, float3 v)
{
#if STP_IEEE_DEPTH
    {
#pragma clang fp contract(off)
        const float b0 = (quad_mul<I>(c0.x, v.x) + quad_mul<I>(c0.y, v.y)) + quad_mul<I>(c0.z, v.z);
        const float b1 = (quad_mul<I>(c0.y, v.x) + quad_mul<I>(c1.x, v.y)) + quad_mul<I>(c1.y, v.z);
        const float b2 = (quad_mul<I>(c0.z, v.x) + quad_mul<I>(c1.y, v.y)) + quad_mul<I>(c1.z, v.z);
        const float n = (quad_mul<I>(c2.x, v.x) + quad_mul<I>(c2.y, v.y)) + quad_mul<I>(c2.z, v.z);
        const float d = (b0 * v.x + b1 * v.y) + b2 * v.z;
        return n * rcp_ieee<FAST>(fmaxf(0.00001f, d));
    }
#endif
    const float a0 = quad_fma<I>(c0.z, v.z, quad_fma<I>(c0.y, v.y, quad_mul<I>(c0.x, v.x)));
    const float a1 = quad_fma<I>(c1.y, v.z, quad_fma<I>(c1.x, v.y, quad_mul<I>(c0.y, v.x)));
    const float a2 = quad_fma<I>(c1.z, v.z, quad_fma<I>(c1.y, v.y, quad_mul<I>(c0.z, v.x)));
    const float num = quad_fma<I>(c2.z, v.z, quad_fma<I>(c2.y, v.y, quad_mul<I>(c2.x, v.x)));
    const float den = fmaf(a2, v.z, fmaf(a1, v.y, a0 * v.x));
    const float rcp = rcp_ieee<FAST>(fmaxf(0.00001f, den));
    return num * rcp;
}

// ---- the same contraction on the MATRIX pipe -----------------------------------------------------------------------------
// v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4x4 outer products D[r][j] = A[r] * B[j] + C[r][j], one block per aligned
// group of four lanes: lane (quad b, j) supplies A = "its" row value and B = its column value and receives, in the four
// registers of D, row r = 0..3 of ITS column -- i.e. with A = the record word that lane r of the quad fetched and B = my pixel's
// ray component, D[r] = (entry r's word) * (my ray component) + C[r]: the quad broadcast and the multiply-add of
// depth_along_ray_quad_ent() for all four candidates of a head group in ONE instruction that issues beside the VALU stream.
// Every element is an fmaf (one rounding; tools/mfma_probe.hip checks layout and bits on the device), and the three chained
// instructions of a dot product are exactly fma(c, z, fma(b, y, a * x)) -- the canonical order of depth_along_ray().
typedef float stp_f4 __attribute__((ext_vector_type(4)));
struct QuadDepthTerms { stp_f4 a0, a1, a2, num; }; // [r]: candidate r of the quad, at my pixel
__device__ __forceinline__ QuadDepthTerms depth_terms_quad_mfma(float4 A, float4 B, float4 C, float3 v)
{
    const stp_f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    QuadDepthTerms t;
    t.a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.x, v.x, z, 0, 0, 0);
    t.a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.y, v.x, z, 0, 0, 0);
    t.a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.z, v.x, z, 0, 0, 0);
    t.num = __builtin_amdgcn_mfma_f32_4x4x1f32(B.z, v.x, z, 0, 0, 0);
    t.a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.y, v.y, t.a0, 0, 0, 0);
    t.a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.w, v.y, t.a1, 0, 0, 0);
    t.a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(B.x, v.y, t.a2, 0, 0, 0);
    t.num = __builtin_amdgcn_mfma_f32_4x4x1f32(B.w, v.y, t.num, 0, 0, 0);
    t.a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A.z, v.z, t.a0, 0, 0, 0);
    t.a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(B.x, v.z, t.a1, 0, 0, 0);
    t.a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(B.y, v.z, t.a2, 0, 0, 0);
    t.num = __builtin_amdgcn_mfma_f32_4x4x1f32(C.x, v.z, t.num, 0, 0, 0);
    return t;
}
template <int I, bool FAST = false> __device__ __forceinline__ float depth_from_terms(const QuadDepthTerms& t, float3 v)
{
    const float den = fmaf(t.a2[I], v.z, fmaf(t.a1[I], v.y, t.a0[I] * v.x));
    const float rcp = rcp_ieee<FAST>(fmaxf(0.00001f, den));
    return t.num[I] * rcp;
}

// The same for an entry record (BinningState: A = (S00 S01 S02 S11), B = (S12 S22 q.x q.y), C = (q.z . . .)).
template <int I, bool FAST = false> __device__ __forceinline__ float depth_along_ray_quad_ent(float4 A, float4 B, float4 C, float3 v)
{
#if STP_IEEE_DEPTH
    {
#pragma clang fp contract(off)
        const float b0 = (quad_mul<I>(A.x, v.x) + quad_mul<I>(A.y, v.y)) + quad_mul<I>(A.z, v.z);
        const float b1 = (quad_mul<I>(A.y, v.x) + quad_mul<I>(A.w, v.y)) + quad_mul<I>(B.x, v.z);
        const float b2 = (quad_mul<I>(A.z, v.x) + quad_mul<I>(B.x, v.y)) + quad_mul<I>(B.y, v.z);
        const float n = (quad_mul<I>(B.z, v.x) + quad_mul<I>(B.w, v.y)) + quad_mul<I>(C.x, v.z);
        const float d = (b0 * v.x + b1 * v.y) + b2 * v.z;
        return n * rcp_ieee<FAST>(fmaxf(0.00001f, d));
    }
#endif
    const float a0 = quad_fma<I>(A.z, v.z, quad_fma<I>(A.y, v.y, quad_mul<I>(A.x, v.x)));
    const float a1 = quad_fma<I>(B.x, v.z, quad_fma<I>(A.w, v.y, quad_mul<I>(A.y, v.x)));
    const float a2 = quad_fma<I>(B.y, v.z, quad_fma<I>(B.x, v.y, quad_mul<I>(A.z, v.x)));
    const float num = quad_fma<I>(C.x, v.z, quad_fma<I>(B.w, v.y, quad_mul<I>(B.z, v.x)));
    const float den = fmaf(a2, v.z, fmaf(a1, v.y, a0 * v.x));
    const float rcp = rcp_ieee<FAST>(fmaxf(0.00001f, den));
    return num * rcp;
}
// depth_along_ray() on an entry record
template <bool FAST = false> __device__ __forceinline__ float depth_along_ray_ent(float4 A, float4 B, float4 C, float3 v)
{
    return depth_along_ray<FAST>(make_float3(A.x, A.y, A.z), make_float3(A.w, B.x, B.y), make_float3(B.z, B.w, C.x), v);
}

__device__ __forceinline__ float3 f4_xyz(float4 a) { return make_float3(a.x, a.y, a.z); }

// reference auxiliary.h:71-81 and stopthepop_common.cuh:68-74; normalize(v) = v * (1/sqrt(v.v))
__device__ __forceinline__ float3 view_ray(const float* __restrict__ inv, float3 cam, float px, float py, int W, int H)
{
#pragma clang fp contract(off)
    const float ndcx = px * (2.0f / (float)W) - 1.0f;
    const float ndcy = py * (2.0f / (float)H) - 1.0f;
    float p[4];
#pragma unroll
    for (int j = 0; j < 4; j++) p[j] = (inv[j] * ndcx + inv[4 + j] * ndcy) + inv[12 + j];
    const float rcp_w = 1.0f / p[3];
    const float dx = p[0] * rcp_w - cam.x, dy = p[1] * rcp_w - cam.y, dz = p[2] * rcp_w - cam.z;
    const float s = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    return make_float3(dx * s, dy * s, dz * s);
}

// reference stopthepop_common.cuh:76-79
__device__ __forceinline__ float opacity_factor(float dx, float dy, float4 co)
{
#pragma clang fp contract(off)
    return 0.5f * (co.x * dx * dx + co.z * dy * dy) + co.y * dx * dy;
}

// The exponent of a blend weight, -(0.5 (a dx^2 + c dy^2) + b dx dy), in ONE canonical evaluation order without
// contraction, used by every forward and backward kernel: the weight decides "alpha < 1/255 -> skip", and a forward and a
// backward that round it differently disagree about which entries a pixel blended (the backward then divides the wrong
// transmittance chain).  Same roundings as opacity_factor(), hence as the oracle's expression.
__device__ __forceinline__ float blend_power(float dx, float dy, float4 co)
{
#pragma clang fp contract(off)
    return -(0.5f * (co.x * dx * dx + co.z * dy * dy) + co.y * dx * dy);
}
// The same with the conic of quad lane I as DPP operands.
template <int I> __device__ __forceinline__ float blend_power_quad(float dx, float dy, float4 co)
{
#pragma clang fp contract(off)
    return -(0.5f * (quad_mul<I>(co.x, dx) * dx + quad_mul<I>(co.z, dy) * dy) + quad_mul<I>(co.y, dx) * dy);
}

// Smallest "power" (largest contribution) a Gaussian reaches inside an axis-aligned pixel rectangle
// and where (reference stopthepop_common.cuh:130-174).  patch = rect size - 1.
// (rcp_x, rcp_y: 1 / (patch_w^2 a) and 1 / (patch_h^2 c) -- they depend on the Gaussian and the patch size only, so a caller that
// tests one Gaussian against several rectangles of one size computes them once)
__device__ __forceinline__ float max_contrib_power_rect_r(float4 co, float2 mean, float2 rmin, float2 rmax,
                                                          float patch_w, float patch_h, float rcp_x, float rcp_y, float2& max_pos)
{
#pragma clang fp contract(off)
    const float x_min_diff = rmin.x - mean.x;
    const float x_left = x_min_diff > 0.0f ? 1.0f : 0.0f;
    const float not_in_x = x_left + (mean.x > rmax.x ? 1.0f : 0.0f);
    const float y_min_diff = rmin.y - mean.y;
    const float y_above = y_min_diff > 0.0f ? 1.0f : 0.0f;
    const float not_in_y = y_above + (mean.y > rmax.y ? 1.0f : 0.0f);
    max_pos = mean;
    float power = 0.0f;
    if ((not_in_y + not_in_x) > 0.0f) {
        const float px = x_left * rmin.x + (1.0f - x_left) * rmax.x;
        const float py = y_above * rmin.y + (1.0f - y_above) * rmax.y;
        const float dx = copysignf(patch_w, x_min_diff);
        const float dy = copysignf(patch_h, y_min_diff);
        const float diffx = mean.x - px, diffy = mean.y - py;
        const float tx = not_in_y * saturatef((dx * co.x * diffx + dx * co.y * diffy) * rcp_x);
        const float ty = not_in_x * saturatef((dy * co.y * diffx + dy * co.z * diffy) * rcp_y);
        max_pos = make_float2(px + tx * dx, py + ty * dy);
        power = opacity_factor(mean.x - max_pos.x, mean.y - max_pos.y, co);
    }
    return power;
}
__device__ __forceinline__ float max_contrib_power_rect(float4 co, float2 mean, float2 rmin, float2 rmax,
                                                        float patch_w, float patch_h, float2& max_pos)
{
#pragma clang fp contract(off)
    const float x_min_diff = rmin.x - mean.x;
    const float x_left = x_min_diff > 0.0f ? 1.0f : 0.0f;
    const float not_in_x = x_left + (mean.x > rmax.x ? 1.0f : 0.0f);
    const float y_min_diff = rmin.y - mean.y;
    const float y_above = y_min_diff > 0.0f ? 1.0f : 0.0f;
    const float not_in_y = y_above + (mean.y > rmax.y ? 1.0f : 0.0f);
    max_pos = mean;
    float power = 0.0f;
    if ((not_in_y + not_in_x) > 0.0f) {
        const float px = x_left * rmin.x + (1.0f - x_left) * rmax.x;
        const float py = y_above * rmin.y + (1.0f - y_above) * rmax.y;
        const float dx = copysignf(patch_w, x_min_diff);
        const float dy = copysignf(patch_h, y_min_diff);
        const float diffx = mean.x - px, diffy = mean.y - py;
        const float rcp_x = rcp_ieee(patch_w * patch_w * co.x);
        const float rcp_y = rcp_ieee(patch_h * patch_h * co.z);
        const float tx = not_in_y * saturatef((dx * co.x * diffx + dx * co.y * diffy) * rcp_x);
        const float ty = not_in_x * saturatef((dy * co.y * diffx + dy * co.z * diffy) * rcp_y);
        max_pos = make_float2(px + tx * dx, py + ty * dy);
        power = opacity_factor(mean.x - max_pos.x, mean.y - max_pos.y, co);
    }
    return power;
}

// Minimum over the rectangle [x0,x1] x [y0,y1] of offsets (pixel - mean) of q(dx,dy) = 0.5 (a dx^2 + c dy^2) + b dx dy,
// the negated blend exponent (co = (a, b, c, opacity)).  q is convex for a positive definite conic: the minimum is 0 when
// the mean lies inside, otherwise it is on the boundary -- on each of the four edges a one-dimensional parabola whose
// vertex is clamped to the edge.  Anything unexpected (a or c not positive, NaN) returns 0, i.e. "may contribute".
__device__ __forceinline__ float min_power_rect(float4 co, float x0, float x1, float y0, float y1)
{
    const float A = co.x, B = co.y, C = co.z;
    if (!(A > 0.0f && C > 0.0f)) return 0.0f;
    if (x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f) return 0.0f;
    auto q = [&](float dx, float dy) { return 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy; };
    auto on_x_edge = [&](float X) { return q(X, fminf(fmaxf(-B * X / C, y0), y1)); }; // dx = X fixed, dy free in [y0, y1]
    auto on_y_edge = [&](float Y) { return q(fminf(fmaxf(-B * Y / A, x0), x1), Y); };
    const float m = fminf(fminf(on_x_edge(x0), on_x_edge(x1)), fminf(on_y_edge(y0), on_y_edge(y1)));
    return m == m ? fmaxf(m, 0.0f) : 0.0f;
}

__device__ __forceinline__ uint64_t make_sort_key(uint32_t tile, float depth) // reference auxiliary.h:238-244
{
    return ((uint64_t)tile << 32) | (uint64_t)__float_as_uint(depth);
}


// ---- per-(entry, sub-tile) decisions that do not depend on anything a pixel knows, made by the entry gather ---------------
constexpr int TILE_PX = 16; // (== TILE of stp_internal.h)
// The hierarchical kernel's 4x4 culling (reference hierarchical_render.cuh:722-743) tests every entry of a tile's list against
// each of the tile's sixteen 4x4 sub-tiles: opacity * exp(-power at the sub-tile's point of maximum contribution) < 1/255.
// That is sixteen evaluations per entry whoever makes them -- in the render kernel, which is VALU-bound, they cost 0.034 ms per
// C2 frame (the four waves of a tile, four sub-tiles each, every batch); the ENTRY GATHER (stp_tilesort.hip, or gather_entries_kernel
// behind STP_SORT=radix) has the entry's mean and conic in registers anyway and waits for memory three quarters of the time (24 %
// VALU-busy before this): there the evaluations cost nothing measurable.  Same functions, same
// operands, same decisions bit for bit: bit 4 w + s of the mask = "the entry is NOT culled for sub-tile column s of
// sub-tile row w"; the render kernel's batch staging reads the mask instead of the entry's conic.
__device__ __forceinline__ uint32_t subtile_cull_mask(float4 co, float2 xy, int tile_x, int tile_y)
{
    const float rcp_x = rcp_ieee(3.0f * 3.0f * co.x), rcp_y = rcp_ieee(3.0f * 3.0f * co.z);
    uint32_t mask = 0;
#pragma unroll 1
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const float x0 = (float)(tile_x * TILE_PX + 4 * sx), y0 = (float)(tile_y * TILE_PX + 4 * w);
            float2 mp;
            const float p = max_contrib_power_rect_r(co, xy, make_float2(x0, y0), make_float2(x0 + 3.0f, y0 + 3.0f), 3.0f, 3.0f, rcp_x, rcp_y, mp);
            const bool cull = fminf(0.99f, co.w * exp_blend(-p)) < ALPHA_THRESHOLD;
            mask |= cull ? 0u : (1u << (4 * w + sx));
        }
    return mask;
}

// The k-buffer kernel's batch staging (stp_render_kbuf.hip) keeps an entry for a 4x4 sub-tile only if its alpha can reach 1/255
// somewhere in it (the exact minimum of the exponent's form over the rectangle plus a rounding margin: a bound of OURS, the
// reference has no such test).  Again sixteen evaluations per entry that do not depend on anything a pixel knows: made here.
// min_power_rect() for a caller that tests one conic against many rectangles: the two quotients -B X / C and -B Y / A become products
// with reciprocals taken once (v_rcp_f32, 1 ulp: the parabola's vertex moves by 1e-7 of itself, where the form is stationary, and the
// callers' margin covers far more).  A bound of OURS, not a reference decision.
__device__ __forceinline__ float min_power_rect_r(float4 co, float nb_over_c, float nb_over_a, float x0, float x1, float y0, float y1)
{
    const float A = co.x, B = co.y, C = co.z;
    if (!(A > 0.0f && C > 0.0f)) return 0.0f;
    if (x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f) return 0.0f;
    auto q = [&](float dx, float dy) { return 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy; };
    auto on_x_edge = [&](float X) { return q(X, fminf(fmaxf(nb_over_c * X, y0), y1)); }; // dx = X fixed, dy free in [y0, y1]
    auto on_y_edge = [&](float Y) { return q(fminf(fmaxf(nb_over_a * Y, x0), x1), Y); };
    const float m = fminf(fminf(on_x_edge(x0), on_x_edge(x1)), fminf(on_y_edge(y0), on_y_edge(y1)));
    return m == m ? fmaxf(m, 0.0f) : 0.0f;
}

__device__ __forceinline__ uint32_t subtile_keep_mask_kbuffer(float4 D, float2 xy, int tile_x, int tile_y)
{
    const float T3 = fabsf(D.x) + fabsf(D.y) + fabsf(D.z);
    const float nb_over_c = -D.y * __builtin_amdgcn_rcpf(D.z), nb_over_a = -D.y * __builtin_amdgcn_rcpf(D.x);
    uint32_t mask = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float y0 = (float)(tile_y * TILE_PX + 4 * w) - xy.y;
        const float fy = fmaxf(fabsf(y0), fabsf(y0 + 3.0f));
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const float x0 = (float)(tile_x * TILE_PX + 4 * sx) - xy.x;
            const float p = min_power_rect_r(D, nb_over_c, nb_over_a, x0, x0 + 3.0f, y0, y0 + 3.0f);
            // rounding of the per-pixel exponent against this one: at most a few ulp of the form's terms, all below T3 * far^2
            const float far = fmaxf(fmaxf(fabsf(x0), fabsf(x0 + 3.0f)), fy);
            const bool keep = !(D.w * __builtin_amdgcn_exp2f(fmaf(T3 * far * far, 2.0e-6f, -p) * 1.44269502162933349609375f) < ALPHA_THRESHOLD * 0.9999f);
            mask |= keep ? (1u << (4 * w + sx)) : 0u;
        }
    }
    return mask;
}

__device__ __forceinline__ uint32_t subtile_mask(int kind, float4 D, float2 xy, int tile_x, int tile_y)
{
    return kind == 1 ? subtile_cull_mask(D, xy, tile_x, tile_y) : subtile_keep_mask_kbuffer(D, xy, tile_x, tile_y);
}

// Workgroup-cooperative staging of BLOCK rows of ROWLEN floats (ROWLEN % 4 == 0) from global memory into LDS rows padded to
// ROWLEN + 1 words, for a FULL block: all of a thread's 16-byte loads are issued before the first LDS write (the generic
// loop with a run-time row length computes an integer division per iteration and ends up with ONE load in flight per
// thread: measured 3.0 TB/s for the SH rows; the unrolled form keeps ROWLEN / 4 in flight).
template <int BLOCK, int ROWLEN>
__device__ __forceinline__ void stage_rows_load(const float* __restrict__ src, int tid, float4 (&v)[ROWLEN / 4])
{
#pragma unroll
    for (int k = 0; k < ROWLEN / 4; k++) v[k] = reinterpret_cast<const float4*>(src)[tid + k * BLOCK];
}
template <int BLOCK, int ROWLEN>
__device__ __forceinline__ void stage_rows_store(float* __restrict__ s_rows, int tid, const float4 (&v)[ROWLEN / 4])
{
#pragma unroll
    for (int k = 0; k < ROWLEN / 4; k++) {
        const int f = 4 * (tid + k * BLOCK);
        const int r = f / ROWLEN, j = f - r * ROWLEN;
        float* d = s_rows + r * (ROWLEN + 1) + j;
        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
    }
}
template <int BLOCK, int ROWLEN>
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ src, float* __restrict__ s_rows, int tid)
{
    float4 v[ROWLEN / 4];
    stage_rows_load<BLOCK, ROWLEN>(src, tid, v);
    stage_rows_store<BLOCK, ROWLEN>(s_rows, tid, v);
}
template <int BLOCK, int ROWLEN>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ dst, const float* __restrict__ s_rows, int tid)
{
    constexpr int N = ROWLEN / 4;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int f = 4 * (tid + k * BLOCK);
        const int r = f / ROWLEN, j = f - r * ROWLEN;
        const float* d = s_rows + r * (ROWLEN + 1) + j;
        reinterpret_cast<float4*>(dst)[tid + k * BLOCK] = make_float4(d[0], d[1], d[2], d[3]);
    }
}

} // namespace stp
