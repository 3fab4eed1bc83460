/*
 * oracle/stp_oracle.cpp -- CPU restatement of the StopThePop rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see stp_oracle.h).  Pinned against the reference's own sources compiled for gfx950
 * (oracle/ref_build/, tests/golden/ref/, tests/test_reference_golden.py); see stp_oracle.h for what that covers.
 *
 * This is our own from-scratch code.  It restates, stage by stage, what the reference computes
 * (citations "ref:" are file:line under /root/reference/cuda_rasterizer unless a path is given).
 * It is written for clarity, not speed: per-Gaussian loops for the streaming stages, per-tile /
 * per-pixel loops for the blend stages, OpenMP over Gaussians and tiles.  It doubles as the
 * "naive host-CPU tile rasterizer" that bench.py times as a reported, non-target baseline.
 *
 * Numerics policy: fp32 throughout, compiled with -ffp-contract=off so that every a*b+c is two
 * IEEE operations.  The HIP kernels evaluate every ordering-critical quantity (view rays, depth
 * keys, rectangles, radii) with the same operation order and contraction disabled, so tile lists
 * and per-pixel blend orders are comparable bit-for-bit; only libm-level functions (exp, log)
 * may differ by an ulp between host and device.
 */
#include "stp_oracle.h"
#include "../include/stp_turbo_colormap.h" // data table only

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// constants (ref: auxiliary.h:18-46, config.h:15-19)
// ------------------------------------------------------------------------------------------------
constexpr int   TILE = 16;                       // BLOCK_X == BLOCK_Y == 16
constexpr float ALPHA_THRESHOLD = 1.0f / 255.0f; // also ALPHA_THRESHOLD_PADDED
constexpr float T_THRESHOLD = 0.0001f;
// Per-pixel blend decisions ONLY (alpha < 1/255: skip; test_T < 1e-4: stop): thresholds a checker may move by a few 1e-7 (orc_set_blend_nudge) to ask
// "is the product's output the reference algorithm's with a decision that sits ON its threshold taken the other way?" -- tests/test_gpu_parity.py,
// check_against_oracle.  The defaults are the reference's constants; preprocessing and tile culling never look at these.
static float g_blend_alpha_thr = ALPHA_THRESHOLD, g_blend_T_thr = T_THRESHOLD, g_cull_alpha_thr = ALPHA_THRESHOLD; // (the last one: the 4x4 sub-tile culling's own alpha test)
// ... and single decisions taken the other way: (pixel index << 32 | Gaussian id) of per-pixel alpha tests whose outcome is inverted (orc_set_forced_alpha_flips,
// sorted) -- a whole frame has dozens of alphas within any band around 1/255, moving the threshold for all of them is too blunt there
static std::vector<uint64_t> g_forced_alpha;
static std::vector<uint32_t> g_forced_T; // pixel indices (sorted) whose "test_T < 1e-4: stop" decision is inverted at the step where test_T sits within 2e-6 (relative) of the threshold
static inline bool T_stops(float test_T, int W, int px, int py)
{
    bool stop = test_T < g_blend_T_thr;
    if (!g_forced_T.empty() && std::fabs(test_T / T_THRESHOLD - 1.0f) <= 2e-6f &&
        std::binary_search(g_forced_T.begin(), g_forced_T.end(), (uint32_t)((size_t)py * (size_t)W + (size_t)px)))
        stop = !stop;
    return stop;
}
static inline bool alpha_skips(float alpha, int W, int px, int py, int id)
{
    bool skip = alpha < g_blend_alpha_thr;
    if (!g_forced_alpha.empty()) {
        const uint64_t key = ((uint64_t)((size_t)py * (size_t)W + (size_t)px) << 32) | (uint64_t)(uint32_t)id;
        if (std::binary_search(g_forced_alpha.begin(), g_forced_alpha.end(), key)) skip = !skip;
    }
    return skip;
}
constexpr uint32_t INVALID_TILE = 0xFFFFFFFFu;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                            -1.0925484305920792f, 0.5462742152960396f};
constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                            0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                            -0.5900435899266435f};

enum SortMode { GLOBAL = 0, PPX_FULL = 1, PPX_KBUFFER = 2, HIER = 3 };
enum SortOrder { Z_DEPTH = 0, DISTANCE = 1, PTD_CENTER = 2, PTD_MAX = 3 };

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

// 3x3 matrix stored column-major, c[col][row], with the same product convention as the vector
// library the reference uses ((A*B)[i][j] = sum_k A[k][j]*B[i][k], k summed left to right).
struct M3 { float c[3][3]; };

inline M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.c[i][j] = a.c[0][j] * b.c[i][0] + a.c[1][j] * b.c[i][1] + a.c[2][j] * b.c[i][2];
    return r;
}
inline M3 transpose(const M3& a)
{
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.c[i][j] = a.c[j][i];
    return r;
}
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length3(V3 a) { return sqrtf(dot3(a, a)); }
inline float saturate(float x) { return (x != x) ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x)); }
inline float frcp(float x) { return 1.0f / x; } // round-to-nearest reciprocal
// ln(x) rounded to fp32 from the double-precision logarithm (the reference calls logf, stopthepop_common.cuh:473,553, whose
// CUDA implementation is within 1 ulp; the correctly rounded value is the one any two implementations can agree on -- the
// HIP path computes it the same way, stp_device.h: log_rounded)
inline float log_rounded(float x) { return (float)std::log((double)x); }

// ------------------------------------------------------------------------------------------------
// forward state
// ------------------------------------------------------------------------------------------------
} // namespace

struct OrcFrame {
    int P = 0, D = 0, M = 0, W = 0, H = 0, R = 0, gx = 0, gy = 0;
    OrcSettings s{};
    bool has_inv = false;
    std::vector<float> depths, rects2D, means2D, cov3D, cov3D_inv, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    std::vector<uint64_t> keys_unsorted, keys;
    std::vector<uint32_t> values_unsorted, point_list;
    std::vector<uint32_t> ranges; // 2 per tile
    std::vector<float> final_T;
    std::vector<uint32_t> n_contrib;
};

namespace {

inline bool requires_depth_along_ray(const OrcSettings& s) // ref: rasterizer.h:66-71
{
    return s.sort_mode != GLOBAL || s.sort_order == PTD_CENTER || s.sort_order == PTD_MAX;
}

// ref: rasterizer_impl.cu:37-52 -- smallest "bit" with 2^bit > n found by bisection on the MSB.
uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// ------------------------------------------------------------------------------------------------
// leaf math
// ------------------------------------------------------------------------------------------------

// ref: auxiliary.h:66-69 (evaluated in double because the literals there are double)
inline float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// ref: auxiliary.h:91-101, plus our tile-row window [ty0, ty1) for sharding
inline void get_rect(V2 p, V2 ext, int gx, int gy, int ty0, int ty1, int& x0, int& y0, int& x1, int& y1)
{
    x0 = std::min(gx, std::max(0, (int)floorf((p.x - ext.x) / (float)TILE)));
    y0 = std::min(gy, std::max(0, (int)floorf((p.y - ext.y) / (float)TILE)));
    x1 = std::min(gx, std::max(0, (int)ceilf((p.x + ext.x) / (float)TILE)));
    y1 = std::min(gy, std::max(0, (int)ceilf((p.y + ext.y) / (float)TILE)));
    y0 = std::max(y0, ty0);
    y1 = std::min(y1, ty1);
    if (y1 < y0) y1 = y0;
}

// rotation matrix exactly as the reference builds it (ref: forward_common.h:158-169,
// stopthepop_common.cuh:24-35): nine values handed to a column-major constructor.
inline M3 quat_matrix(const float* q)
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    M3 R;
    R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z);       R.c[0][2] = 2.f * (x * z + r * y);
    R.c[1][0] = 2.f * (x * y + r * z);       R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
    R.c[2][0] = 2.f * (x * z - r * y);       R.c[2][1] = 2.f * (y * z + r * x);       R.c[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

inline M3 diag(float a, float b, float c)
{
    M3 S;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S.c[i][j] = 0.0f;
    S.c[0][0] = a; S.c[1][1] = b; S.c[2][2] = c;
    return S;
}

// ref: forward_common.h:149-183
inline void compute_cov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    M3 S = diag(mod * scale[0], mod * scale[1], mod * scale[2]);
    M3 R = quat_matrix(rot);
    M3 Mm = mul(S, R);
    M3 Sigma = mul(transpose(Mm), Mm);
    cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

// ref: stopthepop_common.cuh:13-41
inline M3 compute_inv_cov3D(const float* scale, const float* rot, float mod)
{
    M3 S = diag(1.f / (mod * std::max(1e-3f, scale[0])), 1.f / (mod * std::max(1e-3f, scale[1])),
                1.f / (mod * std::max(1e-3f, scale[2])));
    M3 R = quat_matrix(rot);
    M3 Mm = mul(S, R);
    return mul(transpose(Mm), Mm);
}

// ref: forward_common.h:73-106.  view = 16 floats, element [4*col+row].
inline M3 compute_cov2D_full(V3 t, float fx, float fy, float tan_fovx, float tan_fovy, const float* cov3D,
                             const float* view, M3* T_out = nullptr)
{
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
    t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
    M3 J;
    J.c[0][0] = fx / t.z; J.c[0][1] = 0.0f;     J.c[0][2] = -(fx * t.x) / (t.z * t.z);
    J.c[1][0] = 0.0f;     J.c[1][1] = fy / t.z; J.c[1][2] = -(fy * t.y) / (t.z * t.z);
    J.c[2][0] = 0.0f;     J.c[2][1] = 0.0f;     J.c[2][2] = 0.0f;
    M3 Wv; // upper-left 3x3 of the view matrix (columns = view[4*i + j]), then transposed
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Wv.c[i][j] = view[4 * i + j];
    M3 Wm = transpose(Wv);
    M3 T = mul(Wm, J);
    M3 Vrk;
    Vrk.c[0][0] = cov3D[0]; Vrk.c[0][1] = cov3D[1]; Vrk.c[0][2] = cov3D[2];
    Vrk.c[1][0] = cov3D[1]; Vrk.c[1][1] = cov3D[3]; Vrk.c[1][2] = cov3D[4];
    Vrk.c[2][0] = cov3D[2]; Vrk.c[2][1] = cov3D[4]; Vrk.c[2][2] = cov3D[5];
    if (T_out) *T_out = T;
    return mul(mul(transpose(T), transpose(Vrk)), T);
}

// ref: stopthepop_common.cuh:44-55.  pack = 3 x float4: [S00 S01 S02 .][S11 S12 S22 .][u0 u1 u2 .]
// Two evaluation orders, both shared with the HIP kernels so that depth keys compare bit-for-bit: the reference's
// expression with every product and sum rounded on its own (the default, g_ieee_depth = 1), and the fma chains of
// libstp_raster_fma.so (every dot product fma(c, z, fma(b, y, a*x))); the reciprocal is the IEEE quotient 1/x in
// both.  (The CUDA reference lets nvcc contract these sums as it likes.)
int g_lazy_pop = 0;   // test-only switch "lazy_pop": a candidate that FAILS its tests is not shown to the pixel at all (no
                      // pop-before-look for it) -- the claim the wave64 kernels rest on (stp_render_hier.inc filter_push,
                      // stp_render_kbuf.hip) is that this changes no image, no final_T and no gradient; tests/test_oracle_cpu.py
                      // holds the oracle to it bit for bit.  (n_contrib of the k-buffer counts looked-at entries and does change.)
int g_ieee_depth = 1; // switch "ieee_depth", default 1: the reference's expression with NO contraction (every product and
                      // sum rounded separately, in the order stopthepop_common.cuh:47-51 writes them) -- what the
                      // -ffp-contract=off build of the reference itself (oracle/_ref/libstp_ref_ieee.so) computes, and what
                      // the default product library computes since round 4.  0: the fma order of libstp_raster_fma.so.
inline float depth_along_ray(const float* pk, V3 v)
{
    if (g_ieee_depth) {
        const float b0 = (pk[0] * v.x + pk[1] * v.y) + pk[2] * v.z;
        const float b1 = (pk[1] * v.x + pk[4] * v.y) + pk[5] * v.z;
        const float b2 = (pk[2] * v.x + pk[5] * v.y) + pk[6] * v.z;
        const float n = (pk[8] * v.x + pk[9] * v.y) + pk[10] * v.z;
        const float d = (b0 * v.x + b1 * v.y) + b2 * v.z;
        return n * frcp(std::max(0.00001f, d));
    }
    const float a0 = fmaf(pk[2], v.z, fmaf(pk[1], v.y, pk[0] * v.x));
    const float a1 = fmaf(pk[5], v.z, fmaf(pk[4], v.y, pk[1] * v.x));
    const float a2 = fmaf(pk[6], v.z, fmaf(pk[5], v.y, pk[2] * v.x));
    const float num = fmaf(pk[10], v.z, fmaf(pk[9], v.y, pk[8] * v.x));
    const float den = fmaf(a2, v.z, fmaf(a1, v.y, a0 * v.x));
    const float rcp_den = frcp(std::max(0.00001f, den));
    return num * rcp_den;
}

// ref: auxiliary.h:71-81.  inv = 16 floats; "column" i = inv[4*i .. 4*i+3].
inline V3 pix2world(V2 pix, int W, int H, const float* inv)
{
    const float ndcx = pix.x * (2.0f / (float)W) - 1.0f;
    const float ndcy = pix.y * (2.0f / (float)H) - 1.0f;
    float p[4];
    for (int j = 0; j < 4; j++) p[j] = (inv[0 + j] * ndcx + inv[4 + j] * ndcy) + inv[12 + j];
    const float rcp_w = frcp(p[3]);
    return {p[0] * rcp_w, p[1] * rcp_w, p[2] * rcp_w};
}

// ref: stopthepop_common.cuh:68-74.  normalize(v) is evaluated as v * (1/sqrt(v.v)).
inline V3 view_ray(const float* inv, V3 cam, V2 pix, int W, int H)
{
    const V3 pw = pix2world(pix, W, H, inv);
    const V3 d = {pw.x - cam.x, pw.y - cam.y, pw.z - cam.z};
    const float s = 1.0f / sqrtf(dot3(d, d));
    return {d.x * s, d.y * s, d.z * s};
}

// ref: stopthepop_common.cuh:76-79
inline float opacity_factor(float dx, float dy, const float* co)
{
    return 0.5f * (co[0] * dx * dx + co[2] * dy * dy) + co[1] * dx * dy;
}

// ref: stopthepop_common.cuh:130-174 (the branch-free float variant, the only one that is called)
inline float max_contrib_power_rect(const float* co, V2 mean, V2 rmin, V2 rmax, float patch_w, float patch_h,
                                    V2& max_pos)
{
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
        const float rcp_x = frcp(patch_w * patch_w * co[0]);
        const float rcp_y = frcp(patch_h * patch_h * co[2]);
        const float tx = not_in_y * saturate((dx * co[0] * diffx + dx * co[1] * diffy) * rcp_x);
        const float ty = not_in_x * saturate((dy * co[1] * diffx + dy * co[2] * diffy) * rcp_y);
        max_pos = {px + tx * dx, py + ty * dy};
        power = opacity_factor(mean.x - max_pos.x, mean.y - max_pos.y, co);
    }
    return power;
}

// ref: forward_common.h:20-70
inline void color_from_sh(int idx, int deg, int M, V3 mean, V3 cam, const float* shs, uint8_t* clamped, float* rgb)
{
    V3 dir = {mean.x - cam.x, mean.y - cam.y, mean.z - cam.z};
    const float len = length3(dir);
    dir = {dir.x / len, dir.y / len, dir.z / len};
    const float* sh = shs + (size_t)idx * M * 3;
    const float x = dir.x, y = dir.y, z = dir.z;
    for (int ch = 0; ch < 3; ch++) {
        auto c = [&](int k) { return sh[3 * k + ch]; };
        float r = SH_C0 * c(0);
        if (deg > 0) {
            r = r - (SH_C1 * y) * c(1) + (SH_C1 * z) * c(2) - (SH_C1 * x) * c(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + (SH_C2[0] * xy) * c(4) + (SH_C2[1] * yz) * c(5) + (SH_C2[2] * (2.0f * zz - xx - yy)) * c(6) +
                    (SH_C2[3] * xz) * c(7) + (SH_C2[4] * (xx - yy)) * c(8);
                if (deg > 2) {
                    r = r + (SH_C3[0] * y * (3.0f * xx - yy)) * c(9) + (SH_C3[1] * xy * z) * c(10) +
                        (SH_C3[2] * y * (4.0f * zz - xx - yy)) * c(11) +
                        (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * c(12) +
                        (SH_C3[4] * x * (4.0f * zz - xx - yy)) * c(13) + (SH_C3[5] * z * (xx - yy)) * c(14) +
                        (SH_C3[6] * x * (xx - 3.0f * yy)) * c(15);
                }
            }
        }
        r += 0.5f;
        clamped[3 * idx + ch] = (r < 0.0f) ? 1 : 0;
        rgb[3 * idx + ch] = std::max(r, 0.0f);
    }
}

// ref: stopthepop_common.cuh:176-262 (sequential form; load balancing changes no result)
inline int tbc_tile_count(const float* co, V2 xy, float thr, int x0, int y0, int x1, int y1)
{
    int count = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const V2 tmin = {(float)(x * TILE), (float)(y * TILE)};
            const V2 tmax = {(float)((x + 1) * TILE - 1), (float)((y + 1) * TILE - 1)};
            V2 mp;
            const float f = max_contrib_power_rect(co, xy, tmin, tmax, (float)(TILE - 1), (float)(TILE - 1), mp);
            count += (f <= thr) ? 1 : 0;
        }
    return count;
}

// ------------------------------------------------------------------------------------------------
// stage 1: per-Gaussian preprocess (ref: forward.cu:68-229)
// ------------------------------------------------------------------------------------------------
void preprocess(OrcFrame& f, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float mod, const float* rotations,
                const float* cov3D_precomp, const float* view, const float* proj, const float* cam_pos,
                float tan_fovx, float tan_fovy, int32_t* radii)
{
    const int P = f.P, W = f.W, H = f.H;
    const OrcSettings& s = f.s;
    const float focal_y = (float)H / (2.0f * tan_fovy);
    const float focal_x = (float)W / (2.0f * tan_fovx);
    const V3 cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
    const int ty0 = s.tile_y1 > 0 ? s.tile_y0 : 0;
    const int ty1 = s.tile_y1 > 0 ? std::min(s.tile_y1, f.gy) : f.gy;

#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        f.tiles_touched[idx] = 0;
        const V3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        // near culling (ref: auxiliary.h:211-236)
        V3 pv;
        pv.x = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12] * 1.0f;
        pv.y = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13] * 1.0f;
        pv.z = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14] * 1.0f;
        if (pv.z <= 0.2f) continue;

        const float* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * (size_t)idx;
        else {
            compute_cov3D(scales + 3 * (size_t)idx, mod, rotations + 4 * (size_t)idx, &f.cov3D[6 * (size_t)idx]);
            cov3D = &f.cov3D[6 * (size_t)idx];
        }
        const M3 cov = compute_cov2D_full(pv, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view);
        const float opacity = opacities[idx];

        // dilation (ref: forward_common.h:108-131)
        float c2x = cov.c[0][0], c2y = cov.c[0][1], c2z = cov.c[1][1];
        c2x += 0.3f; c2z += 0.3f;
        const float det = c2x * c2z - c2y * c2y;
        float conv_scale = 1.0f;
        if (s.proper_ewa_scaling) {
            const float det_orig = cov.c[0][0] * cov.c[1][1] - cov.c[0][1] * cov.c[0][1];
            conv_scale = sqrtf(std::max(0.000025f, det_orig / det));
        }
        if (det == 0.0f) continue;
        // conic + opacity (ref: forward_common.h:133-144)
        const float det_inv = 1.f / det;
        float co[4] = {c2z * det_inv, -c2y * det_inv, c2x * det_inv, opacity * conv_scale};
        if (co[3] < ALPHA_THRESHOLD) continue;

        const float thr = log_rounded(co[3] / ALPHA_THRESHOLD);
        const float extent = s.tight_opacity_bounding ? (float)std::min(3.33, (double)sqrtf(2.0f * thr)) : 3.33f;
        const float mid = 0.5f * (c2x + c2z);
        const float lambda = mid + sqrtf(std::max(0.01f, mid * mid - det));
        const float radius = extent * sqrtf(lambda);
        if (radius <= 0.0f) continue;

        // projection (ref: auxiliary.h:83-90; the 4x4 product sums (m0*x + m1*y) + (m2*z + m3*w))
        float ph[4];
        for (int j = 0; j < 4; j++)
            ph[j] = (proj[0 + j] * mean.x + proj[4 + j] * mean.y) + (proj[8 + j] * mean.z + proj[12 + j] * 1.0f);
        const float p_w = 1.0f / (ph[3] + 0.0000001f);
        const V2 mean2D = {ndc2pix(ph[0] * p_w, W), ndc2pix(ph[1] * p_w, H)};

        const float ext_x = std::min(s.rect_bounding ? (extent * sqrtf(c2x)) : radius, radius);
        const float ext_y = std::min(s.rect_bounding ? (extent * sqrtf(c2z)) : radius, radius);
        // visibility is decided on the full frame (ref: forward.cu:177-196); tiles_touched counts the
        // tiles inside the tile-row window only (our sharding extension; window == frame by default)
        int fx0, fy0, fx1, fy1;
        get_rect(mean2D, {ext_x, ext_y}, f.gx, f.gy, 0, f.gy, fx0, fy0, fx1, fy1);
        if ((fx1 - fx0) * (fy1 - fy0) == 0) continue;
        int x0, y0, x1, y1;
        get_rect(mean2D, {ext_x, ext_y}, f.gx, f.gy, ty0, ty1, x0, y0, x1, y1);
        int tile_count = (x1 - x0) * (y1 - y0);
        if (s.tile_based_culling) {
            if (tbc_tile_count(co, mean2D, thr, fx0, fy0, fx1, fy1) == 0) continue;
            tile_count = tbc_tile_count(co, mean2D, thr, x0, y0, x1, y1);
        }

        if (!colors_precomp) color_from_sh(idx, f.D, f.M, mean, cam, shs, f.clamped.data(), f.rgb.data());

        if (f.has_inv) { // ref: forward.cu:208-220
            const M3 inv = compute_inv_cov3D(scales + 3 * (size_t)idx, rotations + 4 * (size_t)idx, mod);
            const V3 d = {cam.x - mean.x, cam.y - mean.y, cam.z - mean.z};
            float* pk = &f.cov3D_inv[12 * (size_t)idx];
            const float ux = (-inv.c[0][0]) * d.x + (-inv.c[1][0]) * d.y + (-inv.c[2][0]) * d.z;
            const float uy = (-inv.c[0][1]) * d.x + (-inv.c[1][1]) * d.y + (-inv.c[2][1]) * d.z;
            const float uz = (-inv.c[0][2]) * d.x + (-inv.c[1][2]) * d.y + (-inv.c[2][2]) * d.z;
            pk[0] = inv.c[0][0]; pk[1] = inv.c[0][1]; pk[2] = inv.c[0][2]; pk[3] = 0;
            pk[4] = inv.c[1][1]; pk[5] = inv.c[1][2]; pk[6] = inv.c[2][2]; pk[7] = 0;
            pk[8] = ux; pk[9] = uy; pk[10] = uz; pk[11] = 0;
        }

        const V3 cd = {cam.x - mean.x, cam.y - mean.y, cam.z - mean.z};
        f.depths[idx] = (s.sort_order == Z_DEPTH) ? pv.z : length3(cd);
        radii[idx] = (int)ceilf(radius);
        f.rects2D[2 * (size_t)idx] = ext_x; f.rects2D[2 * (size_t)idx + 1] = ext_y;
        f.means2D[2 * (size_t)idx] = mean2D.x; f.means2D[2 * (size_t)idx + 1] = mean2D.y;
        for (int k = 0; k < 4; k++) f.conic_opacity[4 * (size_t)idx + k] = co[k];
        f.tiles_touched[idx] = (uint32_t)tile_count;
    }
}

inline uint64_t make_key(uint32_t tile, float depth) // ref: auxiliary.h:238-244
{
    uint32_t bits;
    memcpy(&bits, &depth, 4);
    return ((uint64_t)tile << 32) | bits;
}

// ------------------------------------------------------------------------------------------------
// stage 2: duplicate with keys (ref: forward.cu:25-65 and stopthepop_common.cuh:324-621; the
// load-balanced branches emit the same multiset and are not restated)
// ------------------------------------------------------------------------------------------------
void duplicate(OrcFrame& f, const int32_t* radii, const float* inv_vp, const float* cam_pos)
{
    const OrcSettings& s = f.s;
    const bool tbc = s.tile_based_culling != 0;
    const bool per_tile_depth = s.sort_order == PTD_CENTER || s.sort_order == PTD_MAX;
    const bool eval_max = tbc || s.sort_order == PTD_MAX;
    const V3 cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
    const int ty0 = s.tile_y1 > 0 ? s.tile_y0 : 0;
    const int ty1 = s.tile_y1 > 0 ? std::min(s.tile_y1, f.gy) : f.gy;

#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < f.P; idx++) {
        if (radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0u : f.point_offsets[idx - 1];
        const uint32_t off_to = f.point_offsets[idx];
        const V2 xy = {f.means2D[2 * (size_t)idx], f.means2D[2 * (size_t)idx + 1]};
        const V2 ext = {f.rects2D[2 * (size_t)idx], f.rects2D[2 * (size_t)idx + 1]};
        int x0, y0, x1, y1;
        get_rect(xy, ext, f.gx, f.gy, ty0, ty1, x0, y0, x1, y1);
        const float* co = &f.conic_opacity[4 * (size_t)idx];
        const float thr = eval_max ? log_rounded(co[3] / ALPHA_THRESHOLD) : 0.0f;
        const float global_depth = f.depths[idx];
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const V2 tmin = {(float)(x * TILE), (float)(y * TILE)};
                const V2 tmax = {(float)((x + 1) * TILE - 1), (float)((y + 1) * TILE - 1)};
                V2 max_pos = {0, 0};
                float max_fac = 0.0f;
                if (eval_max) max_fac = max_contrib_power_rect(co, xy, tmin, tmax, (float)(TILE - 1), (float)(TILE - 1), max_pos);
                float depth = global_depth;
                if (per_tile_depth) { // ref: stopthepop_common.cuh:439-449
                    const V2 center = {(tmin.x + tmax.x) * 0.5f, (tmin.y + tmax.y) * 0.5f};
                    const V2 target = (s.sort_order == PTD_MAX) ? max_pos : center;
                    const V3 dir = view_ray(inv_vp, cam, target, f.W, f.H);
                    depth = std::max(0.0f, depth_along_ray(&f.cov3D_inv[12 * (size_t)idx], dir) + 8.0f);
                }
                const bool write = !tbc || max_fac <= thr;
                if (write) {
                    if (off < off_to) {
                        f.values_unsorted[off] = (uint32_t)idx;
                        f.keys_unsorted[off] = make_key((uint32_t)(y * f.gx + x), depth);
                    }
                    off++;
                }
            }
        for (; off < off_to; off++) { // ref: stopthepop_common.cuh:503-508
            f.values_unsorted[off] = 0xFFFFFFFFu;
            f.keys_unsorted[off] = make_key(INVALID_TILE, FLT_MAX);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage 3: stable sort on key bits [0, 32+bit) and tile ranges (ref: rasterizer_impl.cu:344-362,
// 133-158).  The device library is a stable LSD radix sort; any stable sort gives the same list.
// ------------------------------------------------------------------------------------------------
void sort_and_ranges(OrcFrame& f)
{
    const int R = f.R;
    const uint32_t bit = higher_msb((uint32_t)(f.gx * f.gy));
    const uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1ull);
    std::vector<uint32_t> order((size_t)R);
    for (int i = 0; i < R; i++) order[i] = (uint32_t)i;
    const uint64_t* ku = f.keys_unsorted.data();
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (ku[a] & mask) < (ku[b] & mask); });
    for (int i = 0; i < R; i++) {
        f.keys[i] = ku[order[i]];
        f.point_list[i] = f.values_unsorted[order[i]];
    }
    std::fill(f.ranges.begin(), f.ranges.end(), 0u);
    const size_t T = (size_t)f.gx * f.gy;
    for (int i = 0; i < R; i++) {
        const uint32_t cur = (uint32_t)(f.keys[i] >> 32);
        const bool valid = cur != INVALID_TILE;
        if (i == 0) {
            if (valid && cur < T) f.ranges[2 * (size_t)cur] = 0;
        } else {
            const uint32_t prev = (uint32_t)(f.keys[i - 1] >> 32);
            if (cur != prev) {
                if (prev < T) f.ranges[2 * (size_t)prev + 1] = (uint32_t)i;
                if (valid && cur < T) f.ranges[2 * (size_t)cur] = (uint32_t)i;
            }
        }
        if (i == R - 1 && valid && cur < T) f.ranges[2 * (size_t)cur + 1] = (uint32_t)R;
    }
}

// ------------------------------------------------------------------------------------------------
// render helpers shared by all modes
// ------------------------------------------------------------------------------------------------
struct RenderCtx {
    const OrcFrame* f;
    const float* feat;   // colours (P x 3)
    const float* bg;
    const float* inv_vp;
    V3 cam;
    bool debug_depth = false;      // DebugVisualization::Depth: the forward renders leave (sum depth*alpha*T, T) in channels 0, 1
    const float* means3D = nullptr;
};

// ref: outputDebugVis, stopthepop_common.cuh:297-301
inline void write_debug_depth(float* out, size_t N, size_t pid, float depth_acc, float T)
{
    out[pid] = depth_acc;
    out[N + pid] = T;
}

// gradient accumulators (double, so the oracle's sums are order-independent to fp32 precision)
struct GradAcc {
    std::vector<double> mean2D, conic, opacity, color; // 2P, 3P (xx,xy,yy), P, 3P
    explicit GradAcc(int P) : mean2D(2 * (size_t)P, 0.0), conic(3 * (size_t)P, 0.0), opacity((size_t)P, 0.0), color(3 * (size_t)P, 0.0) {}
    inline void add(double& dst, double v)
    {
#pragma omp atomic
        dst += v;
    }
};

struct BwdPixel {
    float T_final, dL_dpix[3], final_color[3], T, C[3];
};

// gradient of one blended (pixel, Gaussian) pair, front-to-back formulation
// (ref: hierarchical_render.cuh:1094-1166 == resorted_render.cuh:312-392).  Returns false when
// the pixel saturates (test_T < 1e-4) -- nothing is accumulated in that case.
inline bool blend_backward(const RenderCtx& c, GradAcc& g, BwdPixel& b, int px, int py, int id, float G)
{
    const OrcFrame& f = *c.f;
    const float* co = &f.conic_opacity[4 * (size_t)id];
    const float alpha = std::min(0.99f, co[3] * G);
    const float test_T = b.T * (1.0f - alpha);
    if (T_stops(test_T, f.W, px, py)) return false;
    const float dx = f.means2D[2 * (size_t)id] - (float)px;
    const float dy = f.means2D[2 * (size_t)id + 1] - (float)py;
    const float dchannel_dcolor = alpha * b.T;
    float dL_dalpha = 0.0f;
    for (int ch = 0; ch < 3; ch++) {
        const float col = c.feat[3 * (size_t)id + ch];
        b.C[ch] += col * alpha * b.T;
        const float accum_rec = (b.final_color[ch] - b.C[ch]) / test_T;
        dL_dalpha += (col - accum_rec) * b.dL_dpix[ch];
        g.add(g.color[3 * (size_t)id + ch], (double)(dchannel_dcolor * b.dL_dpix[ch]));
    }
    dL_dalpha *= b.T;
    float bg_dot = 0.0f;
    for (int ch = 0; ch < 3; ch++) bg_dot += c.bg[ch] * b.dL_dpix[ch];
    dL_dalpha += (-b.T_final / (1.f - alpha)) * bg_dot;
    const float dL_dG = co[3] * dL_dalpha;
    const float gdx = G * dx, gdy = G * dy;
    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
    const float dG_ddely = -gdy * co[2] - gdx * co[1];
    const float ddelx_dx = (float)(0.5 * f.W), ddely_dy = (float)(0.5 * f.H);
    g.add(g.mean2D[2 * (size_t)id], (double)(dL_dG * dG_ddelx * ddelx_dx));
    g.add(g.mean2D[2 * (size_t)id + 1], (double)(dL_dG * dG_ddely * ddely_dy));
    g.add(g.conic[3 * (size_t)id], (double)(-0.5f * gdx * dx * dL_dG));
    g.add(g.conic[3 * (size_t)id + 1], (double)(-0.5f * gdx * dy * dL_dG));
    g.add(g.conic[3 * (size_t)id + 2], (double)(-0.5f * gdy * dy * dL_dG));
    g.add(g.opacity[id], (double)(G * dL_dalpha));
    b.T = test_T;
    return true;
}

inline void init_bwd_pixel(const RenderCtx& c, BwdPixel& b, int px, int py, const float* pixel_colors, const float* dL_dpix)
{
    const OrcFrame& f = *c.f;
    const size_t pid = (size_t)f.W * py + px, N = (size_t)f.W * f.H;
    b.T = 1.0f;
    b.T_final = f.final_T[pid];
    for (int ch = 0; ch < 3; ch++) {
        b.C[ch] = 0.0f;
        b.dL_dpix[ch] = dL_dpix[ch * N + pid];
        b.final_color[ch] = pixel_colors[ch * N + pid] - b.T_final * c.bg[ch];
    }
}

// evaluate G = exp(power) and alpha for a pixel; returns false if the pair is skipped
inline bool eval_alpha(const OrcFrame& f, int id, int px, int py, float& G, float& alpha)
{
    const float* co = &f.conic_opacity[4 * (size_t)id];
    const float dx = f.means2D[2 * (size_t)id] - (float)px;
    const float dy = f.means2D[2 * (size_t)id + 1] - (float)py;
    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
    if (power > 0.0f) return false;
    G = expf(power);
    alpha = std::min(0.99f, co[3] * G);
    if (alpha_skips(alpha, f.W, px, py, id)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// GLOBAL mode (ref: forward.cu:234-366, backward.cu:437-595)
// ------------------------------------------------------------------------------------------------
void render_global_fwd(OrcFrame& f, const RenderCtx& c, float* out)
{
    const int W = f.W, H = f.H;
    const size_t N = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < f.gx * f.gy; tile++) {
        const int tx = tile % f.gx, ty = tile / f.gx;
        const uint32_t r0 = f.ranges[2 * (size_t)tile], r1 = f.ranges[2 * (size_t)tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                float T = 1.0f, C[3] = {0, 0, 0}, depth_acc = 0.0f;
                uint32_t contributor = 0, last = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const int id = (int)f.point_list[k];
                    float G, alpha;
                    if (!eval_alpha(f, id, px, py, G, alpha)) continue;
                    const float test_T = T * (1 - alpha);
                    if (T_stops(test_T, f.W, px, py)) break;
                    for (int ch = 0; ch < 3; ch++) C[ch] += c.feat[3 * (size_t)id + ch] * alpha * T;
                    if (c.debug_depth) { // ref: forward.cu:337-341: distance camera - mean, whatever the sort order
                        const V3 d = {c.cam.x - c.means3D[3 * (size_t)id], c.cam.y - c.means3D[3 * (size_t)id + 1], c.cam.z - c.means3D[3 * (size_t)id + 2]};
                        depth_acc += length3(d) * alpha * T;
                    }
                    T = test_T;
                    last = contributor;
                }
                const size_t pid = (size_t)W * py + px;
                f.final_T[pid] = T;
                f.n_contrib[pid] = last;
                if (c.debug_depth) write_debug_depth(out, N, pid, depth_acc, T);
                else for (int ch = 0; ch < 3; ch++) out[ch * N + pid] = C[ch] + T * c.bg[ch];
            }
    }
}

void render_global_bwd(const OrcFrame& f, const RenderCtx& c, GradAcc& g, const float* dL_dpixels)
{
    const int W = f.W, H = f.H;
    const size_t N = (size_t)W * H;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < f.gx * f.gy; tile++) {
        const int tx = tile % f.gx, ty = tile / f.gx;
        const uint32_t r0 = f.ranges[2 * (size_t)tile], r1 = f.ranges[2 * (size_t)tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pid = (size_t)W * py + px;
                const float T_final = f.final_T[pid];
                float T = T_final;
                const uint32_t last_contributor = f.n_contrib[pid];
                uint32_t contributor = r1 - r0;
                float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.0f;
                float dL_dpixel[3];
                for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels[ch * N + pid];
                for (uint32_t k = r1; k-- > r0;) { // back to front
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const int id = (int)f.point_list[k];
                    const float* co = &f.conic_opacity[4 * (size_t)id];
                    const float dx = f.means2D[2 * (size_t)id] - (float)px;
                    const float dy = f.means2D[2 * (size_t)id + 1] - (float)py;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = std::min(0.99f, co[3] * G);
                    if (alpha_skips(alpha, f.W, px, py, id)) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ch++) {
                        const float col = c.feat[3 * (size_t)id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        dL_dalpha += (col - accum_rec[ch]) * dL_dpixel[ch];
                        g.add(g.color[3 * (size_t)id + ch], (double)(dchannel_dcolor * dL_dpixel[ch]));
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot = 0.0f;
                    for (int ch = 0; ch < 3; ch++) bg_dot += c.bg[ch] * dL_dpixel[ch];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    g.add(g.mean2D[2 * (size_t)id], (double)(dL_dG * dG_ddelx * ddelx_dx));
                    g.add(g.mean2D[2 * (size_t)id + 1], (double)(dL_dG * dG_ddely * ddely_dy));
                    g.add(g.conic[3 * (size_t)id], (double)(-0.5f * gdx * dx * dL_dG));
                    g.add(g.conic[3 * (size_t)id + 1], (double)(-0.5f * gdx * dy * dL_dG));
                    g.add(g.conic[3 * (size_t)id + 2], (double)(-0.5f * gdy * dy * dL_dG));
                    g.add(g.opacity[id], (double)(G * dL_dalpha));
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// per-pixel sorted insertion window, shared by k-buffer and the hierarchical head queue
// (ref: resorted_render.cuh:74-119,186-197; hierarchical_render.cuh:386-417,509-522)
// ------------------------------------------------------------------------------------------------
struct Window {
    int cap = 0, num = 0;
    float depth[24];
    float store[24];
    int id[24];
    void init(int capacity)
    {
        cap = capacity; num = 0;
        for (int i = 0; i < cap; i++) { depth[i] = FLT_MAX; store[i] = 0.0f; id[i] = -1; }
    }
    void insert(float d, int gid, float st) // strict '<': a new entry goes after equal old ones
    {
        for (int s = 0; s < cap; s++)
            if (d < depth[s]) { std::swap(d, depth[s]); std::swap(gid, id[s]); std::swap(st, store[s]); }
        num++;
    }
    void pop()
    {
        for (int i = 1; i < cap; i++) { depth[i - 1] = depth[i]; store[i - 1] = store[i]; id[i - 1] = id[i]; }
        depth[cap - 1] = FLT_MAX;
        num--;
    }
};

inline int kbuffer_window(int w) // ref: forward.cu:409-425
{
    if (w <= 1) return 1; if (w <= 2) return 2; if (w <= 4) return 4; if (w <= 8) return 8;
    if (w <= 12) return 12; if (w <= 16) return 16; if (w <= 20) return 20; return 24;
}

// ------------------------------------------------------------------------------------------------
// PPX_KBUFFER mode (ref: resorted_render.cuh:17-221 forward, :223-471 backward)
// ------------------------------------------------------------------------------------------------
template <bool BACKWARD>
void render_kbuffer(OrcFrame& f, const RenderCtx& c, float* out, GradAcc* g, const float* pixel_colors,
                    const float* dL_dpix)
{
    const int W = f.W, H = f.H;
    const size_t N = (size_t)W * H;
    const int WIN = kbuffer_window(f.s.queue_per_pixel);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < f.gx * f.gy; tile++) {
        const int tx = tile % f.gx, ty = tile / f.gx;
        const uint32_t r0 = f.ranges[2 * (size_t)tile], r1 = f.ranges[2 * (size_t)tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const V3 dir = view_ray(c.inv_vp, c.cam, {(float)px, (float)py}, W, H);
                Window win; win.init(WIN);
                float T = 1.0f, C[3] = {0, 0, 0}, depth_acc = 0.0f;
                BwdPixel b;
                if (BACKWARD) init_bwd_pixel(c, b, px, py, pixel_colors, dL_dpix);
                bool done = false;
                uint32_t contributor = 0;
                auto blend_one = [&]() {
                    if (win.num == 0) return;
                    if (!BACKWARD) {
                        const float a = win.store[0];
                        const float test_T = T * (1 - a);
                        if (T_stops(test_T, f.W, px, py)) { win.num--; done = true; return; }
                        for (int ch = 0; ch < 3; ch++) C[ch] += c.feat[3 * (size_t)win.id[0] + ch] * a * T;
                        if (c.debug_depth) depth_acc += win.depth[0] * a * T; // ref: resorted_render.cuh:107
                        T = test_T;
                    } else {
                        if (!blend_backward(c, *g, b, px, py, win.id[0], win.store[0])) { win.num--; done = true; return; }
                    }
                    win.pop();
                };
                for (uint32_t k = r0; k < r1 && !done; k++) {
                    if (!g_lazy_pop) {
                        if (win.num == WIN) blend_one();
                        if (done) break;
                    }
                    contributor++;
                    const int id = (int)f.point_list[k];
                    if (id < 0) break; // ref: resorted_render.cuh:152-158 (never reached for valid tiles)
                    float G, alpha;
                    if (!eval_alpha(f, id, px, py, G, alpha)) continue;
                    const float depth = depth_along_ray(&f.cov3D_inv[12 * (size_t)id], dir);
                    if (depth < 0.0f) continue;
                    if (g_lazy_pop) { // (test switch: the pop only in front of a candidate that passed)
                        if (win.num == WIN) blend_one();
                        if (done) break;
                    }
                    win.insert(depth, id, BACKWARD ? G : alpha);
                }
                if (!done) while (win.num > 0 && !done) blend_one();
                if (!BACKWARD) {
                    const size_t pid = (size_t)W * py + px;
                    f.final_T[pid] = T;
                    f.n_contrib[pid] = contributor;
                    if (c.debug_depth) write_debug_depth(out, N, pid, depth_acc, T);
                    else for (int ch = 0; ch < 3; ch++) out[ch * N + pid] = C[ch] + T * c.bg[ch];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// PPX_FULL mode, forward only (ref: resorted_render.cuh:474-675).  Per pixel, a sliding sort
// window of 1024 candidates advanced by 256: see render_full_fwd below.
// ------------------------------------------------------------------------------------------------
void render_full_fwd(OrcFrame& f, const RenderCtx& c, float* out);

// ------------------------------------------------------------------------------------------------
// HIER mode (ref: hierarchical_render.cuh:207-935; semantic model in SURVEY.md Appendix A).
//
// A 16x16 tile is a 4x4 grid of 4x4-pixel sub-tiles ("tail" level); a sub-tile is a 2x2 grid of
// 2x2-pixel quads ("mid" level); every pixel owns a "head" queue.  Sub-tiles are independent.
// ------------------------------------------------------------------------------------------------
struct KeyId { float key; int id; };

// 32-wide Batcher odd-even merge sort network with strict '>' exchanges, comparator sequence as
// generated by ref: hierarchical_render.cuh:158-192 (16 workers, one comparator each per step).
void batcher_sort32(KeyId* a)
{
    const uint32_t NUM = 32;
    for (uint32_t size = 2; size <= NUM; size *= 2) {
        uint32_t stride = size / 2;
        for (uint32_t t = 0; t < NUM / 2; t++) {
            const uint32_t pos = 2 * t - (t & (stride - 1));
            if (a[pos].key > a[pos + stride].key) std::swap(a[pos], a[pos + stride]);
        }
        const uint32_t first_stride = stride;
        for (stride = first_stride / 2; stride > 0; stride /= 2) {
            for (uint32_t t = 0; t < NUM / 2; t++) {
                const uint32_t offset = t & (first_stride - 1);
                const uint32_t pos = 2 * t - (t & (stride - 1));
                if (offset >= stride)
                    if (a[pos - stride].key > a[pos].key) std::swap(a[pos - stride], a[pos]);
            }
        }
    }
}

// stable two-way merge, old entries before new ones on equal keys
// (ref: hierarchical_render.cuh:24-70 for tail and MID==8; :73-127 for the ring-buffer mid)
void merge_old_new(std::vector<KeyId>& old_list, const KeyId* nw, int n_new)
{
    std::vector<KeyId> out;
    out.reserve(old_list.size() + n_new);
    size_t i = 0; int j = 0;
    while (i < old_list.size() || j < n_new) {
        if (j >= n_new || (i < old_list.size() && old_list[i].key <= nw[j].key)) out.push_back(old_list[i++]);
        else out.push_back(nw[j++]);
    }
    old_list.swap(out);
}

template <bool BACKWARD>
struct HierPixel {
    bool inside = false, active = false;
    int px = 0, py = 0;
    V3 dir{};
    Window head;
    float T = 1.0f, C[3] = {0, 0, 0}, depth_acc = 0.0f;
    BwdPixel b{};
};

template <bool BACKWARD>
struct HierSubTile {
    const OrcFrame* f; const RenderCtx* c; GradAcc* g;
    int HEAD, MID; bool CULL;
    int cx, cy; // sub-tile corner pixel
    V3 tail_dir, mid_dir[4];
    std::vector<KeyId> tail;       // valid entries only, ascending by tail depth
    std::vector<KeyId> mid[4];     // per quad; may contain (FLT_MAX,-1) pads while draining
    HierPixel<BACKWARD> pix[4][4]; // [quad][lane]; lane = (hy*2+hx)

    void blend_front(HierPixel<BACKWARD>& p) // ref: :386-417
    {
        if (!p.active) { p.head.pop(); return; }
        const int id = p.head.id[0];
        const float st = p.head.store[0];
        bool ok;
        if (!BACKWARD) {
            const float test_T = p.T * (1.0f - st);
            if (T_stops(test_T, c->f->W, p.px, p.py)) ok = false;
            else {
                for (int ch = 0; ch < 3; ch++) p.C[ch] += c->feat[3 * (size_t)id + ch] * st * p.T;
                if (c->debug_depth) p.depth_acc += p.head.depth[0] * st * p.T; // ref: :1005-1008
                p.T = test_T;
                ok = true;
            }
        } else ok = blend_backward(*c, *g, p.b, p.px, p.py, id, st);
        if (!ok) { p.active = false; p.head.num--; return; }
        p.head.pop();
    }

    void head_feed(int m, const KeyId* F4, bool checkvalid) // ref: :421-536
    {
        bool any = false;
        for (int l = 0; l < 4; l++) any = any || pix[m][l].active;
        if (!any) return;
        for (int inner = 0; inner < 4; inner++) {
            const int id = F4[inner].id;
            for (int l = 0; l < 4; l++) {
                HierPixel<BACKWARD>& p = pix[m][l];
                if (!g_lazy_pop && p.head.num >= HEAD) blend_front(p); // before the candidate is looked at
                if (checkvalid && id == -1) continue;
                if (!p.active) continue;
                const float depth = depth_along_ray(&f->cov3D_inv[12 * (size_t)id], p.dir);
                if (depth < 0.0f) continue;
                float G, alpha;
                if (!eval_alpha(*f, id, p.px, p.py, G, alpha)) continue;
                if (g_lazy_pop) { // (test switch: the pop only in front of a candidate that passed)
                    if (p.head.num >= HEAD) blend_front(p);
                    if (!p.active) continue;
                }
                p.head.insert(depth, id, BACKWARD ? G : alpha);
            }
        }
    }

    void mid_push(int m, const KeyId* G4) // ref: :558-681
    {
        KeyId nw[4];
        for (int l = 0; l < 4; l++) {
            nw[l].id = G4[l].id;
            nw[l].key = (G4[l].id == -1) ? FLT_MAX : depth_along_ray(&f->cov3D_inv[12 * (size_t)G4[l].id], mid_dir[m]);
        }
        // rank sort, ties resolved by position (ref: :129-155)
        KeyId srt[4];
        for (int l = 0; l < 4; l++) {
            int rank = 0;
            for (int o = 0; o < 4; o++)
                if (o != l && (nw[o].key < nw[l].key || (nw[o].key == nw[l].key && o < l))) rank++;
            srt[rank] = nw[l];
        }
        merge_old_new(mid[m], srt, 4);
        if ((int)mid[m].size() > MID - 4) {
            KeyId F4[4] = {mid[m][0], mid[m][1], mid[m][2], mid[m][3]};
            mid[m].erase(mid[m].begin(), mid[m].begin() + 4);
            head_feed(m, F4, false);
        }
    }

    void emit_group(const KeyId* G4) { for (int m = 0; m < 4; m++) mid_push(m, G4); }

    void batch(const int* ids, int n) // n <= 32 consecutive list entries (ref: :690-846)
    {
        KeyId nw[32];
        for (int i = 0; i < 32; i++) {
            nw[i] = {FLT_MAX, -1};
            if (i >= n) continue;
            const int id = ids[i];
            if (id == -1) continue;
            if (CULL) { // ref: :722-743
                const V2 rmin = {(float)cx, (float)cy};
                const V2 rmax = {rmin.x + 3.0f, rmin.y + 3.0f};
                const float* co = &f->conic_opacity[4 * (size_t)id];
                const V2 xy = {f->means2D[2 * (size_t)id], f->means2D[2 * (size_t)id + 1]};
                V2 mp;
                const float power = max_contrib_power_rect(co, xy, rmin, rmax, 3.0f, 3.0f, mp);
                const float alpha = std::min(0.99f, co[3] * expf(-power));
                if (alpha < g_cull_alpha_thr) continue;
            }
            const float d = depth_along_ray(&f->cov3D_inv[12 * (size_t)id], tail_dir);
            nw[i].key = d;
            nw[i].id = (d == FLT_MAX) ? -1 : id;
        }
        batcher_sort32(nw);
        int n_valid = 0;
        while (n_valid < 32 && nw[n_valid].id != -1) n_valid++;
        merge_old_new(tail, nw, n_valid);
        for (int rep = 0; rep < 2; rep++)
            if (tail.size() > 32) {
                for (int k = 0; k < 4; k++) emit_group(&tail[4 * k]);
                tail.erase(tail.begin(), tail.begin() + 16);
            }
    }

    void drain() // ref: :855-925
    {
        for (int round = 0; round < 2 && !tail.empty(); round++)
            for (int k = 0; k < 4 && !tail.empty(); k++) {
                KeyId G4[4];
                const int take = std::min<size_t>(4, tail.size());
                for (int l = 0; l < 4; l++) G4[l] = l < take ? tail[l] : KeyId{FLT_MAX, -1};
                tail.erase(tail.begin(), tail.begin() + take);
                emit_group(G4);
            }
        for (int m = 0; m < 4; m++) {
            while (!mid[m].empty()) {
                KeyId F4[4];
                const int take = std::min<size_t>(4, mid[m].size());
                for (int l = 0; l < 4; l++) F4[l] = l < take ? mid[m][l] : KeyId{FLT_MAX, -1};
                mid[m].erase(mid[m].begin(), mid[m].begin() + take);
                head_feed(m, F4, true);
            }
            for (int l = 0; l < 4; l++) {
                HierPixel<BACKWARD>& p = pix[m][l];
                while (p.active && p.head.num > 0) blend_front(p);
            }
        }
    }
};

template <bool BACKWARD>
void render_hier(OrcFrame& f, const RenderCtx& c, float* out, GradAcc* g, const float* pixel_colors, const float* dL_dpix)
{
    const int W = f.W, H = f.H;
    const size_t N = (size_t)W * H;
    const int HEAD = f.s.queue_per_pixel, MID = f.s.queue_tile_2x2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < f.gx * f.gy; tile++) {
        const int tx = tile % f.gx, ty = tile / f.gx;
        const uint32_t r0 = f.ranges[2 * (size_t)tile], r1 = f.ranges[2 * (size_t)tile + 1];
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) {
                HierSubTile<BACKWARD> st;
                st.f = &f; st.c = &c; st.g = g; st.HEAD = HEAD; st.MID = MID; st.CULL = f.s.hierarchical_4x4_culling != 0;
                st.cx = tx * TILE + 4 * sx; st.cy = ty * TILE + 4 * sy;
                st.tail_dir = view_ray(c.inv_vp, c.cam, {(float)st.cx + 1.5f, (float)st.cy + 1.5f}, W, H);
                bool any_inside = false;
                for (int m = 0; m < 4; m++) {
                    const int mx = m % 2, my = m / 2;
                    st.mid_dir[m] = view_ray(c.inv_vp, c.cam, {(float)st.cx + 0.5f + 2 * mx, (float)st.cy + 0.5f + 2 * my}, W, H);
                    for (int l = 0; l < 4; l++) {
                        HierPixel<BACKWARD>& p = st.pix[m][l];
                        p.px = st.cx + mx * 2 + (l % 2); p.py = st.cy + my * 2 + (l / 2);
                        p.inside = p.px < W && p.py < H;
                        p.active = p.inside;
                        any_inside = any_inside || p.inside;
                        p.head.init(HEAD);
                        p.T = 1.0f; p.C[0] = p.C[1] = p.C[2] = 0.0f;
                        if (p.inside) {
                            p.dir = view_ray(c.inv_vp, c.cam, {(float)p.px, (float)p.py}, W, H);
                            if (BACKWARD) init_bwd_pixel(c, p.b, p.px, p.py, pixel_colors, dL_dpix);
                        }
                    }
                }
                if (any_inside) {
                    int ids[32];
                    for (uint32_t prog = r0; prog < r1; prog += 32) {
                        const int n = (int)std::min<uint32_t>(32, r1 - prog);
                        for (int i = 0; i < n; i++) ids[i] = (int)f.point_list[prog + i];
                        st.batch(ids, n);
                    }
                    st.drain();
                }
                if (!BACKWARD)
                    for (int m = 0; m < 4; m++)
                        for (int l = 0; l < 4; l++) {
                            HierPixel<BACKWARD>& p = st.pix[m][l];
                            if (!p.inside) continue;
                            const size_t pid = (size_t)W * p.py + p.px;
                            f.final_T[pid] = p.T;
                            if (c.debug_depth) write_debug_depth(out, N, pid, p.depth_acc, p.T);
                            else for (int ch = 0; ch < 3; ch++) out[ch * N + pid] = p.C[ch] + p.T * c.bg[ch];
                        }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// PPX_FULL forward.  ref: resorted_render.cuh:474-675.  For every pixel the block keeps up to 1024
// candidates (depth along the pixel's ray, id), sorted; whenever more list entries remain, the
// front 256 are blended and 256 new candidates join.  Candidates are every list entry (no alpha
// pre-test, and -- unlike the other sorted modes -- no rejection of negative depths).
// ------------------------------------------------------------------------------------------------
void render_full_fwd(OrcFrame& f, const RenderCtx& c, float* out)
{
    const int W = f.W, H = f.H;
    const size_t N = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < f.gx * f.gy; tile++) {
        const int tx = tile % f.gx, ty = tile / f.gx;
        const uint32_t r0 = f.ranges[2 * (size_t)tile], r1 = f.ranges[2 * (size_t)tile + 1];
        std::vector<KeyId> win;
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const V3 dir = view_ray(c.inv_vp, c.cam, {(float)px, (float)py}, W, H);
                float T = 1.0f, C[3] = {0, 0, 0}, depth_acc = 0.0f;
                bool done = false;
                uint32_t contributor = 0, last_contributor = 0;
                win.clear();
                uint32_t next = r0;
                auto refill = [&](uint32_t count) {
                    std::vector<KeyId> nw;
                    for (uint32_t i = 0; i < count && next < r1; i++, next++) {
                        const int id = (int)f.point_list[next];
                        nw.push_back({depth_along_ray(&f.cov3D_inv[12 * (size_t)id], dir), id});
                    }
                    std::stable_sort(nw.begin(), nw.end(), [](const KeyId& a, const KeyId& b) { return a.key < b.key; });
                    merge_old_new(win, nw.data(), (int)nw.size());
                };
                auto blend = [&](size_t count) {
                    for (size_t i = 0; i < count && i < win.size() && !done; i++) {
                        contributor++;
                        const int id = win[i].id;
                        const float* co = &f.conic_opacity[4 * (size_t)id];
                        const float dx = f.means2D[2 * (size_t)id] - (float)px, dy = f.means2D[2 * (size_t)id + 1] - (float)py;
                        const float power = opacity_factor(dx, dy, co); // positive form (ref: resorted_render.cuh:621-630)
                        if (power < 0.0f) continue;
                        const float alpha = std::min(0.99f, co[3] * expf(-power));
                        if (alpha_skips(alpha, f.W, px, py, id)) continue;
                        const float test_T = T * (1 - alpha);
                        if (T_stops(test_T, f.W, px, py)) { done = true; break; }
                        for (int ch = 0; ch < 3; ch++) C[ch] += c.feat[3 * (size_t)id + ch] * alpha * T;
                        if (c.debug_depth) depth_acc += win[i].key * alpha * T; // ref: resorted_render.cuh:647
                        T = test_T;
                        last_contributor = contributor;
                    }
                    win.erase(win.begin(), win.begin() + std::min(count, win.size()));
                };
                refill(1024);
                while (!done && next < r1) { blend(256); refill(256); }
                if (!done) blend(win.size());
                const size_t pid = (size_t)W * py + px;
                f.final_T[pid] = T;
                f.n_contrib[pid] = last_contributor;
                if (c.debug_depth) write_debug_depth(out, N, pid, depth_acc, T);
                else for (int ch = 0; ch < 3; ch++) out[ch * N + pid] = C[ch] + T * c.bg[ch];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// DebugVisualization::Depth, second half.  ref: applyDebugVisualization (rasterizer_impl.cu:54-109: min and max of
// channel 0 over the frame), render_debug_CUDA<DEPTH = true> (forward.cu:674-713) and colormapTurbo
// (stopthepop_common.cuh:641-657).  Pixels of tiles outside a tile-row window keep their zeros, as on the device.
// ------------------------------------------------------------------------------------------------
static const float kTurbo[STP_TURBO_ENTRIES * 3] = STP_TURBO_TABLE_INITIALIZER;

void apply_depth_colormap(float* out, size_t N)
{
    float mn = out[0], mx = out[0];
    for (size_t i = 1; i < N; i++) { mn = std::min(mn, out[i]); mx = std::max(mx, out[i]); }
    for (size_t i = 0; i < N; i++) {
        const float T = out[N + i];
        const float x = std::min(std::max(out[i] + T * mx, mn), mx) / (mx - mn);
        const float interp = std::min(std::max(x * 255.0f, 0.0f), 255.0f);
        const int lo = x > 0.0f ? (int)interp : 0;
        const int hi = lo >= 255 ? 255 : lo + 1;
        const float diff = interp - (float)lo;
        for (int ch = 0; ch < 3; ch++) {
            const float a = kTurbo[3 * lo + ch], b = kTurbo[3 * hi + ch];
            out[ch * N + i] = std::min(std::max(a + (b - a) * diff, 0.0f), 1.0f);
        }
    }
}

int g_ewa_exact_grad = 0; // test-only, see backward_preprocess

// ------------------------------------------------------------------------------------------------
// backward of the per-Gaussian stages (ref: backward.cu:146-312 computeCov2DCUDA,
// :384-434 preprocessCUDA, :22-141 SH, :316-379 cov3D)
// ------------------------------------------------------------------------------------------------
void backward_preprocess(const OrcFrame& f, const int32_t* radii, const float* means3D, const float* shs,
                         const float* opacities, const float* scales, float mod, const float* rotations,
                         const float* cov3D_ptr, const float* view, const float* proj, const float* cam_pos,
                         float tan_fovx, float tan_fovy, const float* dL_dmean2D, const float* dL_dconic,
                         float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot)
{
    const int P = f.P, D = f.D, M = f.M;
    const float h_y = (float)f.H / (2.0f * tan_fovy);
    const float h_x = (float)f.W / (2.0f * tan_fovx);
    const V3 cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const V3 mean = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
        // ---- computeCov2DCUDA ----
        {
            const float* cov3D = cov3D_ptr + 6 * (size_t)idx;
            const float dcx = dL_dconic[4 * (size_t)idx], dcy = dL_dconic[4 * (size_t)idx + 1], dcz = dL_dconic[4 * (size_t)idx + 3];
            V3 t;
            t.x = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12];
            t.y = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13];
            t.z = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
            const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
            const float txtz = t.x / t.z, tytz = t.y / t.z;
            t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
            t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
            const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
            const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
            M3 J;
            J.c[0][0] = h_x / t.z; J.c[0][1] = 0.0f;      J.c[0][2] = -(h_x * t.x) / (t.z * t.z);
            J.c[1][0] = 0.0f;      J.c[1][1] = h_y / t.z; J.c[1][2] = -(h_y * t.y) / (t.z * t.z);
            J.c[2][0] = 0.0f;      J.c[2][1] = 0.0f;      J.c[2][2] = 0.0f;
            M3 Wm;
            Wm.c[0][0] = view[0]; Wm.c[0][1] = view[4]; Wm.c[0][2] = view[8];
            Wm.c[1][0] = view[1]; Wm.c[1][1] = view[5]; Wm.c[1][2] = view[9];
            Wm.c[2][0] = view[2]; Wm.c[2][1] = view[6]; Wm.c[2][2] = view[10];
            M3 Vrk;
            Vrk.c[0][0] = cov3D[0]; Vrk.c[0][1] = cov3D[1]; Vrk.c[0][2] = cov3D[2];
            Vrk.c[1][0] = cov3D[1]; Vrk.c[1][1] = cov3D[3]; Vrk.c[1][2] = cov3D[4];
            Vrk.c[2][0] = cov3D[2]; Vrk.c[2][1] = cov3D[4]; Vrk.c[2][2] = cov3D[5];
            const M3 T = mul(Wm, J);
            const M3 cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
            float c_xx = cov2D.c[0][0], c_xy = cov2D.c[0][1], c_yy = cov2D.c[1][1];
            const float det_cov_orig = c_xx * c_yy - c_xy * c_xy;
            const float h_var = 0.3f;
            c_xx += h_var; c_yy += h_var;
            float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
            if (f.s.proper_ewa_scaling) { // ref: backward.cu:214-238
                const float det_plus = c_xx * c_yy - c_xy * c_xy;
                const float h_scal = sqrtf(std::max(0.000025f, det_cov_orig / det_plus));
                const float dL_dop_v = dL_dopacity[idx];
                const float d_h_scal = dL_dop_v * opacities[idx];
                dL_dopacity[idx] = dL_dop_v * h_scal;
                const float d_inside_root = (det_cov_orig / det_plus) <= 0.000025f ? 0.f : d_h_scal / (2 * h_scal);
                // NB the reference evaluates this closed form with the DILATED c_xx/c_yy (it reuses the
                // variables after "+= h_var"), although the derivative is in terms of the undilated
                // entries.  We restate the reference as is; g_ewa_exact_grad (test-only switch) selects
                // the mathematically exact variant so the rest of this path can be checked against autograd.
                const float x = g_ewa_exact_grad ? c_xx - h_var : c_xx, y = g_ewa_exact_grad ? c_yy - h_var : c_yy;
                const float z = c_xy, w = h_var;
                const float q = w * w + w * (x + y) + x * y - z * z;
                const float denom_f = d_inside_root / (q * q);
                dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
                dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
                dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
            }
            const float denom = c_xx * c_yy - c_xy * c_xy;
            const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            float* dcov = dL_dcov3D + 6 * (size_t)idx;
            if (denom2inv != 0) {
                dL_dc_xx += denom2inv * (-c_yy * c_yy * dcx + 2 * c_xy * c_yy * dcy + (denom - c_xx * c_yy) * dcz);
                dL_dc_yy += denom2inv * (-c_xx * c_xx * dcz + 2 * c_xx * c_xy * dcy + (denom - c_xx * c_yy) * dcx);
                dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dcx - (denom + 2 * c_xy * c_xy) * dcy + c_xx * c_xy * dcz);
                dcov[0] = (T.c[0][0] * T.c[0][0] * dL_dc_xx + T.c[0][0] * T.c[1][0] * dL_dc_xy + T.c[1][0] * T.c[1][0] * dL_dc_yy);
                dcov[3] = (T.c[0][1] * T.c[0][1] * dL_dc_xx + T.c[0][1] * T.c[1][1] * dL_dc_xy + T.c[1][1] * T.c[1][1] * dL_dc_yy);
                dcov[5] = (T.c[0][2] * T.c[0][2] * dL_dc_xx + T.c[0][2] * T.c[1][2] * dL_dc_xy + T.c[1][2] * T.c[1][2] * dL_dc_yy);
                dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_dc_xx + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][1] * dL_dc_yy;
                dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_dc_xx + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_dc_xy + 2 * T.c[1][0] * T.c[1][2] * dL_dc_yy;
                dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_dc_xx + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_dc_xy + 2 * T.c[1][1] * T.c[1][2] * dL_dc_yy;
            } else {
                for (int i = 0; i < 6; i++) dcov[i] = 0;
            }
            const float dL_dT00 = 2 * (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_dc_xx +
                                  (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc_xy;
            const float dL_dT01 = 2 * (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_dc_xx +
                                  (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc_xy;
            const float dL_dT02 = 2 * (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_dc_xx +
                                  (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc_xy;
            const float dL_dT10 = 2 * (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc_yy +
                                  (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_dc_xy;
            const float dL_dT11 = 2 * (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc_yy +
                                  (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_dc_xy;
            const float dL_dT12 = 2 * (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc_yy +
                                  (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_dc_xy;
            const float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
            const float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
            const float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
            const float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
            const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
            const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
            const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
            // transformVec4x3Transpose (ref: auxiliary.h:161-169)
            dL_dmean3D[3 * (size_t)idx + 0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
            dL_dmean3D[3 * (size_t)idx + 1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
            dL_dmean3D[3 * (size_t)idx + 2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
        }
        // ---- preprocessCUDA (backward) ----
        {
            const V3 m = mean;
            const float mhw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
            const float m_w = 1.0f / (mhw + 0.0000001f);
            const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
            const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
            const float gx = dL_dmean2D[3 * (size_t)idx], gy = dL_dmean2D[3 * (size_t)idx + 1];
            dL_dmean3D[3 * (size_t)idx + 0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
            dL_dmean3D[3 * (size_t)idx + 1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
            dL_dmean3D[3 * (size_t)idx + 2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        }
        if (shs) { // ref: backward.cu:22-141
            const V3 dir_orig = {mean.x - cam.x, mean.y - cam.y, mean.z - cam.z};
            const float len = length3(dir_orig);
            const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
            const float* sh = shs + (size_t)idx * M * 3;
            float* dsh = dL_dsh + (size_t)idx * M * 3;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * (size_t)idx + ch] * (f.clamped[3 * (size_t)idx + ch] ? 0.0f : 1.0f);
            float ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                auto c = [&](int k) { return sh[3 * k + ch]; };
                auto w = [&](int k, float v) { dsh[3 * k + ch] = v * dRGB[ch]; };
                float dx = 0, dy = 0, dz = 0;
                w(0, SH_C0);
                if (D > 0) {
                    w(1, -SH_C1 * y); w(2, SH_C1 * z); w(3, -SH_C1 * x);
                    dx = -SH_C1 * c(3); dy = -SH_C1 * c(1); dz = SH_C1 * c(2);
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        w(4, SH_C2[0] * xy); w(5, SH_C2[1] * yz); w(6, SH_C2[2] * (2.f * zz - xx - yy));
                        w(7, SH_C2[3] * xz); w(8, SH_C2[4] * (xx - yy));
                        dx += SH_C2[0] * y * c(4) + SH_C2[2] * 2.f * -x * c(6) + SH_C2[3] * z * c(7) + SH_C2[4] * 2.f * x * c(8);
                        dy += SH_C2[0] * x * c(4) + SH_C2[1] * z * c(5) + SH_C2[2] * 2.f * -y * c(6) + SH_C2[4] * 2.f * -y * c(8);
                        dz += SH_C2[1] * y * c(5) + SH_C2[2] * 2.f * 2.f * z * c(6) + SH_C2[3] * x * c(7);
                        if (D > 2) {
                            w(9, SH_C3[0] * y * (3.f * xx - yy)); w(10, SH_C3[1] * xy * z);
                            w(11, SH_C3[2] * y * (4.f * zz - xx - yy)); w(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                            w(13, SH_C3[4] * x * (4.f * zz - xx - yy)); w(14, SH_C3[5] * z * (xx - yy));
                            w(15, SH_C3[6] * x * (xx - 3.f * yy));
                            dx += (SH_C3[0] * c(9) * 3.f * 2.f * xy + SH_C3[1] * c(10) * yz + SH_C3[2] * c(11) * -2.f * xy +
                                   SH_C3[3] * c(12) * -3.f * 2.f * xz + SH_C3[4] * c(13) * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * c(14) * 2.f * xz + SH_C3[6] * c(15) * 3.f * (xx - yy));
                            dy += (SH_C3[0] * c(9) * 3.f * (xx - yy) + SH_C3[1] * c(10) * xz + SH_C3[2] * c(11) * (-3.f * yy + 4.f * zz - xx) +
                                   SH_C3[3] * c(12) * -3.f * 2.f * yz + SH_C3[4] * c(13) * -2.f * xy + SH_C3[5] * c(14) * -2.f * yz +
                                   SH_C3[6] * c(15) * -3.f * 2.f * xy);
                            dz += (SH_C3[1] * c(10) * xy + SH_C3[2] * c(11) * 4.f * 2.f * yz + SH_C3[3] * c(12) * 3.f * (2.f * zz - xx - yy) +
                                   SH_C3[4] * c(13) * 4.f * 2.f * xz + SH_C3[5] * c(14) * (xx - yy));
                        }
                    }
                }
                ddir[0] += dx * dRGB[ch]; ddir[1] += dy * dRGB[ch]; ddir[2] += dz * dRGB[ch];
            }
            // dnormvdv (ref: auxiliary.h:179-189)
            const V3 v = dir_orig;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dL_dmean3D[3 * (size_t)idx + 0] += ((+sum2 - v.x * v.x) * ddir[0] - v.y * v.x * ddir[1] - v.z * v.x * ddir[2]) * invsum32;
            dL_dmean3D[3 * (size_t)idx + 1] += (-v.x * v.y * ddir[0] + (sum2 - v.y * v.y) * ddir[1] - v.z * v.y * ddir[2]) * invsum32;
            dL_dmean3D[3 * (size_t)idx + 2] += (-v.x * v.z * ddir[0] - v.y * v.z * ddir[1] + (sum2 - v.z * v.z) * ddir[2]) * invsum32;
        }
        if (scales) { // ref: backward.cu:316-379
            const float* q = rotations + 4 * (size_t)idx;
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            const M3 R = quat_matrix(q);
            const float sx = mod * scales[3 * (size_t)idx], sy = mod * scales[3 * (size_t)idx + 1], sz = mod * scales[3 * (size_t)idx + 2];
            const M3 S = diag(sx, sy, sz);
            const M3 Mm = mul(S, R);
            const float* d = dL_dcov3D + 6 * (size_t)idx;
            M3 dSig;
            dSig.c[0][0] = d[0];        dSig.c[0][1] = 0.5f * d[1]; dSig.c[0][2] = 0.5f * d[2];
            dSig.c[1][0] = 0.5f * d[1]; dSig.c[1][1] = d[3];        dSig.c[1][2] = 0.5f * d[4];
            dSig.c[2][0] = 0.5f * d[2]; dSig.c[2][1] = 0.5f * d[4]; dSig.c[2][2] = d[5];
            M3 M2;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M2.c[i][j] = 2.0f * Mm.c[i][j];
            const M3 dL_dM = mul(M2, dSig);
            const M3 Rt = transpose(R);
            M3 dMt = transpose(dL_dM);
            dL_dscale[3 * (size_t)idx + 0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
            dL_dscale[3 * (size_t)idx + 1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
            dL_dscale[3 * (size_t)idx + 2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
            for (int j = 0; j < 3; j++) { dMt.c[0][j] *= sx; dMt.c[1][j] *= sy; dMt.c[2][j] *= sz; }
            float* dq = dL_drot + 4 * (size_t)idx;
            dq[0] = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
            dq[1] = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
            dq[2] = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
            dq[3] = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
        }
    }
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int orc_forward(int P, int D, int M, const float* background, int W, int H, const OrcSettings* settings,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                int* radii, OrcFrame** frame_out)
{
    (void)prefiltered;
    OrcFrame* fp = new OrcFrame();
    OrcFrame& f = *fp;
    f.P = P; f.D = D; f.M = M; f.W = W; f.H = H; f.s = *settings;
    f.gx = (W + TILE - 1) / TILE; f.gy = (H + TILE - 1) / TILE;
    f.has_inv = requires_depth_along_ray(f.s);
    const size_t N = (size_t)W * H, T = (size_t)f.gx * f.gy;
    for (size_t i = 0; i < 3 * N; i++) out_color[i] = 0.0f;
    f.final_T.assign(N, 0.0f); f.n_contrib.assign(N, 0u); f.ranges.assign(2 * T, 0u);
    if (frame_out) *frame_out = fp;
    if (P == 0) { if (!frame_out) delete fp; return 0; }
    if (f.has_inv && (!scales || !rotations)) {
        if (frame_out) *frame_out = nullptr;
        delete fp;
        return -2;
    }
    if (f.s.sort_mode == HIER) {
        const int h = f.s.queue_per_pixel, m = f.s.queue_tile_2x2;
        if (!(h == 4 || h == 8 || h == 16) || !(m == 8 || m == 12 || m == 20)) { // ref: forward.cu:465-481
            if (frame_out) *frame_out = nullptr;
            delete fp;
            return -3;
        }
    }
    f.depths.assign(P, 0.0f); f.clamped.assign(3 * (size_t)P, 0); f.radii.assign(P, 0);
    f.rects2D.assign(2 * (size_t)P, 0.0f); f.means2D.assign(2 * (size_t)P, 0.0f); f.cov3D.assign(6 * (size_t)P, 0.0f);
    if (f.has_inv) f.cov3D_inv.assign(12 * (size_t)P, 0.0f);
    f.conic_opacity.assign(4 * (size_t)P, 0.0f); f.rgb.assign(3 * (size_t)P, 0.0f);
    f.tiles_touched.assign(P, 0u); f.point_offsets.assign(P, 0u);

    int32_t* rad = radii ? radii : f.radii.data();
    preprocess(f, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
               viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, rad);
    if (radii) memcpy(f.radii.data(), radii, sizeof(int32_t) * (size_t)P);

    uint32_t acc = 0; // inclusive scan (ref: rasterizer_impl.cu:313)
    for (int i = 0; i < P; i++) { acc += f.tiles_touched[i]; f.point_offsets[i] = acc; }
    f.R = (int)acc;
    f.keys_unsorted.assign(f.R, 0ull); f.keys.assign(f.R, 0ull);
    f.values_unsorted.assign(f.R, 0u); f.point_list.assign(f.R, 0u);
    duplicate(f, rad, inv_viewprojmatrix, cam_pos);
    sort_and_ranges(f);

    RenderCtx c;
    c.f = &f; c.feat = colors_precomp ? colors_precomp : f.rgb.data(); c.bg = background; c.inv_vp = inv_viewprojmatrix;
    c.cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
    c.debug_depth = f.s.debug_visualization == 1; c.means3D = means3D;
    switch (f.s.sort_mode) {
    case GLOBAL: render_global_fwd(f, c, out_color); break;
    case PPX_KBUFFER: render_kbuffer<false>(f, c, out_color, nullptr, nullptr, nullptr); break;
    case PPX_FULL: render_full_fwd(f, c, out_color); break;
    case HIER: render_hier<false>(f, c, out_color, nullptr, nullptr, nullptr); break;
    default: if (frame_out) *frame_out = nullptr; delete fp; return -4;
    }
    if (c.debug_depth) apply_depth_colormap(out_color, (size_t)W * H);
    const int R = f.R;
    if (!frame_out) delete fp;
    return R;
}

int orc_backward(const OrcFrame* frame, const float* background, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* inv_viewprojmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                 const float* pixel_colors, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot)
{
    if (!frame) return -1;
    OrcFrame& f = *const_cast<OrcFrame*>(frame);
    const int P = f.P;
    if (P == 0) return 0;
    if (f.s.sort_mode == PPX_FULL) return -5; // ref: backward.cu:733-736
    if (f.s.sort_mode == HIER) {
        const int h = f.s.queue_per_pixel;
        if (!(h == 4 || h == 8 || h == 12 || h == 16)) return -3; // ref: backward.cu:745-752
    }
    RenderCtx c;
    c.f = &f; c.feat = colors_precomp ? colors_precomp : f.rgb.data(); c.bg = background; c.inv_vp = inv_viewprojmatrix;
    c.cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
    GradAcc g(P);
    switch (f.s.sort_mode) {
    case GLOBAL: render_global_bwd(f, c, g, dL_dpix); break;
    case PPX_KBUFFER: render_kbuffer<true>(f, c, nullptr, &g, pixel_colors, dL_dpix); break;
    case HIER: render_hier<true>(f, c, nullptr, &g, pixel_colors, dL_dpix); break;
    default: return -4;
    }
    for (int i = 0; i < P; i++) {
        dL_dmean2D[3 * (size_t)i] += (float)g.mean2D[2 * (size_t)i];
        dL_dmean2D[3 * (size_t)i + 1] += (float)g.mean2D[2 * (size_t)i + 1];
        dL_dconic[4 * (size_t)i] += (float)g.conic[3 * (size_t)i];
        dL_dconic[4 * (size_t)i + 1] += (float)g.conic[3 * (size_t)i + 1];
        dL_dconic[4 * (size_t)i + 3] += (float)g.conic[3 * (size_t)i + 2];
        dL_dopacity[i] += (float)g.opacity[i];
        for (int ch = 0; ch < 3; ch++) dL_dcolor[3 * (size_t)i + ch] += (float)g.color[3 * (size_t)i + ch];
    }
    const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : f.cov3D.data();
    backward_preprocess(f, f.radii.data(), means3D, shs, opacities, scales, scale_modifier, rotations, cov3D_ptr,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, dL_dmean2D, dL_dconic, dL_dopacity,
                        dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    return 0;
}

void orc_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    (void)proj; // ref: rasterizer_impl.cu:113-128 -- only the view-space near test is applied
    for (int i = 0; i < P; i++) {
        const float z = view[2] * means3D[3 * (size_t)i] + view[6] * means3D[3 * (size_t)i + 1] + view[10] * means3D[3 * (size_t)i + 2] + view[14] * 1.0f;
        present[i] = z > 0.2f ? 1 : 0;
    }
}

void orc_frame_free(OrcFrame* frame) { delete frame; }
int orc_frame_num_rendered(const OrcFrame* frame) { return frame ? frame->R : -1; }

// Diagnostic for the parity tests: the 4x4-culling alpha (ref: hierarchical_render.cuh:722-743) of every entry of tile
// `tile` with respect to the sub-tile whose corner pixel is (cx, cy), as opacity * exp(-power) evaluated in DOUBLE on the
// fp32 `power` the cull test computes -- i.e. how far each entry's decision sits from the 1/255 threshold, independent
// of which expf implementation rounds it.  Returns the number of entries written (<= cap).
int orc_cull_alpha(const OrcFrame* f, int tile, int cx, int cy, double* out, int cap)
{
    if (!f || !out || tile < 0 || (size_t)(2 * tile + 1) >= f->ranges.size()) return -1;
    const uint32_t lo = f->ranges[2 * (size_t)tile], hi = f->ranges[2 * (size_t)tile + 1];
    int n = 0;
    for (uint32_t i = lo; i < hi && n < cap; i++, n++) {
        const uint32_t id = f->point_list[i];
        const float* co = &f->conic_opacity[4 * (size_t)id];
        const V2 xy = {f->means2D[2 * (size_t)id], f->means2D[2 * (size_t)id + 1]};
        const V2 rmin = {(float)cx, (float)cy};
        const V2 rmax = {rmin.x + 3.0f, rmin.y + 3.0f};
        V2 mp;
        const float power = max_contrib_power_rect(co, xy, rmin, rmax, 3.0f, 3.0f, mp);
        out[n] = (double)co[3] * std::exp(-(double)power);
    }
    return n;
}

int64_t orc_frame_array(const OrcFrame* f, const char* name, const void** data)
{
    if (!f || !name || !data) return -1;
    const std::string n(name);
#define ARR(nm, vec) if (n == nm) { *data = (vec).data(); return (int64_t)(vec).size(); }
    ARR("depths", f->depths) ARR("clamped", f->clamped) ARR("radii", f->radii) ARR("rects2D", f->rects2D)
    ARR("means2D", f->means2D) ARR("cov3D", f->cov3D) ARR("cov3D_inv", f->cov3D_inv) ARR("conic_opacity", f->conic_opacity)
    ARR("rgb", f->rgb) ARR("tiles_touched", f->tiles_touched) ARR("point_offsets", f->point_offsets)
    ARR("keys_unsorted", f->keys_unsorted) ARR("values_unsorted", f->values_unsorted) ARR("keys", f->keys)
    ARR("point_list", f->point_list) ARR("ranges", f->ranges) ARR("final_T", f->final_T) ARR("n_contrib", f->n_contrib)
#undef ARR
    return -1;
}

void orc_set_flag(const char* name, int value)
{
    if (name && std::string(name) == "ewa_exact_grad") g_ewa_exact_grad = value;
    if (name && std::string(name) == "ieee_depth") g_ieee_depth = value;
    if (name && std::string(name) == "lazy_pop") g_lazy_pop = value;
}

void orc_set_blend_nudge(float alpha_delta, float T_delta, float cull_alpha_delta) // (0, 0, 0) = the reference's thresholds
{
    g_blend_alpha_thr = ALPHA_THRESHOLD + alpha_delta;
    g_blend_T_thr = T_THRESHOLD + T_delta;
    g_cull_alpha_thr = ALPHA_THRESHOLD + cull_alpha_delta;
}

void orc_set_forced_alpha_flips(int n, const uint64_t* keys) // n = 0: none
{
    g_forced_alpha.assign(keys, keys + (n > 0 ? n : 0));
    std::sort(g_forced_alpha.begin(), g_forced_alpha.end());
}

void orc_set_forced_T_flips(int n, const uint32_t* pixels) // n = 0: none
{
    g_forced_T.assign(pixels, pixels + (n > 0 ? n : 0));
    std::sort(g_forced_T.begin(), g_forced_T.end());
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

} // extern "C"
