// stp_preprocess.hip -- per-Gaussian forward stage, tile-key emission, tile ranges, visibility.
//
// Replaces (reference file:line under cuda_rasterizer/):
//   preprocessCUDA<3,TBC,LB>              forward.cu:68-229      -> preprocess_kernel
//   duplicateWithKeysCUDA                 forward.cu:25-65       \  duplicate_kernel (one kernel; the
//   duplicateWithKeys_extended<...>       stopthepop_common.cuh:324-621 /  options are runtime-uniform)
//   identifyTileRanges                    rasterizer_impl.cu:133-158 -> tile_ranges_kernel
//   checkFrustum                          rasterizer_impl.cu:113-128 -> mark_visible_kernel
//
// All four are HBM-streaming kernels: one thread per Gaussian (or per sorted duplicate), inputs
// read once with the widest loads the layout allows, SoA outputs written once.  Load balancing
// (the reference's warp-cooperative handling of Gaussians with > 32 tiles) only changes speed,
// never results; here every Gaussian's tile loop is per-thread.
#include "stp_internal.h"
#include "stp_device.h"

namespace stp {

namespace {

// Everything the preprocess kernel needs, by value (kernarg segment; no constant-memory upload).
struct PreArgs {
    int P, D, M, W, H, gx, gy, ty0, ty1;
    float focal_x, focal_y, tan_fovx, tan_fovy, scale_modifier;
    int sort_order, rect_bounding, tight_opacity_bounding, tile_based_culling, proper_ewa_scaling, prefiltered;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* view;
    const float* proj;
    const float* cam;
    int* radii;
    GeometryState g;
    uint32_t* tile_counts; // binning by tile counters: entries per tile (nullptr: not counted)
};

// SH -> RGB, reference forward_common.h:20-70 (same association of the sums)
// `sh`: the Gaussian's own 3M coefficients (a row of the workgroup's LDS staging area in sh_color_kernel)
__device__ __forceinline__ void sh_to_rgb(int idx, int deg, float3 mean, float3 cam, const float* __restrict__ sh,
                                          uint8_t* __restrict__ clamped, float* __restrict__ rgb)
{
#pragma clang fp contract(off)
    float dx = mean.x - cam.x, dy = mean.y - cam.y, dz = mean.z - cam.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    float res[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) res[ch] = kSH_C0 * sh[ch];
    if (deg > 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            res[ch] = res[ch] - (kSH_C1 * y) * sh[3 + ch] + (kSH_C1 * z) * sh[6 + ch] - (kSH_C1 * x) * sh[9 + ch];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                res[ch] = res[ch] + (kSH_C2[0] * xy) * sh[12 + ch] + (kSH_C2[1] * yz) * sh[15 + ch] +
                          (kSH_C2[2] * (2.0f * zz - xx - yy)) * sh[18 + ch] + (kSH_C2[3] * xz) * sh[21 + ch] +
                          (kSH_C2[4] * (xx - yy)) * sh[24 + ch];
            if (deg > 2) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    res[ch] = res[ch] + (kSH_C3[0] * y * (3.0f * xx - yy)) * sh[27 + ch] + (kSH_C3[1] * xy * z) * sh[30 + ch] +
                              (kSH_C3[2] * y * (4.0f * zz - xx - yy)) * sh[33 + ch] +
                              (kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[36 + ch] +
                              (kSH_C3[4] * x * (4.0f * zz - xx - yy)) * sh[39 + ch] + (kSH_C3[5] * z * (xx - yy)) * sh[42 + ch] +
                              (kSH_C3[6] * x * (xx - 3.0f * yy)) * sh[45 + ch];
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float r = res[ch] + 0.5f;
        clamped[3 * (size_t)idx + ch] = (r < 0.0f) ? 1 : 0;
        rgb[3 * (size_t)idx + ch] = fmaxf(r, 0.0f);
    }
}

// Rectangles with more tiles than this are walked by the whole 64-lane wave instead of their owner thread (below).
constexpr int COOP_TILES = 64;

__device__ __forceinline__ int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of a 64-bit ballot below my lane (no shift by the lane index: (1ull << lane) - 1 is the trap of SURVEY 0)
__device__ __forceinline__ int lanes_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)); }
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// What the first half of the per-Gaussian forward hands to the second (everything up to the tile rectangle).
struct PreStageA {
    float3 mean, pv, sc;
    float4 q;
    float c3[6];
    float4 co;
    float thr, radius;
    float2 mean2D, rect_dims;
    int fx0, fy0, fx1, fy1, y0, y1;
};

// reference preprocessCUDA forward.cu:68-186 up to the rectangle; false where the reference returns (culled)
__device__ __forceinline__ bool preprocess_stage_a(const PreArgs& a, int idx, PreStageA& o)
{
#pragma clang fp contract(off)
    // every per-Gaussian input up front, before the first test can branch: the loads are in flight together (one memory
    // latency) instead of one after each early-out
    const float3 mean = make_float3(a.means3D[3 * (size_t)idx], a.means3D[3 * (size_t)idx + 1], a.means3D[3 * (size_t)idx + 2]);
    const float opacity = a.opacities[idx];
    float3 sc_in = make_float3(0.0f, 0.0f, 0.0f);
    float4 q_in = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float c3_in[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (a.cov3D_precomp != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; k++) c3_in[k] = a.cov3D_precomp[6 * (size_t)idx + k];
    }
    if (a.scales != nullptr && a.rotations != nullptr) { // (also WITH a precomputed covariance: Sigma^-1 is always built from these, forward.cu:208-220)
        sc_in = make_float3(a.scales[3 * (size_t)idx], a.scales[3 * (size_t)idx + 1], a.scales[3 * (size_t)idx + 2]);
        q_in = reinterpret_cast<const float4*>(a.rotations)[idx];
    }
    const float* __restrict__ view = a.view;
    // view-space position; near culling at z <= 0.2 (reference auxiliary.h:211-236)
    float3 pv;
    pv.x = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12] * 1.0f;
    pv.y = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13] * 1.0f;
    pv.z = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14] * 1.0f;
    if (pv.z <= 0.2f) {
        if (a.prefiltered) atomicOr(&a.g.status[1], 1u);
        return false;
    }

    // 3D covariance (reference forward_common.h:149-183) or the precomputed one
    float c3[6];
    if (a.cov3D_precomp != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; k++) c3[k] = c3_in[k];
    } else {
        const float3 sc = sc_in;
        const float4 q = q_in;
        const Mat3 S = mat_diag(a.scale_modifier * sc.x, a.scale_modifier * sc.y, a.scale_modifier * sc.z);
        const Mat3 Mm = mat_mul(S, quat_to_mat(q));
        const Mat3 Sig = mat_mul(mat_transpose(Mm), Mm);
        c3[0] = Sig.m[0][0]; c3[1] = Sig.m[0][1]; c3[2] = Sig.m[0][2]; c3[3] = Sig.m[1][1]; c3[4] = Sig.m[1][2]; c3[5] = Sig.m[2][2];
#pragma unroll
        for (int k = 0; k < 6; k++) a.g.cov3D[6 * (size_t)idx + k] = c3[k];
    }

    // EWA projection to a 2D covariance (reference forward_common.h:73-106)
    float3 t = pv;
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    Mat3 J;
    J.m[0][0] = a.focal_x / t.z; J.m[0][1] = 0.0f;              J.m[0][2] = -(a.focal_x * t.x) / (t.z * t.z);
    J.m[1][0] = 0.0f;              J.m[1][1] = a.focal_y / t.z; J.m[1][2] = -(a.focal_y * t.y) / (t.z * t.z);
    J.m[2][0] = 0.0f;              J.m[2][1] = 0.0f;              J.m[2][2] = 0.0f;
    Mat3 Wv;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Wv.m[i][j] = view[4 * i + j];
    const Mat3 T = mat_mul(mat_transpose(Wv), J);
    Mat3 Vrk;
    Vrk.m[0][0] = c3[0]; Vrk.m[0][1] = c3[1]; Vrk.m[0][2] = c3[2];
    Vrk.m[1][0] = c3[1]; Vrk.m[1][1] = c3[3]; Vrk.m[1][2] = c3[4];
    Vrk.m[2][0] = c3[2]; Vrk.m[2][1] = c3[4]; Vrk.m[2][2] = c3[5];
    const Mat3 cov = mat_mul(mat_mul(mat_transpose(T), mat_transpose(Vrk)), T);

    // low-pass dilation, optional Mip-Splatting opacity scaling (reference forward_common.h:108-131)
    float c2x = cov.m[0][0], c2y = cov.m[0][1], c2z = cov.m[1][1];
    c2x += 0.3f; c2z += 0.3f;
    const float det = c2x * c2z - c2y * c2y;
    float conv_scale = 1.0f;
    if (a.proper_ewa_scaling) {
        const float det_orig = cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[0][1];
        conv_scale = sqrtf(fmaxf(0.000025f, det_orig / det));
    }
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    const float4 co = make_float4(c2z * det_inv, -c2y * det_inv, c2x * det_inv, opacity * conv_scale);
    if (co.w < ALPHA_THRESHOLD) return false;

    // screen-space extent (reference forward.cu:151-164)
    const float thr = log_rounded(co.w / ALPHA_THRESHOLD);
    const float extent = a.tight_opacity_bounding ? (float)fmin(3.33, (double)sqrtf(2.0f * thr)) : 3.33f;
    const float mid = 0.5f * (c2x + c2z);
    const float lambda = mid + sqrtf(fmaxf(0.01f, mid * mid - det));
    const float radius = extent * sqrtf(lambda);
    if (radius <= 0.0f) return false;

    // projection of the mean (reference auxiliary.h:83-90; the 4x4 product sums (m0 x + m1 y) + (m2 z + m3 w))
    const float* __restrict__ proj = a.proj;
    float ph[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ph[j] = (proj[j] * mean.x + proj[4 + j] * mean.y) + (proj[8 + j] * mean.z + proj[12 + j] * 1.0f);
    const float p_w = 1.0f / (ph[3] + 0.0000001f);
    const float2 mean2D = make_float2(ndc_to_pix(ph[0] * p_w, a.W), ndc_to_pix(ph[1] * p_w, a.H));

    const float ext_x = fminf(a.rect_bounding ? (extent * sqrtf(c2x)) : radius, radius);
    const float ext_y = fminf(a.rect_bounding ? (extent * sqrtf(c2z)) : radius, radius);
    const float2 rect_dims = make_float2(ext_x, ext_y);
    // Visibility (radii, colours, Sigma^-1 ...) is decided on the FULL frame, exactly as in the reference;
    // tiles_touched counts only the tiles inside this rank's tile-row window [ty0, ty1).  Without
    // sharding the window is the whole frame and the two coincide.
    int fx0, fy0, fx1, fy1;
    get_rect(mean2D, rect_dims, a.gx, a.gy, 0, a.gy, fx0, fy0, fx1, fy1);
    if ((fx1 - fx0) * (fy1 - fy0) == 0) return false;
    o.sc = sc_in; o.q = q_in;
    o.mean = mean; o.pv = pv; o.co = co; o.thr = thr; o.radius = radius; o.mean2D = mean2D; o.rect_dims = rect_dims;
#pragma unroll
    for (int k = 0; k < 6; k++) o.c3[k] = c3[k];
    o.fx0 = fx0; o.fy0 = fy0; o.fx1 = fx1; o.fy1 = fy1;
    o.y0 = min(max(fy0, a.ty0), fy1); o.y1 = max(min(fy1, a.ty1), o.y0);
    return true;
}

__global__ void __launch_bounds__(256) preprocess_kernel(const PreArgs a)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    PreStageA o{};
    bool alive = false;
    if (idx < a.P) {
        a.radii[idx] = 0;
        a.g.tiles_touched[idx] = 0;
        alive = preprocess_stage_a(a, idx, o);
    }

    // ---- tiles touched.  With tile-based culling every tile of the rectangle is tested (reference
    // stopthepop_common.cuh:176-262).  A thread walks a small rectangle itself; a LARGE one (more than COOP_TILES tiles -- a
    // splat of hundreds of pixels covers thousands) is walked by the whole wave, one tile per lane, its owner's parameters
    // broadcast as wave-uniform scalars: the counterpart of the reference's warp-cooperative load balancing (:207-259),
    // built on 64-bit ballots.  Results are identical either way (same test per tile).
    const int x0 = o.fx0, x1 = o.fx1, y0 = o.y0, y1 = o.y1;
    int tile_count = (x1 - x0) * (y1 - y0);
    int full_count = 1;
    const bool tbc = a.tile_based_culling != 0;
    const int rect_tiles = (o.fx1 - o.fx0) * (o.fy1 - o.fy0);
    const bool coop = alive && tbc && a.tile_counts == nullptr && rect_tiles > COOP_TILES;
    if (alive && tbc && !coop) {
        tile_count = 0;
        full_count = 0;
        for (int y = o.fy0; y < o.fy1; y++)
            for (int x = o.fx0; x < o.fx1; x++) {
                const float2 tmin = make_float2((float)(x * TILE), (float)(y * TILE));
                const float2 tmax = make_float2((float)((x + 1) * TILE - 1), (float)((y + 1) * TILE - 1));
                float2 mp;
                const float f = max_contrib_power_rect(o.co, o.mean2D, tmin, tmax, (float)(TILE - 1), (float)(TILE - 1), mp);
                const int hit = (f <= o.thr) ? 1 : 0;
                full_count += hit;
                const int mine = (y >= y0 && y < y1) ? hit : 0;
                tile_count += mine;
                if (mine && a.tile_counts) atomicAdd(&a.tile_counts[y * a.gx + x], 1u);
            }
    } else if (alive && !tbc && a.tile_counts) {
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) atomicAdd(&a.tile_counts[y * a.gx + x], 1u);
    }
    {
        const int lane = wave_lane();
        unsigned long long todo = __ballot(coop);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const float4 co = make_float4(bcast_f(o.co.x, src), bcast_f(o.co.y, src), bcast_f(o.co.z, src), bcast_f(o.co.w, src));
            const float2 m2 = make_float2(bcast_f(o.mean2D.x, src), bcast_f(o.mean2D.y, src));
            const float thr = bcast_f(o.thr, src);
            const int sx0 = bcast_i(o.fx0, src), sy0 = bcast_i(o.fy0, src), sx1 = bcast_i(o.fx1, src), sy1 = bcast_i(o.fy1, src);
            const int wy0 = bcast_i(o.y0, src), wy1 = bcast_i(o.y1, src);
            const int w = sx1 - sx0, n = w * (sy1 - sy0);
            int full = 0, mine = 0;
            for (int t0 = 0; t0 < n; t0 += 64) {
                const int t = t0 + lane;
                const int ty = t / w, y = sy0 + ty, x = sx0 + (t - ty * w);
                bool hit = false;
                if (t < n) {
                    const float2 tmin = make_float2((float)(x * TILE), (float)(y * TILE));
                    const float2 tmax = make_float2((float)((x + 1) * TILE - 1), (float)((y + 1) * TILE - 1));
                    float2 mp;
                    hit = max_contrib_power_rect(co, m2, tmin, tmax, (float)(TILE - 1), (float)(TILE - 1), mp) <= thr;
                }
                full += (int)__popcll(__ballot(hit));
                mine += (int)__popcll(__ballot(hit && y >= wy0 && y < wy1));
            }
            if (lane == src) { full_count = full; tile_count = mine; }
        }
    }
    // (no early return: the workgroup's threads meet again for the scan at the end)
    uint32_t my_entries = 0u;
    if (alive && full_count != 0) {
        my_entries = (uint32_t)tile_count;

    const float3 mean = o.mean, pv = o.pv;
    const float4 co = o.co;
    const float2 mean2D = o.mean2D, rect_dims = o.rect_dims;
    const float radius = o.radius;
    const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    // (SH -> RGB is sh_color_kernel's: it runs BEHIND the num_rendered read-back, see stp_forward)

    if (a.g.cov3D_inv != nullptr) { // reference forward.cu:208-220, stopthepop_common.cuh:13-41 (needs scales + rotations: stp_forward checks)
        const float3 sc = o.sc;
        const float4 q = o.q;
        const Mat3 S = mat_diag(1.f / (a.scale_modifier * fmaxf(1e-3f, sc.x)), 1.f / (a.scale_modifier * fmaxf(1e-3f, sc.y)),
                                1.f / (a.scale_modifier * fmaxf(1e-3f, sc.z)));
        const Mat3 Mm = mat_mul(S, quat_to_mat(q));
        const Mat3 inv = mat_mul(mat_transpose(Mm), Mm);
        const float dx = cam.x - mean.x, dy = cam.y - mean.y, dz = cam.z - mean.z;
        const float ux = (-inv.m[0][0]) * dx + (-inv.m[1][0]) * dy + (-inv.m[2][0]) * dz;
        const float uy = (-inv.m[0][1]) * dx + (-inv.m[1][1]) * dy + (-inv.m[2][1]) * dz;
        const float uz = (-inv.m[0][2]) * dx + (-inv.m[1][2]) * dy + (-inv.m[2][2]) * dz;
        // (a frame in which this never fires -- every frame of a sane scene -- has depth-key denominators below 9.1e36: its
        // render kernels take the reciprocal without the domain check, stp_device.h rcp_ieee<true>)
        const bool tame = fabsf(inv.m[0][0]) < 1.0e36f && fabsf(inv.m[0][1]) < 1.0e36f && fabsf(inv.m[0][2]) < 1.0e36f &&
                          fabsf(inv.m[1][1]) < 1.0e36f && fabsf(inv.m[1][2]) < 1.0e36f && fabsf(inv.m[2][2]) < 1.0e36f; // (false for NaN)
        if (!tame) atomicOr(&a.g.status[1], 2u);
        a.g.cov3D_inv[3 * (size_t)idx + 0] = make_float4(inv.m[0][0], inv.m[0][1], inv.m[0][2], 0.0f);
        a.g.cov3D_inv[3 * (size_t)idx + 1] = make_float4(inv.m[1][1], inv.m[1][2], inv.m[2][2], 0.0f);
        a.g.cov3D_inv[3 * (size_t)idx + 2] = make_float4(ux, uy, uz, 0.0f);
    }

    float depth;
    if (a.sort_order == ORDER_Z) depth = pv.z;
    else {
        const float dx = cam.x - mean.x, dy = cam.y - mean.y, dz = cam.z - mean.z;
        depth = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    a.g.depths[idx] = depth;
    a.radii[idx] = (int)ceilf(radius);
    a.g.rects2D[idx] = rect_dims;
    a.g.means2D[idx] = mean2D;
    a.g.conic_opacity[idx] = co;
    a.g.tiles_touched[idx] = (uint32_t)tile_count;
    if (a.g.gpack != nullptr) { // the same values once more, as one 64-byte line per Gaussian (see gather_entries_kernel)
        const float4 s0 = a.g.cov3D_inv[3 * (size_t)idx + 0], s1 = a.g.cov3D_inv[3 * (size_t)idx + 1], s2 = a.g.cov3D_inv[3 * (size_t)idx + 2];
        float4* const gp = a.g.gpack + 4 * (size_t)idx;
        gp[0] = make_float4(s0.x, s0.y, s0.z, s1.x);
        gp[1] = make_float4(s1.y, s1.z, s2.x, s2.y);
        gp[2] = make_float4(s2.z, mean2D.x, mean2D.y, 0.0f);
        gp[3] = co;
    }
    }

    // ---- first level of the scan tiles_touched -> point_offsets (reference rasterizer_impl.cu:313, cub InclusiveSum): inclusive inside the
    // workgroup, the workgroup's total to block_sums; block_prefix_mailbox_kernel scans the totals and duplicate_kernel adds the two levels.
    // A device-wide scan kernel (rocPRIM: look-back state initialisation + scan, 18 us at 1 M Gaussians) is not needed for this.
    if (a.g.block_sums != nullptr) {
        __shared__ uint32_t s_wave_total[4];
        const int lane = wave_lane(), wave = (int)(threadIdx.x >> 6);
        uint32_t v = my_entries;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) s_wave_total[wave] = v;
        __syncthreads();
        for (int k = 0; k < wave; k++) v += s_wave_total[k];
        if (idx < a.P) a.g.point_offsets[idx] = v;
        if (threadIdx.x == 255) a.g.block_sums[blockIdx.x] = v;
    }
}

// SH -> RGB for the visible Gaussians (reference computeColorFromSH, forward_common.h:20-70, called from preprocessCUDA
// forward.cu:197-205).  A kernel of its own for two reasons: (1) nothing before the render stage needs the colours, so
// the host enqueues it AFTER the num_rendered mailbox write and it keeps the GPU busy while the host wakes up, sizes
// the binning buffer and launches the next stages (the read-back bubble of the reference's flow, rasterizer_impl.cu:317,
// disappears behind it); (2) a thread-per-Gaussian read of 3M consecutive floats is a 192-byte-stride gather: the
// workgroup moves its 256 rows through LDS with coalesced 16-byte loads instead (rows padded to 3M+1 words), exactly as
// preprocess_backward_kernel does.
struct ShArgs {
    int P, D, M;
    const float* means3D;
    const float* shs;
    const float* cam;
    const int* radii;
    uint8_t* clamped;
    float* rgb;
};

#ifndef STP_SH_BLOCK
#define STP_SH_BLOCK 256
#endif
constexpr int SH_BLOCK = STP_SH_BLOCK; // Gaussians (= threads) per workgroup: its LDS staging area is SH_BLOCK x (3M + 1) words

__global__ void __launch_bounds__(SH_BLOCK) sh_color_kernel(const ShArgs a)
{
    extern __shared__ float s_rows[]; // [SH_BLOCK][3M + 1]
    const int tid = (int)threadIdx.x;
    const int base = (int)blockIdx.x * SH_BLOCK;
    const int idx = base + tid;
    const int rows = min(SH_BLOCK, a.P - base);
    const int row_len = 3 * a.M, row_stride = row_len + 1;
    // Order of the loads matters: the coefficient rows FIRST (twelve 16-byte loads per thread in flight), the Gaussian's
    // radius and mean behind them -- one memory latency per workgroup instead of three dependent ones.  (No early exit on
    // "nobody visible" either: it would put the radius in front again.)
    const float* __restrict__ src = a.shs + (size_t)base * row_len;
    const int total = rows * row_len;
    const bool fast = row_len == 48 && rows == SH_BLOCK; // SH degree 3 (M = 16), full block
    float4 v48[12];
    if (fast) stage_rows_load<SH_BLOCK, 48>(src, tid, v48);
    const int ic = min(idx, a.P - 1); // (clamped, unconditional loads: no branch between them, both in flight together)
    const int radius_ic = a.radii[ic];
    const float3 mean = make_float3(a.means3D[3 * (size_t)ic], a.means3D[3 * (size_t)ic + 1], a.means3D[3 * (size_t)ic + 2]);
    const int my_radius = idx < a.P ? radius_ic : 0;
    if (fast) {
        stage_rows_store<SH_BLOCK, 48>(s_rows, tid, v48);
    } else if ((row_len & 3) == 0) {
        for (int f = 4 * tid; f < total; f += 4 * SH_BLOCK) {
            const float4 v = *reinterpret_cast<const float4*>(src + f);
            const int r = f / row_len, j = f - r * row_len;
            float* d = s_rows + r * row_stride + j;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int f = tid; f < total; f += SH_BLOCK) {
            const int r = f / row_len, j = f - r * row_len;
            s_rows[r * row_stride + j] = src[f];
        }
    }
    __syncthreads();
    if (my_radius > 0) {
        const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
        sh_to_rgb(idx, a.D, mean, cam, s_rows + tid * row_stride, a.clamped, a.rgb);
    }
}

struct DupArgs {
    int P, W, H, gx, gy, ty0, ty1;
    int sort_order, tile_based_culling;
    const float* inv_vp;
    const float* cam;
    const int* radii;
    GeometryState g;
    uint64_t* keys;
    uint32_t* values;
    uint32_t* tile_cursor; // binning by tile counters: next free slot of every tile's segment (nullptr: slots by point_offsets)
    uint32_t cap;          // slots the arrays hold: a run-ahead forward (stp_api.hip) launches on a capacity, not on num_rendered; 0xFFFFFFFF = exact
    int n_gauss_blocks;    // workgroups that own Gaussians; the DUP_PAD_BLOCKS behind them fill [num_rendered, cap) with padding entries
    int n_pad_blocks;      // DUP_PAD_BLOCKS for a capacity launch, else 0
    uint32_t* header;      // the binning buffer's header (stp_api.hip: buffer headers), written by the first thread
    uint32_t header_cap;   // entries the buffer was carved for
    uint32_t* zero_ptr;    // cleared by the DUP_ZERO_BLOCKS workgroups behind those: the tile-bit sort's histograms, look-back states and block
    uint32_t zero_words;   // counters (stp_binning.hip: sort_zero_region), which the library's own driver clears with five separate fill launches
};
constexpr int DUP_PAD_BLOCKS = 128, DUP_ZERO_BLOCKS = 32;

// key + write decision of ONE (Gaussian, tile) pair: reference duplicateWithKeys_extended, stopthepop_common.cuh:420-460
struct DupGaussian { float2 xy; float4 co; float thr; float3 p0, p1, p2; float global_depth; };
__device__ __forceinline__ bool duplicate_tile(const DupArgs& a, const DupGaussian& g, float3 cam, bool tbc, bool eval_max, bool per_tile_depth,
                                               int x, int y, uint64_t& key)
{
#pragma clang fp contract(off)
    const float2 tmin = make_float2((float)(x * TILE), (float)(y * TILE));
    const float2 tmax = make_float2((float)((x + 1) * TILE - 1), (float)((y + 1) * TILE - 1));
    float2 max_pos = make_float2(0, 0);
    float max_fac = 0.0f;
    if (eval_max) max_fac = max_contrib_power_rect(g.co, g.xy, tmin, tmax, (float)(TILE - 1), (float)(TILE - 1), max_pos);
    float depth = g.global_depth;
    if (per_tile_depth) { // reference stopthepop_common.cuh:439-449
        const float2 center = make_float2((tmin.x + tmax.x) * 0.5f, (tmin.y + tmax.y) * 0.5f);
        const float2 target = (a.sort_order == ORDER_PTD_MAX) ? max_pos : center;
        const float3 dir = view_ray(a.inv_vp, cam, target.x, target.y, a.W, a.H);
        depth = fmaxf(0.0f, depth_along_ray(g.p0, g.p1, g.p2, dir) + 8.0f);
    }
    key = make_sort_key((uint32_t)(y * a.gx + x), depth);
    return !tbc || max_fac <= g.thr;
}

__global__ void __launch_bounds__(256) duplicate_kernel(const DupArgs a)
{
#pragma clang fp contract(off)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.header) { a.header[0] = STP_HEADER_MAGIC_BINNING; a.header[1] = a.header_cap; a.header[2] = ~a.header_cap; a.header[3] = 0u; }
    if ((int)blockIdx.x >= a.n_gauss_blocks + a.n_pad_blocks) {
        uint4* const z = reinterpret_cast<uint4*>(a.zero_ptr); // (256-byte aligned, a multiple of 64 words)
        for (uint32_t i = ((uint32_t)blockIdx.x - (uint32_t)(a.n_gauss_blocks + a.n_pad_blocks)) * 256u + threadIdx.x; i < a.zero_words / 4u; i += (uint32_t)DUP_ZERO_BLOCKS * 256u)
            z[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    if ((int)blockIdx.x >= a.n_gauss_blocks) {
        // run-ahead forward: the sort and range passes behind this kernel run over `cap` slots; the slots behind the frame's true count (known
        // on the device: the two-level scan's grand total) become padding entries, which sort behind every tile (an overflowing frame -- count
        // above cap -- pads nothing, writes nothing out of bounds and is redone by the host with the exact size)
        const int nb = a.n_gauss_blocks;
        const uint32_t total = a.g.block_prefix[nb - 1] + a.g.block_sums[nb - 1];
        for (uint64_t i = (uint64_t)total + ((uint32_t)blockIdx.x - (uint32_t)nb) * 256u + threadIdx.x; i < (uint64_t)a.cap; i += (uint64_t)DUP_PAD_BLOCKS * 256u) {
            a.values[i] = 0xFFFFFFFFu;
            a.keys[i] = make_sort_key(INVALID_TILE_ID, FLT_MAX);
        }
        return;
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = idx < a.P && a.radii[idx] > 0;   // (no early return: the wave's lanes meet again for the large rectangles)
    const bool tbc = a.tile_based_culling != 0;
    const bool per_tile_depth = a.sort_order == ORDER_PTD_CENTER || a.sort_order == ORDER_PTD_MAX;
    const bool eval_max = tbc || a.sort_order == ORDER_PTD_MAX;
    uint32_t off = 0, off_to = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    DupGaussian g{};
    float3 cam = make_float3(0, 0, 0);
    if (per_tile_depth) cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    if (a.g.block_prefix != nullptr) {
        // second half of the two-level scan: workgroup prefix + the inclusive value inside the workgroup (preprocess_kernel, same 256 Gaussians);
        // the global inclusive scan is left in point_offsets (what the reference's buffer holds) once every thread has read its neighbour's value
        if (idx < a.P) {
            const uint32_t base = a.g.block_prefix[blockIdx.x];
            off = base + (threadIdx.x == 0 ? 0u : a.g.point_offsets[idx - 1]);
            off_to = base + a.g.point_offsets[idx];
        }
        __syncthreads();
        if (idx < a.P) a.g.point_offsets[idx] = off_to;
    } else if (valid) {
        off = (idx == 0) ? 0u : a.g.point_offsets[idx - 1];
        off_to = a.g.point_offsets[idx];
    }
    if (valid) {
        g.xy = a.g.means2D[idx];
        const float2 ext = a.g.rects2D[idx];
        get_rect(g.xy, ext, a.gx, a.gy, a.ty0, a.ty1, x0, y0, x1, y1);
        if (eval_max) {
            g.co = a.g.conic_opacity[idx];
            g.thr = log_rounded(g.co.w / ALPHA_THRESHOLD);
        }
        if (per_tile_depth) {
            g.p0 = f4_xyz(a.g.cov3D_inv[3 * (size_t)idx + 0]);
            g.p1 = f4_xyz(a.g.cov3D_inv[3 * (size_t)idx + 1]);
            g.p2 = f4_xyz(a.g.cov3D_inv[3 * (size_t)idx + 2]);
        }
        g.global_depth = a.g.depths[idx];
    }
    // A rectangle of more than COOP_TILES tiles is walked by the whole wave (one tile per lane, the owner's data as
    // wave-uniform scalars, write slots from a ballot prefix count so that the row-major order of the sequential loop is
    // kept): the counterpart of the reference's warp-cooperative duplication (stopthepop_common.cuh:510-621) on 64 lanes.
    const bool coop = valid && a.tile_cursor == nullptr && (x1 - x0) * (y1 - y0) > COOP_TILES;
    if (valid && !coop) {
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                uint64_t key;
                if (duplicate_tile(a, g, cam, tbc, eval_max, per_tile_depth, x, y, key)) {
                    if (a.tile_cursor) { // straight into the tile's segment; the order inside it is settled by the tile sort
                        const uint32_t slot = atomicAdd(&a.tile_cursor[(uint32_t)(key >> 32)], 1u);
                        a.values[slot] = (uint32_t)idx;
                        a.keys[slot] = key;
                    } else if (off < off_to && off < a.cap) {
                        a.values[off] = (uint32_t)idx;
                        a.keys[off] = key;
                    }
                    off++;
                }
            }
        if (!a.tile_cursor) // pad what the (slightly more generous) preprocess count reserved but culling did not use
            for (; off < off_to && off < a.cap; off++) { // (reference stopthepop_common.cuh:503-508, 614-619)
                a.values[off] = 0xFFFFFFFFu;
                a.keys[off] = make_sort_key(INVALID_TILE_ID, FLT_MAX);
            }
    }
    const int lane = wave_lane();
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        DupGaussian s;
        s.xy = make_float2(bcast_f(g.xy.x, src), bcast_f(g.xy.y, src));
        s.co = make_float4(bcast_f(g.co.x, src), bcast_f(g.co.y, src), bcast_f(g.co.z, src), bcast_f(g.co.w, src));
        s.thr = bcast_f(g.thr, src);
        s.p0 = make_float3(bcast_f(g.p0.x, src), bcast_f(g.p0.y, src), bcast_f(g.p0.z, src));
        s.p1 = make_float3(bcast_f(g.p1.x, src), bcast_f(g.p1.y, src), bcast_f(g.p1.z, src));
        s.p2 = make_float3(bcast_f(g.p2.x, src), bcast_f(g.p2.y, src), bcast_f(g.p2.z, src));
        s.global_depth = bcast_f(g.global_depth, src);
        const int sx0 = bcast_i(x0, src), sy0 = bcast_i(y0, src), sx1 = bcast_i(x1, src), sy1 = bcast_i(y1, src);
        const uint32_t s_to = (uint32_t)bcast_i((int)off_to, src), s_idx = (uint32_t)bcast_i(idx, src);
        uint32_t base = (uint32_t)bcast_i((int)off, src);
        const int w = sx1 - sx0, n = w * (sy1 - sy0);
        for (int t0 = 0; t0 < n; t0 += 64) {
            const int t = t0 + lane;
            const int ty = t / w;
            uint64_t key = 0;
            const bool wr = t < n && duplicate_tile(a, s, cam, tbc, eval_max, per_tile_depth, sx0 + (t - ty * w), sy0 + ty, key);
            const unsigned long long wm = __ballot(wr);
            const uint32_t slot = base + (uint32_t)lanes_below(wm);
            if (wr && slot < s_to && slot < a.cap) { a.values[slot] = s_idx; a.keys[slot] = key; }
            base += (uint32_t)__popcll(wm);
        }
        for (uint32_t p = base + (uint32_t)lane; p < s_to && p < a.cap; p += 64u) {
            a.values[p] = 0xFFFFFFFFu;
            a.keys[p] = make_sort_key(INVALID_TILE_ID, FLT_MAX);
        }
    }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges, uint32_t T)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    const bool valid = cur != INVALID_TILE_ID && cur < T;
    if (idx == 0) {
        if (valid) ranges[cur].x = 0;
    } else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            if (prev < T) ranges[prev].y = (uint32_t)idx;
            if (valid) ranges[cur].x = (uint32_t)idx;
        }
    }
    if (idx == L - 1 && valid) ranges[cur].y = (uint32_t)L;
}

// Workgroups of a launch are dispatched in index order, and a render workgroup's time is its tile's list: with the longest lists LAST (wherever
// the scene puts them) the launch ends with a few long workgroups on an otherwise idle chip.  One workgroup orders the window's tiles by list
// length, longest first (counting sort on 256 length classes: 16 entries apiece up to 2048, 128 apiece beyond); the render kernels' workgroup j
// then takes tile order[j].  The entry records are per tile: nothing the render kernels read is shared between neighbouring tiles, so the XCD-aware
// contiguous order they use otherwise buys them nothing.
// gather_order (second array, may be nullptr): the same idea for the entry gather, whose workgroups DO share data between neighbouring tiles (the
// Gaussians' 64-byte lines, in the XCD's L2): every XCD keeps its contiguous run of tiles (workgroup j runs on XCD j mod 8) and takes them longest
// first INSIDE the run -- workgroup 8 k + x gets the k-th longest tile of XCD x's run.
__global__ void __launch_bounds__(1024) tile_order_kernel(int tile0, int T, const uint2* __restrict__ ranges_, uint32_t* __restrict__ order, uint32_t* __restrict__ gather_order)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_hx[8 * 256];
    const uint2* __restrict__ ranges = ranges_ + tile0;
    const int tid = (int)threadIdx.x;
    auto klass = [](uint32_t len) { return 255u - (len < 2048u ? (len >> 4) : min(127u, (len - 2048u) >> 7) + 128u); }; // 0 = the longest
    const int xq = T >> 3, xr = T & 7; // XCD x owns the tiles [x (xq + 1), ...) for x < xr, [xr (xq + 1) + (x - xr) xq, ...) beyond (the kernels' own map)
    auto xcd_of = [&](int t) { const int cut = xr * (xq + 1); return t < cut ? t / (xq + 1) : xr + (xq ? (t - cut) / xq : 0); };
    if (tid < 256) s_hist[tid] = 0u;
    for (int i = tid; i < 8 * 256; i += 1024) s_hx[i] = 0u;
    __syncthreads();
    for (int t = tid; t < T; t += 1024) {
        const uint32_t c = klass(ranges[t].y - ranges[t].x);
        atomicAdd(&s_hist[c], 1u);
        if (gather_order) atomicAdd(&s_hx[xcd_of(t) * 256 + c], 1u);
    }
    __syncthreads();
    // exclusive scans of 256 counters by one wave each, four per lane: wave 0 the frame's, waves 1..8 the XCDs'
    const int wv = tid >> 6, ln = tid & 63;
    if (wv < (gather_order ? 9 : 1)) {
        uint32_t* const h = wv == 0 ? s_hist : s_hx + (wv - 1) * 256;
        uint32_t c[4], sum = 0;
        for (int k = 0; k < 4; k++) { c[k] = h[4 * ln + k]; sum += c[k]; }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (ln >= o) inc += v; }
        uint32_t run = inc - sum;
        for (int k = 0; k < 4; k++) { h[4 * ln + k] = run; run += c[k]; }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 1024) {
        const uint32_t c = klass(ranges[t].y - ranges[t].x);
        order[atomicAdd(&s_hist[c], 1u)] = (uint32_t)t;
        if (gather_order) { const int x = xcd_of(t); gather_order[8u * atomicAdd(&s_hx[x * 256 + c], 1u) + (uint32_t)x] = (uint32_t)t; }
    }
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float z = view[2] * means3D[3 * (size_t)idx] + view[6] * means3D[3 * (size_t)idx + 1] + view[10] * means3D[3 * (size_t)idx + 2] + view[14] * 1.0f;
    present[idx] = z > 0.2f ? 1 : 0;
}

} // namespace

hipError_t launch_preprocess(const FrameParams& f, const GeometryState& g, int* radii, uint32_t* tile_counts, hipStream_t st)
{
    PreArgs a;
    a.P = f.P; a.D = f.D; a.M = f.M; a.W = f.W; a.H = f.H; a.gx = f.gx; a.gy = f.gy; a.ty0 = f.ty0; a.ty1 = f.ty1;
    a.focal_x = f.focal_x; a.focal_y = f.focal_y; a.tan_fovx = f.tan_fovx; a.tan_fovy = f.tan_fovy; a.scale_modifier = f.scale_modifier;
    a.sort_order = f.s.sort_order; a.rect_bounding = f.s.rect_bounding; a.tight_opacity_bounding = f.s.tight_opacity_bounding;
    a.tile_based_culling = f.s.tile_based_culling; a.proper_ewa_scaling = f.s.proper_ewa_scaling; a.prefiltered = f.prefiltered;
    a.means3D = f.means3D; a.scales = f.scales; a.rotations = f.rotations; a.opacities = f.opacities; a.shs = f.shs;
    a.cov3D_precomp = f.cov3D_precomp; a.colors_precomp = f.colors_precomp; a.view = f.viewmatrix; a.proj = f.projmatrix; a.cam = f.cam_pos;
    a.radii = radii; a.g = g; a.tile_counts = tile_counts;
    hipLaunchKernelGGL(preprocess_kernel, dim3((f.P + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_duplicate(const FrameParams& f, const GeometryState& g, const int* radii, const BinningState& b, uint32_t* tile_cursor, uint32_t cap, uint32_t header_cap,
                            uint32_t* zero_ptr, size_t zero_words, hipStream_t st)
{
    DupArgs a;
    a.P = f.P; a.W = f.W; a.H = f.H; a.gx = f.gx; a.gy = f.gy; a.ty0 = f.ty0; a.ty1 = f.ty1;
    a.sort_order = f.s.sort_order; a.tile_based_culling = f.s.tile_based_culling;
    a.inv_vp = f.inv_viewprojmatrix; a.cam = f.cam_pos; a.radii = radii; a.g = g;
    a.tile_cursor = tile_cursor;
    a.keys = tile_cursor ? b.keys : b.keys_unsorted; a.values = tile_cursor ? b.point_list : b.point_list_unsorted;
    const bool capped = cap != 0xFFFFFFFFu && g.block_prefix != nullptr && tile_cursor == nullptr; // (a capacity launch needs the count on the device)
    a.cap = capped ? cap : 0xFFFFFFFFu;
    a.n_gauss_blocks = (f.P + 255) / 256;
    a.n_pad_blocks = capped ? DUP_PAD_BLOCKS : 0;
    a.zero_ptr = zero_ptr; a.zero_words = (uint32_t)zero_words;
    a.header = b.header; a.header_cap = header_cap;
    hipLaunchKernelGGL(duplicate_kernel, dim3(a.n_gauss_blocks + a.n_pad_blocks + (zero_words ? DUP_ZERO_BLOCKS : 0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sh_color(const FrameParams& f, const GeometryState& g, const int* radii, hipStream_t st)
{
    if (f.colors_precomp != nullptr || f.shs == nullptr || f.M <= 0) return hipSuccess; // reference forward.cu:200: precomputed colours win
    ShArgs a;
    a.P = f.P; a.D = f.D; a.M = f.M; a.means3D = f.means3D; a.shs = f.shs; a.cam = f.cam_pos; a.radii = radii; a.clamped = g.clamped; a.rgb = g.rgb;
    const size_t lds = (size_t)SH_BLOCK * (3 * a.M + 1) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sh_color_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sh_color_kernel, dim3((f.P + SH_BLOCK - 1) / SH_BLOCK), dim3(SH_BLOCK), lds, st, a);
    return hipGetLastError();
}

// num_rendered (the scan's last element) and the status word into host-mapped memory: the forward's one host hand-over
__global__ void mailbox_kernel(const uint32_t* __restrict__ last_offset, const uint32_t* __restrict__ status, volatile uint32_t* mailbox, uint32_t ticket,
                               uint32_t* log_need)
{
    mailbox[0] = *last_offset;
    mailbox[1] = *status;
    mailbox[3] = log_need ? atomicExch(log_need, 0u) : 0u; // what the recording forwards before this one reported (stp_blend.h: report_log_need)
    __threadfence_system();
    mailbox[2] = ticket; // the host may be watching this word (stp_forward): it goes out after the two values
    __threadfence_system();
}

// Second level of the scan + the hand-over, one workgroup: exclusive scan of the preprocess workgroups' totals (thread t takes a run of
// consecutive totals), num_rendered = the grand total and the status word into host-mapped memory, the ticket last (see mailbox_kernel).
__global__ void __launch_bounds__(1024) block_prefix_mailbox_kernel(const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_prefix, int n_blocks,
                                                                     const uint32_t* __restrict__ status, volatile uint32_t* mailbox, uint32_t ticket,
                                                                     uint32_t* log_need)
{
    __shared__ uint32_t s_wave_total[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_blocks + 1023) / 1024, i0 = min(tid * per, n_blocks), i1 = min(i0 + per, n_blocks);
    uint32_t mine = 0u;
    for (int i = i0; i < i1; i++) mine += block_sums[i];
    uint32_t v = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_wave_total[wave] = v;
    __syncthreads();
    for (int k = 0; k < wave; k++) v += s_wave_total[k];
    uint32_t run = v - mine; // exclusive prefix of this thread's run
    for (int i = i0; i < i1; i++) { block_prefix[i] = run; run += block_sums[i]; }
    if (tid == 1023) {
        mailbox[0] = v;
        mailbox[1] = status[1];
        mailbox[3] = log_need ? atomicExch(log_need, 0u) : 0u; // (see mailbox_kernel)
        __threadfence_system();
        mailbox[2] = ticket;
        __threadfence_system();
    }
}

hipError_t launch_block_prefix_mailbox(const FrameParams& f, const GeometryState& g, uint32_t* mailbox_dev, uint32_t ticket, uint32_t* log_need, hipStream_t st)
{
    hipLaunchKernelGGL(block_prefix_mailbox_kernel, dim3(1), dim3(1024), 0, st, g.block_sums, g.block_prefix, (f.P + 255) / 256, g.status, mailbox_dev, ticket, log_need);
    return hipGetLastError();
}

hipError_t launch_mailbox(const uint32_t* last_offset, const uint32_t* status, uint32_t* mailbox_dev, uint32_t ticket, uint32_t* log_need, hipStream_t st)
{
    hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(1), 0, st, last_offset, status, mailbox_dev, ticket, log_need);
    return hipGetLastError();
}

// One launch for the three small clears of a forward (each hipMemsetAsync is a 5 us fill kernel of its own): the status
// words, the tile ranges (reference rasterizer_impl.cu:354) and the tile flags (0 = "this tile's log is valid", all ones =
// "the forward recorded no log", see carve_image).
__global__ void __launch_bounds__(256) frame_init_kernel(uint32_t* __restrict__ status, uint2* __restrict__ ranges, uint32_t* __restrict__ tile_flags,
                                                          uint32_t flag_value, uint32_t* __restrict__ tile_counts, int tile0, int T,
                                                          uint32_t* __restrict__ header, uint32_t log_depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64) status[i] = 0u;
    if (i == 0) { header[0] = STP_HEADER_MAGIC_IMAGE; header[1] = log_depth; header[2] = ~log_depth; header[3] = 0u; } // the image buffer describes itself (stp_api.hip: buffer headers)
    if (i < T) { // the tiles of the frame's tile-row window (the arrays hold no others: carve_image)
        ranges[tile0 + i] = make_uint2(0u, 0u);
        tile_flags[tile0 + i] = flag_value;
        if (tile_counts) tile_counts[tile0 + i] = 0u;
    }
}

hipError_t launch_frame_init(const GeometryState& g, const ImageState& img, int tile0, int T, bool with_log, bool tile_counters, hipStream_t st)
{
    const int n = T > 64 ? T : 64;
    hipLaunchKernelGGL(frame_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g.status, img.ranges, img.tile_flags, with_log ? 0u : 0xFFFFFFFFu,
                       tile_counters ? img.tile_counts : nullptr, tile0, T, img.header, (uint32_t)(with_log ? img.log_depth : 0));
    return hipGetLastError();
}

hipError_t launch_ranges(const FrameParams& f, const BinningState& b, const ImageState& img, int R, hipStream_t st)
{
    const size_t T = (size_t)f.gx * f.gy;
    hipError_t e = hipSuccess; // (the ranges were zeroed by frame_init_kernel)
    if (R > 0) {
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, st, R, b.keys, img.ranges, (uint32_t)T);
        e = hipGetLastError();
    }
    return e;
}

bool tile_order_enabled()
{
    // MEASURED (round 6, one box, alternating, profiles/r06_experiments/tile_order_ab.txt; forward / replay ms): C2-full 0.855 / 0.855 -> 0.822 / 0.824 (432 -> 443
    // frames/s: even the homogeneous frame ends on a tail), C2L (40 % of the Gaussians in 12 clusters) 1.03 / 0.98 -> 0.75 / 0.72 (373 -> 465 frames/s), C2H
    // 0.96 / 0.855 -> 0.61 / 0.51, C3 +1.3 %, C5 unchanged; the order kernel itself 9 us.  STP_TILE_ORDER=0: the XCD-contiguous order of rounds 1-5.
    static const bool on = [] { const char* e = std::getenv("STP_TILE_ORDER"); return !(e && e[0] == '0'); }();
    return on;
}
// STP_GATHER_ORDER (experiments, round 6; default 0): 0 = the entry gather in the XCD-contiguous spatial order, 1 = longest first inside every XCD's run, 2 = longest first over
// the frame.  MEASURED (sort stage ms, one box, alternating): 1: C2-full 0.320 -> 0.324, C5 1.110 -> 1.172, C3 0.970 -> 0.955, C2L 0.381 -> 0.372; 2: C2-full 0.311 -> 0.320,
// C5 1.14 -> 1.21, C3 0.944 -> 0.900, C2L 0.378 -> 0.328 -- what the gather gains at the tail it loses in the L2 (neighbouring tiles share their Gaussians' lines).
bool gather_order_enabled()
{
    return gather_order_mode() != 0;
}
int gather_order_mode()
{
    static const int m = [] { const char* e = std::getenv("STP_GATHER_ORDER"); return e ? std::atoi(e) : 0; }();
    return m;
}
hipError_t launch_tile_order(const FrameParams& f, const ImageState& img, hipStream_t st)
{
    const int T = f.gx * (f.ty1 - f.ty0);
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, f.gx * f.ty0, T, img.ranges, img.tile_cursor + f.gx * f.ty0,
                       gather_order_enabled() ? img.tile_counts + f.gx * f.ty0 : nullptr);
    return hipGetLastError();
}

// Entry data in list order for the per-pixel-sort render kernels (BinningState::entA..entF): one thread per tile-list
// entry gathers its Gaussian's Sigma^-1 pack, conic/opacity, 2D mean and colour ONCE; afterwards every consumer
// (batch staging, mid and head feeds, blends, the replay backward) reads them with unit stride inside its tile's
// segment instead of gathering by Gaussian id from four arrays each time.
__global__ void __launch_bounds__(256) gather_entries_kernel(int R, const uint32_t* __restrict__ point_list, const uint64_t* __restrict__ keys,
                                                             const float4* __restrict__ gpack, const float* __restrict__ features,
                                                             float4* __restrict__ entA, float4* __restrict__ entB, float4* __restrict__ entC,
                                                             float4* __restrict__ entD, float4* __restrict__ entF, int mask_kind, int gx, uint32_t T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int id = (int)point_list[i];
    if (id < 0) return; // (a padding entry behind the last tile's segment)
    const float4* __restrict__ gp = gpack + 4 * (size_t)id; // one 64-byte line written by preprocess_kernel
    const float4 pa = gp[0], pb = gp[1], pc = gp[2], pd = gp[3];
    entA[i] = pa;
    entB[i] = pb;
    entC[i] = make_float4(pc.x, pc.y, pc.z, __int_as_float(id));
    entD[i] = pd;
    float spare = 0.0f;
    const uint32_t tile = (uint32_t)(keys[i] >> 32);
    if (mask_kind && tile < T) spare = __uint_as_float(subtile_mask(mask_kind, pd, make_float2(pc.y, pc.z), (int)(tile % (uint32_t)gx), (int)(tile / (uint32_t)gx)));
    entF[i] = make_float4(features[3 * (size_t)id], features[3 * (size_t)id + 1], features[3 * (size_t)id + 2], spare);
}

hipError_t launch_gather_entries(const FrameParams& f, const GeometryState& g, const BinningState& b, int R, hipStream_t st)
{
    if (R <= 0 || (f.s.sort_mode != MODE_HIER && f.s.sort_mode != MODE_KBUFFER)) return hipSuccess;
    const float* features = f.colors_precomp ? f.colors_precomp : g.rgb;
    hipLaunchKernelGGL(gather_entries_kernel, dim3((R + 255) / 256), dim3(256), 0, st, R, b.point_list, b.keys, g.gpack, features, b.entA, b.entB, b.entC,
                       b.entD, b.entF, subtile_mask_kind(f.s), f.gx, (uint32_t)(f.gx * f.gy));
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t st)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, viewmatrix, present);
    return hipGetLastError();
}

} // namespace stp
