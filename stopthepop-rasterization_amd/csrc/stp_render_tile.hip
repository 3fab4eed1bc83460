// stp_render_tile.hip -- "one thread per pixel, tile list streamed through LDS" render kernels.
//
// Replaces (reference file:line under cuda_rasterizer/):
//   renderCUDA<3,false>            forward.cu:234-366            -> render_global_fwd_kernel
//   renderCUDA<3>  (backward)      backward.cu:437-595           -> render_global_bwd_kernel
//   renderkBufferCUDA<3,W,false>   stopthepop/resorted_render.cuh:17-221  -> render_kbuffer_kernel<W,false>
//   renderkBufferBackwardCUDA<3,W> stopthepop/resorted_render.cuh:223-471 -> render_kbuffer_kernel<W,true>
//   renderSortedFullCUDA<3,false>  stopthepop/resorted_render.cuh:474-675 -> render_full_fwd_kernel
//
// Layout: one 256-thread workgroup (4 wave64) per 16x16 tile; the tile's (tile,depth)-sorted list is
// staged 256 entries at a time into LDS with one coalesced id load + one gather per thread, then
// every lane walks the staged batch with broadcast LDS reads (all lanes look at entry j together).
// In GLOBAL backward all 64 lanes of a wave hold the SAME Gaussian at every step, so the nine
// gradient terms are reduced across the wave with DPP and leave as nine atomics per wave instead
// of 9 x 64.
#include <cstdlib>
#include <cstring>
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

namespace {

constexpr int BLOCK = 256;

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give every XCD
// a contiguous run of tiles -- neighbouring tiles share Gaussians, which then hit in that XCD's L2.
__device__ __forceinline__ int remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Sum over the 64 lanes of a wave; result valid in every lane (uniform).
__device__ __forceinline__ float wave_sum(float v)
{
    int t;
    t = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true); v += __int_as_float(t);  // quad_perm [1,0,3,2]
    t = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true); v += __int_as_float(t);  // quad_perm [2,3,0,1]
    t = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true); v += __int_as_float(t); // row_half_mirror
    t = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true); v += __int_as_float(t); // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

struct TileCtx {
    int tile, tx, ty, px, py;
    bool inside;
    uint2 range;
};

__device__ __forceinline__ TileCtx tile_ctx(const RenderArgs& a)
{
    TileCtx c;
    const int rows = a.ty1 - a.ty0;
    const int t = a.tile_order ? (int)a.tile_order[blockIdx.x] : remap_tile((int)blockIdx.x, a.gx * rows);
    c.tx = t % a.gx;
    c.ty = a.ty0 + t / a.gx;
    c.tile = c.ty * a.gx + c.tx;
    c.px = c.tx * TILE + (threadIdx.x & 15);
    c.py = c.ty * TILE + (threadIdx.x >> 4);
    c.inside = c.px < a.W && c.py < a.H;
    c.range = a.ranges[c.tile];
    return c;
}

// ------------------------------------------------------------------------------------------------
// GLOBAL forward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) render_global_fwd_kernel(const RenderArgs a)
{
    __shared__ float2 s_xy[BLOCK];
    __shared__ float4 s_co[BLOCK];
    __shared__ float s_col[3][BLOCK];
    __shared__ float s_dist[BLOCK]; // debug depth visualisation only: |cam - mean| of the staged entries

    const TileCtx c = tile_ctx(a);
    const float pxf = (float)c.px, pyf = (float)c.py;
    bool done = !c.inside;
    const int total = (int)(c.range.y - c.range.x);
    const int rounds = (total + BLOCK - 1) / BLOCK;
    int todo = total;

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, depth_acc = 0.0f;
    uint32_t contributor = 0, last_contributor = 0;

    for (int i = 0; i < rounds; i++, todo -= BLOCK) {
        if (__syncthreads_and(done)) break;
        const int progress = i * BLOCK + (int)threadIdx.x;
        if ((int)c.range.x + progress < (int)c.range.y) {
            const int id = (int)a.point_list[c.range.x + progress];
            s_xy[threadIdx.x] = a.means2D[id];
            s_co[threadIdx.x] = a.conic_opacity[id];
            s_col[0][threadIdx.x] = a.features[3 * (size_t)id + 0];
            s_col[1][threadIdx.x] = a.features[3 * (size_t)id + 1];
            s_col[2][threadIdx.x] = a.features[3 * (size_t)id + 2];
            if (a.debug_depth) { // reference forward.cu:337-341
                const float ddx = a.cam[0] - a.means3D[3 * (size_t)id], ddy = a.cam[1] - a.means3D[3 * (size_t)id + 1], ddz = a.cam[2] - a.means3D[3 * (size_t)id + 2];
                s_dist[threadIdx.x] = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            }
        }
        __syncthreads();
        const int n = min(BLOCK, todo);
        for (int j = 0; !done && j < n; j++) {
            contributor++;
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = blend_power(dx, dy, co);
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, co.w * exp_blend(power));
            if (alpha < ALPHA_THRESHOLD) continue;
            const float test_T = T * (1 - alpha);
            if (test_T < T_THRESHOLD) { done = true; continue; }
            C0 += s_col[0][j] * alpha * T;
            C1 += s_col[1][j] * alpha * T;
            C2 += s_col[2][j] * alpha * T;
            if (a.debug_depth) depth_acc += s_dist[j] * alpha * T;
            T = test_T;
            last_contributor = contributor;
        }
    }
    if (c.inside) {
        const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * c.py + c.px;
        a.final_T[pid] = T;
        a.n_contrib[pid] = last_contributor;
        if (a.debug_depth) { // reference outputDebugVis, stopthepop_common.cuh:297-301
            a.out_color[pid] = depth_acc;
            a.out_color[N + pid] = T;
        } else {
            a.out_color[pid] = C0 + T * a.bg[0];
            a.out_color[N + pid] = C1 + T * a.bg[1];
            a.out_color[2 * N + pid] = C2 + T * a.bg[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GLOBAL backward: back-to-front with the stored n_contrib / final_T
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) render_global_bwd_kernel(const RenderArgs a)
{
    __shared__ int s_id[BLOCK];
    __shared__ float2 s_xy[BLOCK];
    __shared__ float4 s_co[BLOCK];
    __shared__ float s_col[3][BLOCK];

    const TileCtx c = tile_ctx(a);
    const float pxf = (float)c.px, pyf = (float)c.py;
    const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * c.py + c.px;
    const int total = (int)(c.range.y - c.range.x);
    const int rounds = (total + BLOCK - 1) / BLOCK;
    int todo = total;

    const float T_final = c.inside ? a.final_T[pid] : 0.0f;
    float T = T_final;
    uint32_t contributor = (uint32_t)total;
    const uint32_t last_contributor = c.inside ? a.n_contrib[pid] : 0u;
    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, dL_dpixel[3] = {0, 0, 0};
    float last_alpha = 0.0f;
    if (c.inside)
        for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = a.dL_dpix[ch * N + pid];
    const float bg_dot = a.bg[0] * dL_dpixel[0] + a.bg[1] * dL_dpixel[1] + a.bg[2] * dL_dpixel[2];
    const float ddelx_dx = 0.5f * (float)a.W, ddely_dy = 0.5f * (float)a.H;
    const int lane = lane_id();

    for (int i = 0; i < rounds; i++, todo -= BLOCK) {
        __syncthreads();
        const int progress = i * BLOCK + (int)threadIdx.x;
        if ((int)c.range.x + progress < (int)c.range.y) {
            const int id = (int)a.point_list[c.range.y - progress - 1];
            s_id[threadIdx.x] = id;
            s_xy[threadIdx.x] = a.means2D[id];
            s_co[threadIdx.x] = a.conic_opacity[id];
            s_col[0][threadIdx.x] = a.features[3 * (size_t)id + 0];
            s_col[1][threadIdx.x] = a.features[3 * (size_t)id + 1];
            s_col[2][threadIdx.x] = a.features[3 * (size_t)id + 2];
        }
        __syncthreads();
        const int n = min(BLOCK, todo);
        for (int j = 0; j < n; j++) {
            // every lane steps through the same entry; lanes that skip contribute zeros to the reduction
            contributor--;
            bool use = c.inside && contributor < last_contributor;
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = blend_power(dx, dy, co);
            use = use && !(power > 0.0f);
            const float G = exp_blend(power);
            const float alpha = fminf(0.99f, co.w * G);
            use = use && !(alpha < ALPHA_THRESHOLD);
            if (!__any(use)) continue;

            float g_col[3] = {0, 0, 0}, g_mx = 0, g_my = 0, g_cxx = 0, g_cxy = 0, g_cyy = 0, g_op = 0;
            if (use) {
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float col = s_col[ch][j];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = col;
                    dL_dalpha += (col - accum_rec[ch]) * dL_dpixel[ch];
                    g_col[ch] = dchannel_dcolor * dL_dpixel[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_cxx = -0.5f * gdx * dx * dL_dG;
                g_cxy = -0.5f * gdx * dy * dL_dG;
                g_cyy = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            const float r0 = wave_sum(g_col[0]), r1 = wave_sum(g_col[1]), r2 = wave_sum(g_col[2]);
            const float r3 = wave_sum(g_mx), r4 = wave_sum(g_my), r5 = wave_sum(g_cxx), r6 = wave_sum(g_cxy), r7 = wave_sum(g_cyy);
            const float r8 = wave_sum(g_op);
            if (lane < 9) { // lane k hands over term k: nine lanes, one atomic instruction, one 64-byte record
                float v = r0;
                switch (lane) {
                case 1: v = r1; break;
                case 2: v = r2; break;
                case 3: v = r3; break;
                case 4: v = r4; break;
                case 5: v = r5; break;
                case 6: v = r6; break;
                case 7: v = r7; break;
                case 8: v = r8; break;
                default: break;
                }
                atomicAdd(grad_slot(a, s_id[j], lane), v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PPX_KBUFFER forward / backward: per-pixel sorted window keyed by depth along the pixel's own ray
// ------------------------------------------------------------------------------------------------
// (min_power_rect: stp_device.h)

// MODE 0 = forward, 1 = backward that re-runs the window sort (the reference's scheme; nine atomics per blended pair),
// 2 = training forward: additionally records every pixel's blend order in the blend log, so that the backward is the
// replay kernel of stp_render_replay.hip (the same log format and thread -> pixel mapping as the hierarchical mode).
constexpr int KB_FWD = 0, KB_BWD = 1, KB_FWD_RECORD = 2, KB_FWD_DEPTH = 3; // 3: forward of the debug depth visualisation

template <int WIN, int MODE>
__global__ void __launch_bounds__(BLOCK) render_kbuffer_kernel(const RenderArgs a)
{
    constexpr bool BACKWARD = MODE == KB_BWD;
    constexpr bool RECORD = MODE == KB_FWD_RECORD;
    constexpr bool DEPTHVIZ = MODE == KB_FWD_DEPTH;
    // one staging round = BLOCK entry records (A, B, C, D of BinningState), read with unit stride from the list-ordered
    // entry arrays; the forward passes carry the list position through the window, the backward pass the Gaussian id
    __shared__ float4 s_A[BLOCK];
    __shared__ float4 s_B[BLOCK];
    __shared__ float4 s_C[BLOCK];
    __shared__ float4 s_D[BLOCK];
    // Strip pre-test (ours; results unchanged): a wave's pixels are four rows of the tile, and most entries of a tile's
    // list cannot reach alpha >= 1/255 anywhere in a given 16x4 strip.  The thread that stages an entry computes the
    // EXACT minimum of the exponent's quadratic form over each of the four strips (min_power_rect below -- not the
    // reference's max-contribution estimate, stopthepop_common.cuh:130-174, which in the corner case returns an interior
    // point and is therefore not a bound) and leaves four bits; a wave skips the per-pixel evaluation of an entry whose
    // bit is clear (with a margin of 1e-3 on the threshold against rounding: a skipped entry fails the exact per-pixel
    // test for every pixel of the strip).  The pop-before-look step and the contributor count still run.
    __shared__ uint32_t s_hit[BLOCK];

    TileCtx c = tile_ctx(a);
    if constexpr (BACKWARD) {
        // fallback duty only when the forward recorded blend logs: just the tiles whose log overflowed
        if (a.flag_mode == 1 && a.tile_flags[c.tile] == 0u) return;
    }
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    if constexpr (RECORD) { // the replay kernel's pixel of (wave, lane): wave = row of four 4x4 sub-tiles, 2x2 quads inside
        const int sb = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
        c.px = c.tx * TILE + 4 * sb + 2 * (m & 1) + (q & 1);
        c.py = c.ty * TILE + 4 * w + 2 * (m >> 1) + (q >> 1);
        c.inside = c.px < a.W && c.py < a.H;
    }
    const float pxf = (float)c.px, pyf = (float)c.py;
    bool done = !c.inside;
    const int total = (int)(c.range.y - c.range.x);
    const int rounds = (total + BLOCK - 1) / BLOCK;
    int todo = total;

    const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    const float3 dir = view_ray(a.inv_vp, cam, pxf, pyf, a.W, a.H);

    Window<WIN> win; // payload: Gaussian id, or (recording forward) the entry's position in the tile list
    win.init();
    FwdPixel fp;
    BwdPixel bp;
    if constexpr (BACKWARD) init_bwd_pixel(bp, a, c.inside, c.px, c.py);
    else init_fwd_pixel(fp);
    uint32_t contributor = 0;
    float depth_acc = 0.0f;
    char* const log_base = RECORD ? log_wave_slice(a.blend_log, c.tile, w, a.log_depth) : nullptr; // [record / 8][lane][record % 8], stp_blend.h
    int nrec = 0;
    const float4* const eF = a.entF + c.range.x;
    const float4* const eCl = a.entC + c.range.x;
    const float4* const eDl = a.entD + c.range.x;

    auto blend_one = [&]() {
        if (win.num == 0) return;
        bool ok;
        if constexpr (BACKWARD) ok = blend_backward(bp, a, c.px, c.py, win.id[0], win.store[0]);
        else {
            // The forward windows do not carry alpha (two selects per slot and insertion less): it is evaluated again here,
            // from the same entry record with the same instructions, hence to the same bits as when the entry passed the
            // candidate test.
            const int pos = win.id[0];
            const float4 colr = eF[pos], eCp = eCl[pos], eDp = eDl[pos];
            const float col[3] = {colr.x, colr.y, colr.z};
            const float alpha0 = fminf(0.99f, eDp.w * exp_blend(blend_power(eCp.y - pxf, eCp.z - pyf, eDp)));
            const float T_before = fp.T;
            ok = blend_forward_c(fp, col, alpha0);
            if constexpr (DEPTHVIZ) { if (ok) depth_acc += win.depth[0] * alpha0 * T_before; } // reference resorted_render.cuh:107
            if constexpr (RECORD) {
                if (ok) {
                    if (nrec < a.log_depth) *reinterpret_cast<log_t*>(log_base + log_record_offset<true>(2u * (uint32_t)nrec, (uint32_t)lane << LOG_PIECE_SHIFT)) = (log_t)pos;
                    nrec++;
                }
            }
        }
        if (!ok) { win.num--; done = true; return; }
        win.pop();
    };

    for (int i = 0; i < rounds; i++, todo -= BLOCK) {
        if (__syncthreads_and(done)) break;
        const int progress = i * BLOCK + (int)threadIdx.x;
        if ((int)c.range.x + progress < (int)c.range.y) {
            const size_t gi = (size_t)c.range.x + progress;
            s_A[threadIdx.x] = a.entA[gi];
            s_B[threadIdx.x] = a.entB[gi];
            const float4 sc = a.entC[gi], sd = a.entD[gi];
            s_C[threadIdx.x] = sc;
            s_D[threadIdx.x] = sd;
            uint32_t hit = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ww++) {
                const float x0 = (float)(c.tx * TILE), y0 = (float)(c.ty * TILE + 4 * ww);
                const float p = min_power_rect(sd, x0 - sc.y, x0 + 15.0f - sc.y, y0 - sc.z, y0 + 3.0f - sc.z);
                hit |= (sd.w * exp_blend(-p) < ALPHA_THRESHOLD * 0.999f) ? 0u : (1u << ww);
            }
            s_hit[threadIdx.x] = hit;
        }
        __syncthreads();
        const int n = min(BLOCK, todo);
        for (int j = 0; !done && j < n; j++) {
            if (win.num == WIN) blend_one(); // before the next candidate is looked at
            if (done) break;
            contributor++;
            if (((s_hit[j] >> w) & 1u) == 0u) continue; // (the same for every lane of the wave)
            const float4 eCj = s_C[j];
            const float4 co = s_D[j];
            const float dx = eCj.y - pxf, dy = eCj.z - pyf;
            const float power = blend_power(dx, dy, co);
            if (power > 0.0f) continue;
            const float G = exp_blend(power);
            const float alpha = fminf(0.99f, co.w * G);
            if (alpha < ALPHA_THRESHOLD) continue;
            const float depth = depth_along_ray_ent(s_A[j], s_B[j], eCj, dir);
            if (depth < 0.0f) continue;
            win.insert(depth, BACKWARD ? __float_as_int(eCj.w) : i * BLOCK + j, BACKWARD ? G : 0.0f);
        }
    }
    if (!done)
        while (win.num > 0 && !done) blend_one();

    if constexpr (!BACKWARD) {
        if (c.inside) {
            const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * c.py + c.px;
            a.final_T[pid] = fp.T;
            a.n_contrib[pid] = RECORD ? (uint32_t)nrec : contributor; // (recording forward: the pixel's number of log records)
            if constexpr (DEPTHVIZ) {
                a.out_color[pid] = depth_acc;
                a.out_color[N + pid] = fp.T;
            } else {
                a.out_color[pid] = fp.C[0] + fp.T * a.bg[0];
                a.out_color[N + pid] = fp.C[1] + fp.T * a.bg[1];
                a.out_color[2 * N + pid] = fp.C[2] + fp.T * a.bg[2];
            }
        }
        if constexpr (RECORD) {
            if (nrec > a.log_depth || total > LOG_MAX_LIST) a.tile_flags[c.tile] = 1u; // log overflow: this tile's backward re-sorts
            report_log_need(a.log_need, nrec, a.log_tag);
        }
    }
}

} // namespace

static RenderArgs make_args(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img)
{
    RenderArgs a{};
    a.W = f.W; a.H = f.H; a.gx = f.gx; a.ty0 = f.ty0; a.ty1 = f.ty1;
    a.ranges = img.ranges; a.point_list = b.point_list; a.means2D = g.means2D; a.conic_opacity = g.conic_opacity;
    a.cov3D_inv = g.cov3D_inv; a.features = f.colors_precomp ? f.colors_precomp : g.rgb; // reference rasterizer_impl.cu:367,473
    a.inv_vp = f.inv_viewprojmatrix; a.cam = f.cam_pos; a.bg = f.background;
    a.entA = b.entA; a.entB = b.entB; a.entC = b.entC; a.entD = b.entD; a.entF = b.entF;
    a.final_T = img.final_T; a.n_contrib = img.n_contrib;
    a.blend_log = img.blend_log; a.tile_flags = img.tile_flags; a.flag_mode = 0;
    static const bool counters = [] { const char* e = std::getenv("STP_SORT"); return e && std::strcmp(e, "counters") == 0; }(); // (that path uses tile_cursor itself)
    a.tile_order = (tile_order_used(f) && !counters && !f.split_launch) ? img.tile_cursor + f.gx * f.ty0 : nullptr;
    a.log_depth = img.log_depth; a.log_need = f.log_need; a.log_tag = f.log_tag;
    a.debug_depth = f.s.debug_visualization == STP_DEBUG_DEPTH ? 1 : 0; a.means3D = f.means3D;
    return a;
}

// implemented in stp_render_hier_fwd.hip / stp_render_hier_bwd.hip / stp_render_full.hip
hipError_t launch_hier_fwd(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err);
hipError_t launch_hier_bwd(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err);
hipError_t launch_hier_rec(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err);
hipError_t launch_hier_dbg(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err);
hipError_t launch_depth_colormap(float* out_color, int N, uint32_t* minmax, hipStream_t st); // stp_debug_viz.hip
hipError_t launch_hier_replay(const FrameParams& f, const RenderArgs& a, hipStream_t st);
hipError_t launch_full_fwd(const FrameParams& f, const RenderArgs& a, hipStream_t st);
hipError_t launch_kbuffer_wave(int mode, const FrameParams& f, const RenderArgs& a, hipStream_t st, bool* handled); // stp_render_kbuf.hip

template <int MODE> static hipError_t launch_kbuffer(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    const dim3 grid(f.gx * (f.ty1 - f.ty0)), block(BLOCK);
    const int w = f.s.queue_per_pixel; // reference forward.cu:409-425 / backward.cu:712-731
#define STP_KB(WIN) hipLaunchKernelGGL((render_kbuffer_kernel<WIN, MODE>), grid, block, 0, st, a)
    if (w <= 1) STP_KB(1);
    else if (w <= 2) STP_KB(2);
    else if (w <= 4) STP_KB(4);
    else if (w <= 8) STP_KB(8);
    else if (w <= 12) STP_KB(12);
    else if (w <= 16) STP_KB(16);
    else if (w <= 20) STP_KB(20);
    else STP_KB(24);
#undef STP_KB
    return hipGetLastError();
}

hipError_t launch_render_forward(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img,
                                 float* out_color, hipStream_t st, std::string* err)
{
    RenderArgs a = make_args(f, g, b, img);
    a.out_color = out_color;
    const dim3 grid(f.gx * (f.ty1 - f.ty0)), block(BLOCK);
    if (grid.x == 0) return hipSuccess;
    switch (f.s.sort_mode) {
    case MODE_GLOBAL:
        hipLaunchKernelGGL(render_global_fwd_kernel, grid, block, 0, st, a);
        return hipGetLastError();
    case MODE_KBUFFER: {
        // the wave64 kernel of stp_render_kbuf.hip (STP_KBUFFER=tile keeps the one-entry-per-wave kernel of this file for
        // comparison; the results are the same)
        static const char* const kb_env = std::getenv("STP_KBUFFER");
        static const bool kb_tile = kb_env && std::strcmp(kb_env, "tile") == 0;
        if (!kb_tile) {
            bool handled = false;
            const hipError_t e = launch_kbuffer_wave(a.debug_depth ? KB_FWD_DEPTH : uses_blend_log(f.s) ? KB_FWD_RECORD : KB_FWD, f, a, st, &handled);
            if (handled) return e;
        }
        if (a.debug_depth) return launch_kbuffer<KB_FWD_DEPTH>(f, a, st);
        return uses_blend_log(f.s) ? launch_kbuffer<KB_FWD_RECORD>(f, a, st) : launch_kbuffer<KB_FWD>(f, a, st);
    }
    case MODE_FULL: return launch_full_fwd(f, a, st);
    case MODE_HIER:
        if (a.debug_depth) return launch_hier_dbg(f, a, st, err);
        return uses_blend_log(f.s) ? launch_hier_rec(f, a, st, err) : launch_hier_fwd(f, a, st, err);
    default: if (err) *err = "invalid sort mode"; return hipErrorInvalidValue;
    }
}

// The forward of the debug depth visualisation: the render kernels above leave sum(depth * alpha * T) in channel 0 and T
// in channel 1; the frame's extrema and the colormap finish the image (reference applyDebugVisualization,
// rasterizer_impl.cu:54-109).
hipError_t launch_render_debug_finish(const FrameParams& f, const ImageState& img, float* out_color, hipStream_t st)
{
    if (f.s.debug_visualization != STP_DEBUG_DEPTH) return hipSuccess;
    return launch_depth_colormap(out_color, f.W * f.H, img.dbg_minmax, st);
}

hipError_t launch_render_backward(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img,
                                  const BackwardParams& bw, hipStream_t st, std::string* err)
{
    RenderArgs a = make_args(f, g, b, img);
    a.pixel_colors = bw.pixel_colors; a.dL_dpix = bw.dL_dpix; a.grad_rec = bw.grad_rec; a.grad_stride = bw.grad_stride;
    const dim3 grid(f.gx * (f.ty1 - f.ty0)), block(BLOCK);
    if (grid.x == 0) return hipSuccess;
    switch (f.s.sort_mode) {
    case MODE_GLOBAL:
        hipLaunchKernelGGL(render_global_bwd_kernel, grid, block, 0, st, a);
        return hipGetLastError();
    case MODE_KBUFFER:
        if (uses_blend_log(f.s)) { // replay the forward's blend log; the re-sorting kernel only takes overflowed tiles
            hipError_t e = launch_hier_replay(f, a, st);
            if (e != hipSuccess) return e;
            a.flag_mode = 1;
        }
        return launch_kbuffer<KB_BWD>(f, a, st);
    case MODE_HIER:
        if (uses_blend_log(f.s)) { // replay the forward's blend log; the resorting kernel only takes overflowed tiles
            hipError_t e = launch_hier_replay(f, a, st);
            if (e != hipSuccess) return e;
            a.flag_mode = 1;
        }
        return launch_hier_bwd(f, a, st, err);
    default: if (err) *err = "Backward not supported for full per-pixel sort"; return hipErrorInvalidValue;
    }
}

} // namespace stp
