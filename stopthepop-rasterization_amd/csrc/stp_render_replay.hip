// stp_render_replay.hip -- backward of the hierarchical mode by REPLAYING the forward's blend log.
//
// No counterpart in the reference: its backward (hierarchical_render.cuh:1038-1175) re-runs the complete
// three-level resort to rediscover the order in which every pixel blended its Gaussians.  MI355X has 288 GB of
// HBM, so the training forward (render_hier_kernel<..., MODE_FWD_RECORD>) simply writes that order down --
// 4 bytes (the tile-list position) per blended (pixel, Gaussian) pair, BLEND_LOG_DEPTH = 256 records per pixel,
// 1 KiB per pixel, 2.1 GB at 1080p -- and this kernel walks each pixel's log front to back.  The gradient
// maths per pair is the reference's (blend_backward_terms); the result is the same sum in a different order.
// Tiles whose log overflowed (a pixel with more than 256 blended entries) are flagged by the forward and left
// to the resorting backward kernel, which then runs only on those tiles.
//
// Layout: one 256-thread workgroup per tile, thread -> pixel mapping identical to the forward (wave = row of four
// 4x4 sub-tiles), log laid out [tile][wave][k][lane] so that the 64 lanes of a wave read record k with one
// coalesced 256-byte load.  No sorting state: the kernel is a straight loop over k with a one-deep prefetch of
// the next record, the per-Gaussian data gathered from L2, and the nine gradient terms summed in a per-wave
// direct-mapped LDS cache of 64-bit fixed-point sums (see stp_render_hier.inc for why not fp32 LDS atomics):
// slot = list position mod 128, tagged; the sums of a slot go to memory (nine global atomics) only when another
// list position claims the slot, or at the end.
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

namespace {

constexpr int RW = 128; // cache slots per wave

__device__ __forceinline__ int replay_remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__global__ void __launch_bounds__(256, 3) render_hier_replay_kernel(const RenderArgs a)
{
    __shared__ unsigned long long s_acc[4][9 * RW];
    __shared__ int s_tag[4][RW];
    __shared__ int s_claim[4][RW];

    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = replay_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    if (a.tile_flags[tile] != 0u) return; // log overflow: the resorting backward takes this tile
    const uint2 range = a.ranges[tile];
    const int px = tile_x * TILE + 4 * s + 2 * (m & 1) + (q & 1), py = tile_y * TILE + 4 * w + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;

    unsigned long long* const acc = s_acc[w];
    int* const tag = s_tag[w];
    int* const claim = s_claim[w];
    for (int i = lane; i < RW; i += 64) {
        tag[i] = -1;
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k * RW + i] = 0ull;
    }

    BwdPixel bp;
    init_bwd_pixel(bp, a, inside, px, py);
    int n = inside ? (int)a.n_contrib[(size_t)a.W * py + px] : 0;
    int nmax = n;
    float md = fmaxf(fmaxf(fabsf(bp.dL_dpix[0]), fabsf(bp.dL_dpix[1])), fabsf(bp.dL_dpix[2]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nmax = max(nmax, __shfl_xor(nmax, off));
        md = fmaxf(md, __shfl_xor(md, off));
    }
    // fixed-point scale of this wave's sums (stp_render_hier.inc: "on-chip gradient window")
    int md_exp = 0;
    if (md > 0.0f && md < 3.0e38f) (void)frexpf(md, &md_exp);
    const double fx_scale = ldexp(1.0, 31 - md_exp), fx_inv = ldexp(1.0, md_exp - 31);
    const float fx_cap = ldexpf(1.0f, min(md_exp + 20, 126));
    wave_sync();

    const uint32_t* const log_base = a.blend_log + ((size_t)(tile * 4 + w) * BLEND_LOG_DEPTH) * 64 + lane;
    const float pxf = (float)px, pyf = (float)py;

    auto evict = [&](int slot, int old_pos) __attribute__((always_inline)) {
        const int old_id = (int)a.point_list[range.x + old_pos];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const long long v = (long long)acc[k * RW + slot];
            acc[k * RW + slot] = 0ull;
            if (v != 0) atomicAdd(grad_slot(a, old_id, k), (float)((double)v * fx_inv));
        }
    };

    // one-deep prefetch: (pos, id) of record k are in registers when iteration k starts
    int pos = (0 < n) ? (int)log_base[0] : -1;
    int id = (pos >= 0) ? (int)a.point_list[range.x + pos] : 0;
    for (int k = 0; k < nmax; k++) {
        const bool have = k < n;
        FrontData fd{};
        if (have) fd = load_front(a, id);
        const int cur_pos = pos, cur_id = id;
        if (k + 1 < n) {
            pos = (int)log_base[(size_t)(k + 1) * 64];
            id = (int)a.point_list[range.x + pos];
        }
        float g[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool ok = false;
        if (have) {
            const float dx = fd.xy.x - pxf, dy = fd.xy.y - pyf;
            const float power = -0.5f * (fd.co.x * dx * dx + fd.co.z * dy * dy) - fd.co.y * dx * dy;
            const float G = expf(power);
            ok = blend_backward_terms(bp, a, px, py, fd, G, g);
            if (!ok) n = k; // (an ulp of difference against the forward's transmittance: stop where it says so)
        }
        // ---- accumulate (converged code: every lane of the wave is here) ----
        // Neighbouring pixels blend the same entry at the same step more often than not: sum the terms of a
        // 2x2 quad (and then of a whole 4x4 sub-tile) with DPP when all its lanes hold the same list position,
        // so that one lane goes to LDS instead of 4 (16) lanes hitting the same address.
        const int key = ok ? cur_pos : -2 - lane; // unique when not blending
        bool writer = ok;
        {
            const int k0 = quad_bcast_i<0>(key), k1 = quad_bcast_i<1>(key), k2 = quad_bcast_i<2>(key), k3 = quad_bcast_i<3>(key);
            const bool quad_same = (k0 == k1) && (k0 == k2) && (k0 == k3);
            if (quad_same) {
#pragma unroll
                for (int kk = 0; kk < 9; kk++) {
                    g[kk] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g[kk]), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
                    g[kk] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g[kk]), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
                }
                writer = writer && q == 0;
            }
            // whole sub-tile (16-lane row) on one position: its four quad sums are combined with two more DPP steps
            const unsigned long long qs = __ballot(quad_same);
            const int r0 = __builtin_amdgcn_mov_dpp(key, 0x140, 0xF, 0xF, true);  // row_mirror: lane i <- 15 - i
            const int r1 = __builtin_amdgcn_mov_dpp(key, 0x141, 0xF, 0xF, true);  // row_half_mirror: lane i <- 7 - i (per half)
            const bool row_quads = ((qs >> (lane & ~15)) & 0xFFFFull) == 0xFFFFull;
            const unsigned long long rm = __ballot(row_quads && key == r0 && key == r1);
            if (((rm >> (lane & ~15)) & 0xFFFFull) == 0xFFFFull) {
#pragma unroll
                for (int kk = 0; kk < 9; kk++) {
                    g[kk] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g[kk]), 0x141, 0xF, 0xF, true));
                    g[kk] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g[kk]), 0x140, 0xF, 0xF, true));
                }
                writer = writer && x == 0;
            }
        }
        const int slot = cur_pos & (RW - 1);
        const int owner = writer ? tag[slot] : cur_pos;
        const bool miss = writer && owner != cur_pos;
        if (miss) claim[slot] = lane; // several lanes may want the slot: one wins
        wave_sync();
        if (miss && claim[slot] == lane) {
            if (owner >= 0) evict(slot, owner);
            tag[slot] = cur_pos;
        }
        wave_sync();
        if (writer) {
            float gmax = fabsf(g[0]);
#pragma unroll
            for (int kk = 1; kk < 9; kk++) gmax = fmaxf(gmax, fabsf(g[kk]));
            if (tag[slot] == cur_pos && gmax < fx_cap) {
#pragma unroll
                for (int kk = 0; kk < 9; kk++) {
                    const double tq = fma((double)g[kk], fx_scale, 6755399441055744.0);
                    const long long qv = __double_as_longlong(tq) - 0x4338000000000000ll;
                    atomicAdd(&acc[kk * RW + slot], (unsigned long long)qv);
                }
            } else { // lost the slot to another position in this very step, or a term too large for the fixed point
#pragma unroll
                for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk]);
            }
        }
    }
    wave_sync();
    for (int slot = lane; slot < RW; slot += 64) {
        const int owner = tag[slot];
        if (owner >= 0) evict(slot, owner);
    }
}

} // namespace

hipError_t launch_hier_replay(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(render_hier_replay_kernel, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
