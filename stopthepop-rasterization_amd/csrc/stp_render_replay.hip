// stp_render_replay.hip -- backward of the per-pixel-sort modes by REPLAYING the forward's blend log.
//
// No counterpart in the reference: its backward (hierarchical_render.cuh:1038-1175) re-runs the complete
// three-level resort to rediscover the order in which every pixel blended its Gaussians.  MI355X has 288 GB of
// HBM, so the training forward (render_hier_kernel<..., MODE_FWD_RECORD>) simply writes that order down --
// 2 bytes (the tile-list position) per blended (pixel, Gaussian) pair, BLEND_LOG_DEPTH = 256 records per pixel,
// 512 B per pixel, 1.07 GB at 1080p -- and this kernel walks each pixel's log front to back.  The gradient
// maths per pair is the reference's (blend_backward_terms); the result is the same sum in a different order.
// Tiles whose log overflowed (a pixel with more than 256 blended entries) are flagged by the forward and left
// to the resorting backward kernel, which then runs only on those tiles.
//
// Layout: one 256-thread workgroup per tile, thread -> pixel mapping identical to the forward (wave = row of four
// 4x4 sub-tiles), log laid out [tile][wave][k][lane] so that the 64 lanes of a wave read record k with one
// coalesced 128-byte load.  No sorting state: the kernel is a straight loop over k with the next record prefetched,
// the entry's data read from the list-ordered entry records, lanes on the same list position merged with DPP, and
// the nine gradient terms summed on chip as 64-bit fixed point (see stp_render_hier.inc for why not fp32 LDS
// atomics): tiles with at most 512 list entries keep one set of sums per POSITION for the whole workgroup, longer
// lists go through a per-wave direct-mapped cache (slot = position mod 112, tagged) whose slots are written back
// -- nine lanes, one atomic instruction into the Gaussian's 64-byte gradient record -- when another position claims
// them, or at the end.  The k-buffer mode records the same log and uses this kernel as its backward too.
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

#ifdef STP_REPLAY_STATS
__device__ unsigned long long g_replay_stats[16];
#endif

namespace {

#ifndef STP_REPLAY_PAIRMERGE
#define STP_REPLAY_PAIRMERGE 3 // merge levels: 1 = inside 2x2 quads (lane^1, lane^2), 2 = + mirror in the 8-lane half, 3 = + mirror in the 16-lane row
                               // (C2-full: 1.38 ms without, 0.99 / 0.95 / 0.94 ms with 1 / 2 / 3)
#endif

#ifndef STP_REPLAY_RW
#define STP_REPLAY_RW 112 // 112 slots x 72 B x 4 waves + tags = 38 KB: four workgroups per CU (128 slots: three); measured best on C2
#endif
#ifndef STP_REPLAY_OCC
#define STP_REPLAY_OCC 4
#endif
constexpr int RW = STP_REPLAY_RW; // cache slots per wave
// Tiles whose list is at most DIRECT_CAP entries long need no cache at all: the workgroup's LDS holds one set of sums
// for EVERY list position (9 x 512 x 8 B = 36 KB, the same footprint as the four per-wave caches), a blend adds to
// acc[term][position] directly -- no tags, no claim protocol, no evictions -- and the sums leave the chip once, at
// the end.  C2 (321 entries per tile on average, 406 at most) runs entirely on this path.
constexpr int DIRECT_CAP = 512;
constexpr int LDS_WORDS64 = (4 * 9 * RW * 8 + 3 * 4 * RW * 4 + 7) / 8 > 9 * DIRECT_CAP ? (4 * 9 * RW * 8 + 3 * 4 * RW * 4 + 7) / 8 : 9 * DIRECT_CAP;

__device__ __forceinline__ int replay_remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// RETRIES: extra claim rounds for lanes that lose a slot to another position of the same step (see below)
template <int RETRIES>
__global__ void __launch_bounds__(256, STP_REPLAY_OCC) render_hier_replay_kernel(const RenderArgs a)
{
    __shared__ unsigned long long s_raw[LDS_WORDS64]; // cached path: acc[4][9*RW], tag/claim/gid[4][RW]; direct path: acc[9][DIRECT_CAP]
    __shared__ float s_md[4];

    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = replay_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    if (a.tile_flags[tile] != 0u) return; // log overflow: the resorting backward takes this tile
    const uint2 range = a.ranges[tile];
    const int px = tile_x * TILE + 4 * s + 2 * (m & 1) + (q & 1), py = tile_y * TILE + 4 * w + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;

    const int list_len = (int)(range.y - range.x);
    const bool direct = list_len <= DIRECT_CAP; // workgroup-uniform
    unsigned long long* const acc = s_raw + w * 9 * RW;
    int* const tag = reinterpret_cast<int*>(s_raw + 4 * 9 * RW) + w * RW;
    int* const claim = tag + 4 * RW;
    int* const gid = tag + 8 * RW; // Gaussian id of the slot's owner
    unsigned long long* const dacc = s_raw; // direct path: [term][position]
    if (direct) {
        for (int i = (int)threadIdx.x; i < 9 * DIRECT_CAP; i += 256) dacc[i] = 0ull;
    } else {
        for (int i = lane; i < RW; i += 64) {
            tag[i] = -1;
#pragma unroll
            for (int k = 0; k < 9; k++) acc[k * RW + i] = 0ull;
        }
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
    // fixed-point scale of the sums (stp_render_hier.inc: "on-chip gradient window"): one for the workgroup, because
    // the direct path shares its accumulators between the four waves
    if (lane == 0) s_md[w] = md;
    __syncthreads(); // (also: the accumulators are zeroed)
    md = fmaxf(fmaxf(s_md[0], s_md[1]), fmaxf(s_md[2], s_md[3]));
    int md_exp = 0;
    if (md > 0.0f && md < 3.0e38f) (void)frexpf(md, &md_exp);
    const double fx_scale = ldexp(1.0, 31 - md_exp), fx_inv = ldexp(1.0, md_exp - 31);
    const float fx_cap = ldexpf(1.0f, min(md_exp + 20, 126));
    wave_sync();

    const log_t* const log_base = reinterpret_cast<const log_t*>(a.blend_log) + ((size_t)(tile * 4 + w) * BLEND_LOG_DEPTH) * 64 + lane;
    const float pxf = (float)px, pyf = (float)py;

    // Hand the sums of up to four slots to memory with ONE atomic instruction: the 16-lane group g takes the
    // slot its source lane names, lanes 0..8 of the group one of the nine sums each, all going to the same
    // 64-byte gradient record (one request to the memory pipeline; tools/global_atomic_bench.hip: 9x the rate of
    // nine single-lane atomics).  `mask` = lanes holding a slot to evict in `slot_v`; wave-uniform control flow.
    const int grp = lane >> 4, term = lane & 15;
    auto evict_lanes = [&](unsigned long long mask, int slot_v) __attribute__((always_inline)) {
        while (mask != 0ull) {
            // source lane of my group: the grp-th set bit of the mask (scalar bit tricks, then one select chain)
            const int s0 = __builtin_ctzll(mask);
            unsigned long long m1 = mask & (mask - 1);
            const int s1 = m1 ? __builtin_ctzll(m1) : -1;
            unsigned long long m2 = m1 & (m1 - 1);
            const int s2 = (m1 && m2) ? __builtin_ctzll(m2) : -1;
            unsigned long long m3 = m2 & (m2 - 1);
            const int s3 = (m1 && m2 && m3) ? __builtin_ctzll(m3) : -1;
            mask = (m1 && m2 && m3) ? (m3 & (m3 - 1)) : 0ull;
            const int src = grp == 0 ? s0 : grp == 1 ? s1 : grp == 2 ? s2 : s3;
            const int slot = __shfl(slot_v, src < 0 ? 0 : src);
            if (src >= 0 && term < 9) {
                const long long v = (long long)acc[term * RW + slot];
                if (v != 0) {
                    acc[term * RW + slot] = 0ull;
                    atomicAdd(grad_slot(a, gid[slot], term), (float)((double)v * fx_inv));
                }
            }
        }
    };

    // Two dependent loads lead to a blend: log record (list position) -> the entry's record in the list-ordered entry
    // arrays (mean, Gaussian id, conic/opacity, colour: BinningState::entC/entD/entF).  They are software pipelined one
    // step apart: when iteration k starts, the data of record k and the position of record k+1 are in registers (or in
    // flight since the previous iteration).
    const float4* const eC = a.entC + range.x;
    const float4* const eD = a.entD + range.x;
    const float4* const eF = a.entF + range.x;
    auto log_at = [&](int k) __attribute__((always_inline)) { return (k < n) ? (int)log_base[(size_t)k * 64] : -1; };
    struct Entry { float4 c, d, f; };
    auto entry_at = [&](int p) __attribute__((always_inline)) { // (position 0 where there is no record: a harmless read, no branch)
        const int q = max(p, 0);
        return Entry{eC[q], eD[q], eF[q]};
    };
    int pos = log_at(0), pos1 = log_at(1);
    Entry en = entry_at(pos);
    for (int k = 0; k < nmax; k++) {
        const bool have = k < n;
        const Entry cur = en;
        const int cur_pos = pos, cur_id = __float_as_int(cur.c.w);
        // issue the next round of loads before touching this step's data
        en = entry_at(pos1);
        const int pos2 = (k + 2 < n) ? (int)log_base[(size_t)(k + 2) * 64] : -1;
        pos = pos1;
        pos1 = pos2;
        FrontData cur_fd;
        cur_fd.co = cur.d;
        cur_fd.xy = make_float2(cur.c.y, cur.c.z);
        cur_fd.c[0] = cur.f.x; cur_fd.c[1] = cur.f.y; cur_fd.c[2] = cur.f.z;
        float g[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool ok = false;
        if (have) {
            const float dx = cur_fd.xy.x - pxf, dy = cur_fd.xy.y - pyf;
            const float power = -0.5f * (cur_fd.co.x * dx * dx + cur_fd.co.z * dy * dy) - cur_fd.co.y * dx * dy;
            const float G = exp_blend(power);
            ok = blend_backward_terms(bp, a, px, py, cur_fd, G, g);
            if (!ok) n = k; // (an ulp of difference against the forward's transmittance: stop where it says so)
        }
#if STP_REPLAY_PAIRMERGE
        // Pairwise merge (DPP): a lane and its partner -- lane^1, lane^2, then the mirror lanes of its 8-lane half and row -- that
        // hold the same list position sum their terms in registers and only one of them goes to LDS.  Per step 55 lanes
        // blend on 17.5 distinct positions (C2); the LDS atomics are the limiter and serialise on equal addresses, the
        // VALU has headroom: C2-full 1.38 -> 0.94 ms.  The partner's value enters as the DPP operand of one v_fmac per term.  (A pre-reduction that needs a whole quad on one position fires
        // for one quad in ten and does not pay.)
        {
            int key = ok ? cur_pos : -2 - lane; // unique when not blending
#define STP_MERGE_LEVEL(CTRL, LOWER)                                                                                    \
            {                                                                                                       \
                const int pk = __builtin_amdgcn_mov_dpp(key, CTRL, 0xF, 0xF, true);                                 \
                const bool match = pk == key;                                                                       \
                const float mf = (match && (LOWER)) ? 1.0f : 0.0f;                                                  \
                dpp_hazard_guard(); /* g[] was written by ordinary VALU instructions a moment ago */                \
                _Pragma("unroll") for (int kk = 0; kk < 9; kk++) g[kk] = partner_fma<CTRL>(g[kk], mf, g[kk]);       \
                if (match && !(LOWER)) { ok = false; key = -2 - lane; }                                             \
            }
            STP_MERGE_LEVEL(0xB1, (q & 1) == 0) // partner lane ^ 1 (quad_perm [1,0,3,2])
            STP_MERGE_LEVEL(0x4E, (q & 2) == 0) // partner lane ^ 2 (quad_perm [2,3,0,1])
#if STP_REPLAY_PAIRMERGE >= 2
            STP_MERGE_LEVEL(0x141, (x & 7) < 4)  // partner 7 - i inside each 8-lane half (row_half_mirror)
#endif
#if STP_REPLAY_PAIRMERGE >= 3
            STP_MERGE_LEVEL(0x140, x < 8)        // partner 15 - i inside the 16-lane row (row_mirror)
#endif
#undef STP_MERGE_LEVEL
        }
#endif
        if (direct) { // every list position has its own sums in LDS: nine adds, nothing else
            if (ok) {
                float gmax = fabsf(g[0]);
#pragma unroll
                for (int kk = 1; kk < 9; kk++) gmax = fmaxf(gmax, fabsf(g[kk]));
                if (gmax < fx_cap) {
#pragma unroll
                    for (int kk = 0; kk < 9; kk++) {
                        const double tq = fma((double)g[kk], fx_scale, 6755399441055744.0);
                        const long long qv = __double_as_longlong(tq) - 0x4338000000000000ll;
                        atomicAdd(&dacc[kk * DIRECT_CAP + cur_pos], (unsigned long long)qv);
                    }
                } else { // a term too large for the fixed point (never seen): straight to memory
#pragma unroll
                    for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk]);
                }
            }
            continue;
        }
        // ---- accumulate through the per-wave cache (converged code: every lane of the wave is here) ----
        const int key = ok ? cur_pos : -2 - lane; // unique when not blending
        bool writer = ok;
#ifdef STP_REPLAY_STATS
        {   // per wave-step: blending lanes, writer lanes after the pre-reductions, distinct positions among writers / blenders
            int first_w = writer ? 1 : 0, first_b = ok ? 1 : 0, first_q = ok ? 1 : 0, first_r = ok ? 1 : 0;
            for (int j = 0; j < 64; j++) {
                const int kj = __shfl(key, j);
                const int wj = __shfl((int)writer, j);
                const int oj = __shfl((int)ok, j);
                if (j < lane && kj == key) {
                    if (wj) first_w = 0;
                    if (oj) first_b = 0;
                    if (oj && (j >> 2) == (lane >> 2)) first_q = 0;
                    if (oj && (j >> 4) == (lane >> 4)) first_r = 0;
                }
            }
            const int dq = __popcll(__ballot(ok && first_q)), dr = __popcll(__ballot(ok && first_r));
            if (lane == 0) { atomicAdd(&g_replay_stats[5], (unsigned long long)dq); atomicAdd(&g_replay_stats[6], (unsigned long long)dr); }
            const int nb = __popcll(__ballot(ok)), nw = __popcll(__ballot(writer));
            const int dw = __popcll(__ballot(writer && first_w)), db = __popcll(__ballot(ok && first_b));
            if (lane == 0) {
                atomicAdd(&g_replay_stats[0], 1ull); atomicAdd(&g_replay_stats[1], (unsigned long long)nb);
                atomicAdd(&g_replay_stats[2], (unsigned long long)nw); atomicAdd(&g_replay_stats[3], (unsigned long long)dw);
                atomicAdd(&g_replay_stats[4], (unsigned long long)db);
            }
        }
#endif
        const int slot = (RW & (RW - 1)) == 0 ? (cur_pos & (RW - 1)) : (int)((unsigned)cur_pos % (unsigned)RW); // (cur_pos >= 0 for every lane that uses it)
        // Every add goes through a cache slot.  Lanes whose slot belongs to another position claim it (one winner per
        // slot), the winner hands the old sums to memory and takes the slot over.
        auto add_fixed = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 9; kk++) {
                const double tq = fma((double)g[kk], fx_scale, 6755399441055744.0);
                const long long qv = __double_as_longlong(tq) - 0x4338000000000000ll;
                atomicAdd(&acc[kk * RW + slot], (unsigned long long)qv);
            }
        };
        auto claim_round = [&](bool pending) __attribute__((always_inline)) {
            const int owner = pending ? tag[slot] : cur_pos;
            const bool miss = pending && owner != cur_pos;
            if (miss) claim[slot] = lane; // several lanes may want the slot: one wins
            wave_sync();
            const bool won = miss && claim[slot] == lane;
#ifdef STP_REPLAY_STATS
            { const int ne = __popcll(__ballot(won && owner >= 0)), nm = __popcll(__ballot(miss)); if (lane == 0) { atomicAdd(&g_replay_stats[7], (unsigned long long)ne); atomicAdd(&g_replay_stats[8], (unsigned long long)nm); atomicAdd(&g_replay_stats[9], 1ull); } }
#endif
            evict_lanes(__ballot(won && owner >= 0), slot);
            wave_sync();
            if (won) {
                tag[slot] = cur_pos;
                gid[slot] = cur_id;
            }
            wave_sync();
        };
        claim_round(writer);
        bool lost = false;
        if (writer) {
            float gmax = fabsf(g[0]);
#pragma unroll
            for (int kk = 1; kk < 9; kk++) gmax = fmaxf(gmax, fabsf(g[kk]));
            if (tag[slot] == cur_pos && gmax < fx_cap) add_fixed();
            else if (!(gmax < fx_cap)) { // a term too large for the fixed point (never seen): straight to memory
#pragma unroll
                for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk]);
            } else lost = true; // the slot went to another position of this very step
        }
        // Lanes that lost go round again and evict the winner in turn -- an eviction is one nine-lane atomic per four
        // slots, far cheaper than nine single-lane atomics.  Rare when neighbouring pixels walk the list together (C2:
        // one step in a hundred), the rule when every pixel has its own order over a long list (k-buffer, C3).
        // Compile-time choice: the hierarchical mode runs without retries (on C2 they cost 4-6 % in code quality and
        // one step in a hundred would use them), the k-buffer mode with three (straight-line, not a loop: with a back
        // edge here the compiler waits for the step's prefetch loads before the loop header).
#pragma unroll
        for (int retry = 0; retry < RETRIES; retry++) {
            if (__builtin_expect(__any(lost), 0)) {
                wave_sync(); // the adds above are issued before their slot can be evicted
                claim_round(lost);
                if (lost && tag[slot] == cur_pos) { add_fixed(); lost = false; }
            }
        }
        if (lost) { // still contested: nine single-lane atomics
#pragma unroll
            for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk]);
        }
    }
    if (direct) { // the sums of every position leave the chip once: 16-lane group = one position, nine lanes = its nine sums
        __syncthreads();
        for (int p = (int)(threadIdx.x >> 4); p < list_len; p += 16) {
            if (term < 9) {
                const long long v = (long long)dacc[term * DIRECT_CAP + p];
                if (v != 0) atomicAdd(grad_slot(a, __float_as_int(eC[p].w), term), (float)((double)v * fx_inv));
            }
        }
        return;
    }
    wave_sync();
    for (int base = 0; base < RW; base += 64) { // final flush: every slot that has an owner
        const int slot = base + lane;
        evict_lanes(__ballot(slot < RW && tag[min(slot, RW - 1)] >= 0), slot);
    }
}

} // namespace

#ifdef STP_REPLAY_STATS
extern "C" int stp_debug_replay_stats(unsigned long long* out16)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_replay_stats), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_replay_stats), z, sizeof(z));
    return (int)e;
}
#endif

hipError_t launch_hier_replay(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    if (f.s.sort_mode == MODE_KBUFFER) hipLaunchKernelGGL(render_hier_replay_kernel<3>, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(render_hier_replay_kernel<0>, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
