// tools/experiments/replay_transposed.hip -- REJECTED EXPERIMENT of round 3, kept for the record (not built, not shipped).
//
// A re-design of the replay backward: the 16 lanes of a DPP row are 16 consecutive RECORDS OF ONE PIXEL instead of 16 pixels
// (log layout [chunk][lane][16 records], transmittance and accumulated colour by 16-lane DPP prefix scans, per-row work
// lists in LDS, carries rotated into lane 0, colour sums as 32-bit fixed point).  The idea: a pixel never blends an entry
// twice, so a row's LDS adds hit distinct addresses and the same-address serialisation that bounds the shipped kernel goes
// away, along with the merge code.  It is correct (143 GPU tests green) and executes 19 % fewer VALU instructions
// (3.75e8 against 4.6e8 per C2-full launch) -- and runs in the same time:
//
//     C2-full, one MI355X, same box, alternating runs        shipped kernel   this kernel
//     render_replay_kernel                                     0.922-0.929 ms   0.921-0.927 ms   (first version 0.949)
//     recording forward (chunked log: a store touches 16 lines) 0.953-0.960 ms   0.985-0.995 ms   (+3 %)
//     windowed walk reading the chunked log (C3 / C5)          1.71 / 1.69 ms   1.91 / 1.89 ms   (+11 %)
//
// PMC (gpurun_out r03_t3b): VALU busy 68 %, LDS busy 51 % of which 58 % bank conflicts (a row's positions are distinct but
// not consecutive: 16 lanes on 16 bank pairs collide like birthdays), waves parked in s_waitcnt 51 % of their cycles.
// Ablations: without the LDS adds 0.734 ms, with all entry records from one cache line 0.759 ms, init + flush alone 0.067 ms
// -- three pipes (VALU, LDS atomics, the L1's gathers: a row's 16 records touch ~10 lines per load where 64 neighbouring
// pixels touch ~6) each 50-70 % busy and poorly overlapped at four waves per SIMD, which the 39 KB of LDS per workgroup
// fix.  Prefetching the LDS reads in front of the adds changed nothing (0.935 -> 0.925).  The forward's +3 % and the
// windowed walk's +11 % are pure loss, so the round-2 layout and kernel stay.
//
// ---------------------------------------------------------------------------------------------------------------------
// stp_render_replay.hip -- backward of the per-pixel-sort modes by REPLAYING the forward's blend log.
//
// No counterpart in the reference: its backward (hierarchical_render.cuh:1038-1175) re-runs the complete
// three-level resort to rediscover the order in which every pixel blended its Gaussians.  MI355X has 288 GB of
// HBM, so the training forward (render_hier_kernel<..., MODE_FWD_RECORD>, render_kbuffer_kernel<WIN, KB_FWD_RECORD>)
// simply writes that order down -- 2 bytes (the tile-list position) per blended (pixel, Gaussian) pair,
// BLEND_LOG_DEPTH = 256 records per pixel, 512 B per pixel, 1.07 GB at 1080p -- and this kernel walks the logs.
// The gradient maths per pair is the reference's (hierarchical_render.cuh:1094-1166); the result is the same sum
// in a different order.  Tiles whose log overflowed (a pixel with more than 256 blended entries, a list longer than
// 65535) are flagged by the forward and left to the re-sorting backward kernels, which then run only on those tiles.
//
// One 256-thread workgroup per tile, wave = row of four 4x4 sub-tiles as in the forward.  The nine gradient terms of a
// blend are summed on chip in ONE set of fixed-point sums per LIST POSITION shared by the workgroup (see
// stp_render_hier.inc for why not fp32 LDS atomics): 64-bit for the six geometric terms, 32-bit for the three colour
// terms (|alpha T dL/dpixel| <= max |dL/dpixel| bounds them), 60 B per position, 512 positions per window; the sums
// leave the chip once per window (16-lane group = one position, nine lanes = nine sums, one atomic instruction into
// the Gaussian's 64-byte gradient record).
//
// TRANSPOSED WALK (round 3; lists that fit one window, i.e. every tile of C2-full).  Rounds 1-2 gave every lane a
// PIXEL and walked the 64 logs in step: neighbouring pixels blend the same entries at the same time, so the nine LDS adds
// of a step serialise on equal addresses (1.75 lanes per address after DPP merging and de-phasing), and the transmittance
// chain makes every lane evaluate one blend per step.  Here the 16 lanes of a DPP row are 16 CONSECUTIVE RECORDS OF ONE
// PIXEL (the log keeps a pixel's records in 32-byte chunks for exactly this read):
//   * a pixel never blends an entry twice, so the 16 lanes of a row -- and mostly the four rows, which work on four
//     different sub-tiles -- add to DISTINCT positions: ds_add at its conflict-free rate, no merge code at all;
//   * the transmittance in front of each record is a prefix product of (1 - alpha) over the row, the colour accumulated
//     up to it a prefix sum -- four 16-lane DPP scans per step instead of a serial chain per lane (the forward logged only
//     blends that it performed, so nothing has to be re-decided here: no saturation test, no early exit);
//   * each row walks ITS OWN sub-tile's 16 pixels chunk by chunk at its own pace (a pixel of n records costs ceil(n/16)
//     steps of its row; lanes beyond n idle), so no lane waits for the wave's longest log.
// Longer lists (C2-min partly, C3, C5) -- and tiles where a sub-tile's pixels blend more than 124 x 16 records together --
// keep the pixel-per-lane walk, window by window (below).
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

#ifdef STP_REPLAY_STATS
__device__ unsigned long long g_replay_stats[16];
#endif

namespace {

#ifndef STP_REPLAY_PAIRMERGE
#define STP_REPLAY_PAIRMERGE 1 // windowed walk, merge levels: 1 = inside 2x2 quads (lane^1, lane^2); + the two mirror levels where the splats are large
#endif
#ifndef STP_REPLAY_OCC
#define STP_REPLAY_OCC 4
#endif
#ifndef STP_REPLAY_WINDOW
#define STP_REPLAY_WINDOW 512
#endif
#ifndef STP_REPLAY_TRANSPOSED
#define STP_REPLAY_TRANSPOSED 1 // 0: the pixel-per-lane walk for every tile (rounds 1-2), for A/B runs
#endif
constexpr int WINDOW = STP_REPLAY_WINDOW; // list positions per window (512 x 60 B = 30 KB of LDS + 8 KB pixel table: four workgroups per CU)
constexpr int EXHAUSTED = 0x7fffffff; // "position" of a lane that has no record left
constexpr int ITEM_CAP = 124, ITEM_STRIDE = ITEM_CAP + 4; // transposed walk: work items a row may have (16 pixels x chunks), + 4 slots of lookahead

__device__ __forceinline__ int replay_remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- 16-lane DPP row helpers (gfx9 DPP: row_shr without bound_ctrl leaves lanes without a source untouched) ----------
template <int N> __device__ __forceinline__ float row_shr_or(float fill, float v) // lane x: v of lane x - N, or fill for x < N
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x110 + N, 0xF, 0xF, false));
}
// Inclusive scans over the 16 lanes of a row, one DPP instruction per step: `v_op_dpp v, v, v row_shr:N` without bound_ctrl
// leaves the lanes that have no source (x < N) as they are, which is exactly the Hillis-Steele step.  (The compiler's own
// lowering of update_dpp is three instructions per step: v_mov identity, v_mov_dpp, the operation.)  s_nop 1: a DPP
// operand must not have been written by a VALU instruction in the two preceding issue slots, and the hazard recognizer does
// not look inside inline asm; in the three interleaved sums the other two chains fill the slots.
__device__ __forceinline__ void row_scan_mul(float& v) // inclusive prefix product over the row
{
    asm volatile("s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
}
__device__ __forceinline__ void row_scan_add3(float& a0, float& a1, float& a2) // three inclusive prefix sums, interleaved
{
#define STP_SCAN3(N)                                                                                                     \
    "v_add_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_add_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_add_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" STP_SCAN3(1) STP_SCAN3(2) STP_SCAN3(4) STP_SCAN3(8) "s_nop 1" : "+v"(a0), "+v"(a1), "+v"(a2));
#undef STP_SCAN3
}
__device__ __forceinline__ float row_ror1(float v) // lane x: v of lane (x - 1) mod 16 of my row -- lane 0 receives lane 15's
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xF, 0xF, true)); // row_ror:1 (every lane has a source)
}

// (One kernel for every kind of tile: as two launches the mixed case -- C2-min -- loses more to the half-empty grids
// than the lean loops gain.)
__global__ void __launch_bounds__(256, STP_REPLAY_OCC) render_replay_kernel(const RenderArgs a)
{
    __shared__ unsigned long long s_acc64[6 * WINDOW]; // [term 3..8][position - window start]
    __shared__ unsigned int s_acc32[3 * WINDOW];       // [term 0..2][position - window start]
    __shared__ float4 s_pix[4 * 64 * 2];               // transposed walk: what a blend needs of its pixel, [wave][pixel][2]
    __shared__ uint16_t s_items[16 * ITEM_STRIDE];     // transposed walk: every row's work items (see below), [wave][row][item]
    __shared__ float s_md[4];
    __shared__ int s_nitems[4];                        // transposed walk: items of each wave's longest row; -1 = some row's list does not fit

    const int lane = (int)(threadIdx.x & 63), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = replay_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    if (a.tile_flags[tile] != 0u) return; // log overflow: the re-sorting backward takes this tile
    const uint2 range = a.ranges[tile];
    const int px = tile_x * TILE + 4 * s + 2 * (m & 1) + (q & 1), py = tile_y * TILE + 4 * w + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;
    const int list_len = (int)(range.y - range.x);
    if (list_len <= 0) return;

    for (int i = (int)threadIdx.x; i < 6 * WINDOW; i += 256) s_acc64[i] = 0ull;
    for (int i = (int)threadIdx.x; i < 3 * WINDOW; i += 256) s_acc32[i] = 0u;

    BwdPixel bp;
    init_bwd_pixel(bp, a, inside, px, py);
    int n = inside ? min((int)a.n_contrib[(size_t)a.W * py + px], BLEND_LOG_DEPTH) : 0;
    float md = fmaxf(fmaxf(fabsf(bp.dL_dpix[0]), fabsf(bp.dL_dpix[1])), fabsf(bp.dL_dpix[2]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmaxf(md, __shfl_xor(md, off));
    // fixed-point scale of the sums (stp_render_hier.inc: "on-chip gradient window"): one for the workgroup
    if (lane == 0) s_md[w] = md;
    // the pixel table of the transposed walk: (dL/dpixel rgb, colour.r) (colour.g, colour.b, -T_final * sum(bg dL/dpixel), px | py << 16)
    s_pix[(w * 64 + lane) * 2 + 0] = make_float4(bp.dL_dpix[0], bp.dL_dpix[1], bp.dL_dpix[2], bp.final_color[0]);
    s_pix[(w * 64 + lane) * 2 + 1] = make_float4(bp.final_color[1], bp.final_color[2], -bp.T_final * bp.bg_dot, __uint_as_float((uint32_t)px | ((uint32_t)py << 16)));
    // Work items of the transposed walk.  Item = one chunk of 16 records of one pixel: j (the pixel inside its sub-tile, bits
    // 0-3) | c (the chunk, bits 4-7) | v (valid records in it, 1..16, bits 8-12) | bit 13: the pixel's next chunk follows.
    // Lane x owns pixel x of its row: it knows its own chunk count, a 16-lane prefix sum gives its first slot, and it writes its
    // chunks there; 0 = no item (a row that has fewer items than the wave's longest, and the two slots of lookahead).
    bool transposed = STP_REPLAY_TRANSPOSED && list_len <= WINDOW; // (workgroup-uniform; a row with more than ITEM_CAP items: see below)
    if (transposed) {
        uint16_t* const my_items = s_items + (w * 4 + s) * ITEM_STRIDE;
        for (int i = x; i < ITEM_STRIDE; i += 16) my_items[i] = 0;
        const int chunks = (n + 15) >> 4;
        int first = chunks; // inclusive prefix sum over the row, then exclusive
        first += __builtin_amdgcn_update_dpp(0, first, 0x111, 0xF, 0xF, false); // row_shr:1
        first += __builtin_amdgcn_update_dpp(0, first, 0x112, 0xF, 0xF, false); // row_shr:2
        first += __builtin_amdgcn_update_dpp(0, first, 0x114, 0xF, 0xF, false); // row_shr:4
        first += __builtin_amdgcn_update_dpp(0, first, 0x118, 0xF, 0xF, false); // row_shr:8
        int row_total = x == 15 ? first : 0; // (lane 15 holds its row's total)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) row_total = max(row_total, __shfl_xor(row_total, off));
        first -= chunks;
        wave_sync(); // (the zero fill above is in LDS before the items land on it)
        if (row_total <= ITEM_CAP)
            for (int c = 0; c < chunks; c++)
                my_items[first + c] = (uint16_t)(x | (c << 4) | (min(n - 16 * c, 16) << 8) | (c + 1 < chunks ? 0x2000 : 0));
        if (lane == 0) s_nitems[w] = row_total <= ITEM_CAP ? row_total : -1;
    }
    __syncthreads(); // (also: the accumulators are zeroed)
    if (transposed) transposed = s_nitems[0] >= 0 && s_nitems[1] >= 0 && s_nitems[2] >= 0 && s_nitems[3] >= 0; // else: the pixel-per-lane walk below
    md = fmaxf(fmaxf(s_md[0], s_md[1]), fmaxf(s_md[2], s_md[3]));
    // M = 2^md_exp >= max |dL/dpixel| of the tile.  A term t of the six geometric sums is stored as round(t 2^31 / M) in
    // 64 bits (|t| < 2^20 M fits 51 bits), a colour term -- |alpha T dL/dpixel| < M -- as round(t 2^22 / M) in 32 bits
    // (256 pixels: |sum| < 2^30; resolution M 2^-23, finer than an fp32 sum of that size keeps).  Anything that does not fit
    // (never seen; NaN included) goes to memory directly, as does everything of a tile whose M is not finite.
    int md_exp = 0;
    const bool md_ok = md > 0.0f && md < 3.0e38f;
    if (md_ok) (void)frexpf(md, &md_exp);
    md_exp = max(md_exp, -100);
    const double fx_scale = ldexp(1.0, 31 - md_exp), fx_inv = ldexp(1.0, md_exp - 31);
    const float fx_scale32 = ldexpf(1.0f, 22 - md_exp), fx_inv32 = ldexpf(1.0f, md_exp - 22);
    const float fx_cap = (md_ok || md == 0.0f) ? ldexpf(1.0f, min(md_exp + 20, 126)) : 0.0f; // (0: nothing fits, everything goes to memory)

    const char* const log_wave = log_wave_slice(a.blend_log, tile, w);
    const float4* const eC = a.entC + range.x; // list-ordered entry records: mean + Gaussian id, conic + opacity, colour
    const float4* const eD = a.entD + range.x;
    const float4* const eF = a.entF + range.x;
    struct Entry { float4 c, d, f; };
    auto entry_at = [&](int p) __attribute__((always_inline)) { // (a harmless read of entry 0 where there is no record: no branch)
        const uint32_t off = (uint32_t)((uint32_t)p < (uint32_t)list_len ? p : 0) << 4;
        auto at = [&](const float4* base) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off); };
        return Entry{at(eC), at(eD), at(eF)};
    };
    // nine terms of one lane -> the window's sums (lo = first position of the window); everything that does not fit the
    // fixed point, or lies in front of the window, goes to the Gaussian's gradient record in memory
    auto add_terms = [&](int cur_pos, int cur_id, const float (&g)[9], int lo, const bool merged) __attribute__((always_inline)) {
        // (the colour terms need no range test: |alpha T dL/dpixel| < M by construction, and a lane merged by DPP carries at
        // most 16 of them, which the conversion below covers)
        float gmax = fmaxf(fmaxf(fabsf(g[3]), fabsf(g[4])), fabsf(g[5]));
        gmax = fmaxf(fmaxf(gmax, fabsf(g[6])), fmaxf(fabsf(g[7]), fabsf(g[8])));
        if (cur_pos >= lo && gmax < fx_cap) {
            const int p = cur_pos - lo;
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                int qv;
                if (merged) qv = __float2int_rn(g[kk] * fx_scale32);
                else qv = __float_as_int(fmaf(g[kk], fx_scale32, 12582912.0f)) - 0x4B400000; // round to nearest through 1.5 * 2^23 (|g scale| < 2^22)
                atomicAdd(&s_acc32[kk * WINDOW + p], (unsigned int)qv);
            }
#pragma unroll
            for (int kk = 3; kk < 9; kk++) {
                // round-to-nearest integer of g*scale through the 1.5*2^52 trick (|g*scale| < 2^51 + margin)
                const double tq = fma((double)g[kk], fx_scale, 6755399441055744.0);
                const long long qv = __double_as_longlong(tq) - 0x4338000000000000ll;
                atomicAdd(&s_acc64[(kk - 3) * WINDOW + p], (unsigned long long)qv);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk]);
        }
    };
    // the window's sums leave the chip: 16-lane group = one position, nine lanes = its nine sums, one atomic
    // instruction (one request) into the Gaussian's 64-byte gradient record
    auto flush_window = [&](int lo) __attribute__((always_inline)) {
        __syncthreads();
        const int term = (int)(threadIdx.x & 15), cnt = min(WINDOW, list_len - lo);
        for (int p = (int)(threadIdx.x >> 4); p < cnt; p += 16) {
            if (term < 3) {
                const int v = (int)s_acc32[term * WINDOW + p];
                if (v != 0) {
                    s_acc32[term * WINDOW + p] = 0u;
                    atomicAdd(grad_slot(a, __float_as_int(eC[lo + p].w), term), (float)v * fx_inv32);
                }
            } else if (term < 9) {
                const long long v = (long long)s_acc64[(term - 3) * WINDOW + p];
                if (v != 0) {
                    s_acc64[(term - 3) * WINDOW + p] = 0ull;
                    atomicAdd(grad_slot(a, __float_as_int(eC[lo + p].w), term), (float)((double)v * fx_inv));
                }
            }
        }
    };

    if (transposed) {
        // ---- transposed walk: row s of the wave = sub-tile s; its 16 lanes = 16 consecutive records of one of its pixels ----
        // Work item = one chunk (records 16c .. 16c+15) of one pixel; a row's items are all chunks of its pixel 0, then of
        // pixel 1, ... (s_items, built above).  Items are processed in lock-step by the wave's four rows, item i of every
        // row in iteration i; three items are in flight: the log load of item i+2 and the entry loads of item i+1 are
        // issued before item i is evaluated (log record -> entry record are dependent loads).  Two register sets alternate
        // (the loop body is written out twice): no register moves at the back edge.
        const uint16_t* const my_items = s_items + (w * 4 + s) * ITEM_STRIDE;
        const char* const ptab = reinterpret_cast<const char*>(s_pix + (w * 64 + 16 * s) * 2); // my row's 16 pixels, 32 B each
        const uint32_t row_log = (uint32_t)(16 * s) * LOG_LANE_BYTES + (uint32_t)x * (uint32_t)sizeof(log_t);
        auto load_log = [&](uint32_t it) __attribute__((always_inline)) -> int { // the item's record of my lane
            const uint32_t off = row_log + ((it & 15u) << 5) + ((it & 0xF0u) << 7); // pixel j: 32 B, chunk c: 2 KB
            return (int)*reinterpret_cast<const log_t*>(log_wave + off);
        };
        auto valid_in = [&](uint32_t it) __attribute__((always_inline)) -> bool { return (uint32_t)x < ((it >> 8) & 31u); };
        const int n_items = __builtin_amdgcn_readfirstlane(s_nitems[w]); // items of the wave's longest row (wave-uniform)
        const float half_W = 0.5f * (float)a.W, half_H = 0.5f * (float)a.H;
        // carries between the chunks of one pixel, valid in lane 0 of the row (identity elsewhere): the transmittance and the
        // colour accumulated in front of the chunk; lane 0 folds them into its own factor / term, and the scans do the rest
        float cT = 1.0f, cC0 = 0.0f, cC1 = 0.0f, cC2 = 0.0f;
        const bool lane0 = x == 0;
        // What a step reads from LDS -- its item, the item two ahead, its pixel's table row -- is read one step EARLIER, in front
        // of that step's nine LDS adds: DS operations of a wave complete in order, so a read issued behind the adds would wait
        // for all of them (measured: half of the wave cycles of the first version were spent in s_waitcnt).
        struct Pre { uint32_t it, itn, itl; float4 p0, p1; }; // item i, i+1, i+2 and the table row of item i's pixel
        auto prefetch = [&](int i, uint32_t it, uint32_t itn) __attribute__((always_inline)) -> Pre {
            Pre r;
            r.it = it; r.itn = itn; r.itl = my_items[i + 2];
            r.p0 = *reinterpret_cast<const float4*>(ptab + ((it & 15u) << 5));
            r.p1 = *reinterpret_cast<const float4*>(ptab + ((it & 15u) << 5) + 16);
            return r;
        };
        auto step = [&](int i, const Pre& pre, Pre& pre_next, const Entry& en, const int cur_pos, const int pos_next, Entry& en_next, int& pc_next, int& pos_new) __attribute__((always_inline)) {
            const uint32_t it = pre.it, itn = pre.itn, itl = pre.itl;
            // the next round of loads, before this item's data is touched
            pc_next = valid_in(itn) ? pos_next : 0;
            en_next = entry_at(pc_next);
            pos_new = load_log(itl);
            pre_next = prefetch(i + 1, itn, itl);
            // ---- this item: up to 16 records of pixel j ----
            const bool valid = valid_in(it);
            const bool cont = (it & 0x2000u) != 0u; // the row's next item is the next chunk of the same pixel
            const float4 p0 = pre.p0, p1 = pre.p1;
            const uint32_t pxy = __float_as_uint(p1.w);
            const float pxf = (float)(pxy & 0xFFFFu), pyf = (float)(pxy >> 16);
            const float4 co = en.d;
            const float dx = en.c.y - pxf, dy = en.c.z - pyf;
            const float G = exp_blend(blend_power(dx, dy, co));
            const float alpha = valid ? fminf(0.99f, co.w * G) : 0.0f; // (the forward's alpha, bit for bit: same record, same operations)
            const float one_m = 1.0f - alpha;
            float P = one_m * cT;                                // lane 0 brings the transmittance in front of the chunk
            row_scan_mul(P);                                     // P = transmittance behind my record
            const float T_bef = row_shr_or<1>(cT, P);            // ... in front of it
            const float wgt = alpha * T_bef;
            float C0 = fmaf(en.f.x, wgt, cC0), C1 = fmaf(en.f.y, wgt, cC1), C2 = fmaf(en.f.z, wgt, cC2);
            row_scan_add3(C0, C1, C2);                           // colour accumulated up to and including my record
            // carries into the pixel's next chunk: lane 15's values, rotated into lane 0
            {
                const float rT = row_ror1(P), r0 = row_ror1(C0), r1 = row_ror1(C1), r2 = row_ror1(C2);
                const bool keep = cont && lane0;
                cT = keep ? rT : 1.0f; cC0 = keep ? r0 : 0.0f; cC1 = keep ? r1 : 0.0f; cC2 = keep ? r2 : 0.0f;
            }
            if (valid) {
                // gradient of this record (reference hierarchical_render.cuh:1094-1166; front-to-back form of stp_blend.h)
                float g[9];
                const float rcp_T = __builtin_amdgcn_rcpf(P), rcp_1ma = __builtin_amdgcn_rcpf(one_m);
                float dL_dalpha = (en.f.x - (p0.w - C0) * rcp_T) * p0.x;
                dL_dalpha = fmaf(en.f.y - (p1.x - C1) * rcp_T, p0.y, dL_dalpha);
                dL_dalpha = fmaf(en.f.z - (p1.y - C2) * rcp_T, p0.z, dL_dalpha);
                dL_dalpha = fmaf(dL_dalpha, T_bef, p1.z * rcp_1ma);
                g[0] = wgt * p0.x; g[1] = wgt * p0.y; g[2] = wgt * p0.z;
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                g[3] = dL_dG * dG_ddelx * half_W;
                g[4] = dL_dG * dG_ddely * half_H;
                const float hG = -0.5f * dL_dG;
                g[5] = hG * gdx * dx;
                g[6] = hG * gdx * dy;
                g[7] = hG * gdy * dy;
                g[8] = G * dL_dalpha;
                add_terms(cur_pos, __float_as_int(en.c.w), g, 0, false);
            }
        };
        int posA = load_log(my_items[0]), posB = load_log(my_items[1]);
        int pcA = valid_in(my_items[0]) ? posA : 0, pcB = 0;
        Entry enA = entry_at(pcA), enB;
        Pre preA = prefetch(0, my_items[0], my_items[1]), preB;
        for (int i = 0; i < n_items; i += 2) {
            step(i, preA, preB, enA, pcA, posB, enB, pcB, posA);
            if (i + 1 >= n_items) break;
            step(i + 1, preB, preA, enB, pcB, posA, enA, pcA, posB);
        }
        flush_window(0);
        return;
    }

    // ---- longer lists: every lane walks its own pixel's log, window by window; a lane pauses at its first record beyond
    // ---- the window, the workgroup meets at a barrier, writes the window out and moves on -- every pixel visits the list
    // ---- in (nearly) increasing position, so only the few records that the re-sort moved across a window boundary fall
    // ---- back to global atomics.  Lanes that hold the same position merge their terms pairwise with DPP first (the LDS
    // ---- adds serialise on equal addresses).
    const float pxf = (float)px, pyf = (float)py;
    auto log_at = [&](uint32_t k) __attribute__((always_inline)) -> int { // record k of this lane's pixel
        return (int)*reinterpret_cast<const log_t*>(log_wave + log_offset((uint32_t)lane, k));
    };
    // the gradient terms of one record (reference maths); false = nothing to add (no record, or the pixel saturates here)
    auto blend_terms = [&](bool act, const Entry& cur, float (&g)[9]) __attribute__((always_inline)) -> bool {
        bool ok = false;
        if (act) {
            FrontData fd;
            fd.co = cur.d;
            fd.xy = make_float2(cur.c.y, cur.c.z);
            fd.c[0] = cur.f.x; fd.c[1] = cur.f.y; fd.c[2] = cur.f.z;
            const float dx = fd.xy.x - pxf, dy = fd.xy.y - pyf;
            const float power = blend_power(dx, dy, fd.co);
            const float G = exp_blend(power);
            ok = blend_backward_terms(bp, a, px, py, fd, G, g);
        }
        if (!ok) { // (the merge below multiplies a partner that does not blend by zero: its terms must be finite)
#pragma unroll
            for (int kk = 0; kk < 9; kk++) g[kk] = 0.0f;
        }
        return ok;
    };
    // merge lanes on the same position, then add to the window's sums (lo = first position of the window)
    // deep (wave-uniform): also the two mirror levels inside the 16-lane row
    auto merge_and_add = [&](bool ok, int cur_pos, int cur_id, float (&g)[9], int lo, const bool deep) __attribute__((always_inline)) {
#if STP_REPLAY_PAIRMERGE
        // Pairwise merge (DPP): a lane and its partner -- lane^1, lane^2, then the mirror lanes of its 8-lane half and
        // row -- that hold the same list position sum their terms in registers and only one of them goes to LDS.  The
        // partner's value enters as the DPP operand of one v_fmac per term; a partner that does not match is SELECTED away
        // (its terms may be stale -- a lane that does not blend keeps its last values -- and 0 * Inf would poison the sum).
        {
            int key = ok ? cur_pos : -2 - lane; // unique when not blending
#define STP_MERGE_LEVEL(CTRL, LOWER)                                                                                    \
            {                                                                                                       \
                const int pk = __builtin_amdgcn_mov_dpp(key, CTRL, 0xF, 0xF, true);                                 \
                const bool match = pk == key;                                                                       \
                const float mf = (match && (LOWER)) ? 1.0f : 0.0f;                                                  \
                /* g[] was written by ordinary VALU instructions a moment ago: the guard takes the registers in and hands them */ \
                /* out again, so that no write can sink below it (DPP read-after-write hazard, tools/check_dpp_hazards.py) */     \
                asm volatile("s_nop 1" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]), "+v"(g[8])); \
                _Pragma("unroll") for (int kk = 0; kk < 9; kk++) g[kk] = partner_fma<CTRL>(g[kk], mf, g[kk]);       \
                if (match && !(LOWER)) { ok = false; key = -2 - lane; }                                             \
            }
            STP_MERGE_LEVEL(0xB1, (q & 1) == 0) // partner lane ^ 1 (quad_perm [1,0,3,2])
            STP_MERGE_LEVEL(0x4E, (q & 2) == 0) // partner lane ^ 2 (quad_perm [2,3,0,1])
            if (deep) {
                STP_MERGE_LEVEL(0x141, (x & 7) < 4)  // partner 7 - i inside each 8-lane half (row_half_mirror)
                STP_MERGE_LEVEL(0x140, x < 8)        // partner 15 - i inside the 16-lane row (row_mirror)
            }
#undef STP_MERGE_LEVEL
        }
#endif
        if (ok) add_terms(cur_pos, cur_id, g, lo, true);
    };

    // Where the splats are larger than the wave's 16x4 pixels every lane blends the same entries: the two mirror levels of
    // the merge are what keeps the LDS adds apart there.  Decided once per wave: do most of its pixels START on the same entry?
    bool same_start;
    {
        const int p0 = n > 0 ? log_at(0) : 0x7fffffff;
        int pmin = p0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pmin = min(pmin, __shfl_xor(pmin, o));
        same_start = __popcll(__ballot(p0 == pmin && n > 0)) >= 40;
    }
    float g[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // Two dependent loads lead to a blend: log record (list position) -> the entry's record.  They are software
    // pipelined one step apart: `pos` / `en` hold the lane's next record and its entry, `pos1` the position of the one
    // after, each loaded an iteration before it is needed.
    int k = 0; // records consumed by this lane
    int pos = (0 < n) ? log_at(0) : EXHAUSTED;
    int pos1 = (1 < n) ? log_at(1) : EXHAUSTED;
    Entry en = entry_at(pos);
    const int n_win = (list_len + WINDOW - 1) / WINDOW; // workgroup-uniform
    for (int win = 0; win < n_win; win++) {
        const int lo = win * WINDOW, hi = lo + WINDOW;
        for (;;) {
            const bool act = pos < hi; // my next record belongs to this window (or to an earlier one: a straggler)
            if (!__any(act)) break;
            const Entry cur = en;
            const int cur_pos = pos, cur_id = __float_as_int(cur.c.w);
            // issue the next round of loads before touching this step's data
            k += (int)act;
            const int rec = log_at((uint32_t)min(k + 1, BLEND_LOG_DEPTH - 1));
            pos = act ? pos1 : pos;
            pos1 = act ? (k + 1 < n ? rec : EXHAUSTED) : pos1;
            en = entry_at(pos);
            const bool ok = blend_terms(act, cur, g);
            if (act && !ok) { n = k; pos = EXHAUSTED; pos1 = EXHAUSTED; } // (saturated one record earlier than the forward said)
            merge_and_add(ok, cur_pos, cur_id, g, lo, same_start);
        }
        flush_window(lo);
        if (win + 1 < n_win) __syncthreads();
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
    hipLaunchKernelGGL(render_replay_kernel, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
