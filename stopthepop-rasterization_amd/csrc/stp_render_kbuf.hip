// stp_render_kbuf.hip -- PPX_KBUFFER forward passes (plain, recording, depth visualisation) on the wave64 machinery of the
// hierarchical kernel's head level.
//
// Replaces renderkBufferCUDA<3, W, false> (reference stopthepop/resorted_render.cuh:17-221): per pixel a sorted window of W
// entries keyed by the depth along the pixel's own ray; every entry of the tile's list is looked at in list order -- "if the
// window is full, blend its front; then, if the entry passes the tests, insert it" -- and the window is drained at the end.
//
// What the result depends on, and what it does not (the same argument as stp_render_hier.inc, filter_push): an entry that
// FAILS the tests can only cost the pixel a pop of the window's front -- which the next passing entry would have popped
// anyway before being inserted, and which a second failing entry no longer finds.  The sequence of blended entries is
// therefore the same whichever of the failing entries a pixel is shown, and a pixel's work is "its passing entries, in list
// order".  The previous kernel (stp_render_tile.hip, still the re-sorting BACKWARD; STP_KBUFFER=tile selects its forward) walked the whole
// list with all 64 lanes of a wave on the same entry: 600 cycles for every entry that reached any of the wave's 64 pixels,
// a third of the lanes doing anything.  Here:
//
//   * thread -> pixel as in the hierarchical and replay kernels (wave = row of four 4x4 sub-tiles, sub-tile = 16-lane DPP
//     row, 2x2 quad = DPP quad), so the recording forward writes the same blend log and the replay kernel is its backward;
//   * a batch of 32 list entries is tested against the wave's four sub-tiles by the 64 lanes together (lane = entry x pair of
//     sub-tiles): the EXACT minimum of the exponent's quadratic form over the sub-tile's rectangle, with a margin that
//     covers every rounding of the per-pixel evaluation -- a bound, unlike the reference's max-contribution estimate;
//     survivors are compacted per sub-tile with ballots, in list order;
//   * each quad tests a sub-tile's survivors against its own four pixels (lane q takes survivor 4g + q; quad_can_blend) and
//     parks what is left in its FIFO; head steps then run on groups of four parked entries with every quad of the wave on
//     its OWN entries -- the step itself (entry record fetched by one lane of the quad, DPP-operand evaluation, always-full
//     window with fused pop + insert) is the hierarchical head level's with HEAD = W.
//
// n_contrib: the reference counts the entries a pixel looked at before it saturated.  A pop that saturates is either the
// pop in front of the entry that follows the insertion which filled the window (count = that insertion's position + 1) or a
// pop of the final drain (count = the whole list); both are known here without having looked at the entries in between.
#include "stp_internal.h"
#include "stp_blend.h"

#include <cstdlib>
#include <cstring>

namespace stp {

#ifdef STP_KB_STATS
// debug build only: where in the window do candidates land?  [0..15] passing candidates by the number of window slots they pass on their way
// in from the back (0 = appended behind everything), [16..31] candidate steps by the LARGEST such distance among the wave's 64 lanes,
// [32] candidate steps (per wave), [33] passing (lane, candidate) pairs, [34] live lanes in those steps
static __device__ unsigned long long g_kb_stats[40];
extern "C" int stp_debug_kb_stats(unsigned long long* out40)
{
    hipError_t e = hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_kb_stats), sizeof(unsigned long long) * 40);
    unsigned long long z[40] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stats), z, sizeof(z));
    return (int)e;
}
#endif

namespace {

constexpr int KB_CAP = 32; // list positions a quad's FIFO may hold (one round of 16 survivors adds up to 16)

__device__ __forceinline__ int kb_remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// can the entry with record rows C (mean in .yz) and D (conic, opacity) reach 1/255 at any of the four pixels of the 2x2 quad whose CENTRE is (qxc, qyc)?
// (stp_render_hier.inc quad_can_blend: an upper bound of opacity * exp(power) over the four pixels that covers every rounding of the per-pixel
// evaluation; NaN is kept, the exact test decides)
__device__ __forceinline__ bool kb_quad_can_blend(const float4 C, const float4 D, const float qxc, const float qyc)
{
    const float dx = C.y - qxc, dy = C.z - qyc;
    const float gx = fmaf(D.y, dy, D.x * dx), gy = fmaf(D.z, dy, D.y * dx);
    const float q2 = fmaf(gy, dy, gx * dx);
    const float m2 = fminf(fmaf(D.y, 0.5f, -fabsf(gx + gy)), fmaf(D.y, -0.5f, -fabsf(gx - gy)));
    const float qmin2 = fmaf(D.x + D.z, 0.25f, q2) + m2; // 2 x the smallest negated exponent among the four pixels
    const float far = fmaxf(fabsf(dx), fabsf(dy)) + 0.5f;
    const float S = (fabsf(D.x) + fabsf(D.z) + fabsf(D.y)) * far * far;
    const float pup = fmaf(qmin2, -0.5f, S * 2.0e-6f);
    const float v = D.w * __builtin_amdgcn_exp2f(pup * 1.44269502162933349609375f);
    return !(v < ALPHA_THRESHOLD * 0.9999f);
}

constexpr int KBW_FWD = 0, KBW_RECORD = 2, KBW_DEPTH = 3; // (the values of the hierarchical kernel's modes)

template <int WIN> constexpr int kb_waves() { return WIN <= 4 ? 4 : WIN <= 16 ? 3 : 2; } // waves per SIMD the kernel is compiled for

template <int WIN, int MODE, bool FRCP>
__global__ void __launch_bounds__(256, kb_waves<WIN>()) render_kbuffer_wave_kernel(const RenderArgs a)
{
    constexpr bool RECORD = MODE == KBW_RECORD;
    constexpr bool DEPTHVIZ = MODE == KBW_DEPTH;
    __shared__ int s_stage[16 * 32];      // [sub-tile][survivor]: list positions of the staged batch, per sub-tile
    __shared__ int s_fifo[64 * KB_CAP];   // [quad][slot]

    const int lane = (int)(threadIdx.x & 63);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = a.tile_order ? (int)a.tile_order[blockIdx.x] : kb_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);
    const int cx = tile_x * TILE + 4 * s, cy = tile_y * TILE + 4 * w;
    const int px = cx + 2 * (m & 1) + (q & 1), py = cy + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;
    bool active = inside;
    if (total <= 0) { // an empty tile is background (and its "entry 0" -- what pads and stand-ins read -- may not exist: stp_render_hier.inc)
        if (inside) {
            const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * py + px;
            a.final_T[pid] = 1.0f;
            a.n_contrib[pid] = 0u;
            if constexpr (DEPTHVIZ) { a.out_color[pid] = 0.0f; a.out_color[N + pid] = 1.0f; }
            else { a.out_color[pid] = a.bg[0]; a.out_color[N + pid] = a.bg[1]; a.out_color[2 * N + pid] = a.bg[2]; }
        }
        return;
    }

    const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    const float3 pix_dir = view_ray(a.inv_vp, cam, (float)px, (float)py, a.W, a.H);

    const float4* const eA = a.entA + range.x;
    const float4* const eB = a.entB + range.x;
    const float4* const eC = a.entC + range.x;
    const float4* const eD = a.entD + range.x;
    const float4* const eF = a.entF + range.x;
    const int list_last = max(total - 1, 0);
    auto ent_row = [&](const float4* base, int pos) __attribute__((always_inline)) -> float4 { // SGPR base + 32-bit offset
#if defined(STP_KB_ENT_NT) && STP_KB_ENT_NT   // experiment: the entry records as non-temporal loads (they stream through; the log's lines should stay in the L2)
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + ((uint32_t)pos << 4)));
        return make_float4(v.x, v.y, v.z, v.w);
#endif
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + ((uint32_t)pos << 4));
    };

    // blend log (recording forward): blocked layout, whole pieces from blending lanes only (stp_blend.h: BlockedLogCursor)
    BlockedLogCursor logc{RECORD ? log_wave_slice(a.blend_log, tile, w, a.log_depth) : nullptr, 2u * (uint32_t)a.log_depth, (uint32_t)lane << LOG_PIECE_SHIFT};
    auto log_append = [&](bool upd, int pay) __attribute__((always_inline)) { logc.append(upd, pay); };
    auto log_records = [&]() __attribute__((always_inline)) -> int { return logc.records(); };
    auto log_finish = [&]() __attribute__((always_inline)) { logc.flush(); };

    Window<WIN> head;
    head.init_padded();
    FwdPixel fp;
    init_fwd_pixel(fp);
    float depth_acc = 0.0f;
    int contrib = total; // n_contrib of the plain / depth forward (see the header)
    int cfull = total;   // what it becomes if the NEXT pop saturates the pixel

    // The window does not carry alpha (W registers and W selects per step): it is evaluated again at the pop, from the same
    // record with the same operations, hence to the same bits as when the entry passed its tests.  What the pop needs of
    // the front entry -- mean, conic + opacity, colour -- is fetched when the entry BECOMES the front, one step earlier.
    float4 frC = make_float4(0, 0, 0, 0), frD = frC, frF = frC;
    auto fetch_front = [&]() __attribute__((always_inline)) {
        const int nx = head.id[0]; // (a pad carries position 0: a harmless read)
        frC = ent_row(eC, nx); frD = ent_row(eD, nx); frF = ent_row(eF, nx);
    };
    auto pop_forward = [&]() __attribute__((always_inline)) { // consumes slot 0 of the always-full window; replace_front() follows
        const float alpha0 = fminf(0.99f, frD.w * exp_blend(blend_power(frC.y - (float)px, frC.z - (float)py, frD)));
        const float test_T = fp.T * (1.0f - alpha0);
        const bool doing = active && !(head.depth[0] < 0.0f); // a pad in front = the reference's window is not full: nothing to blend
        const bool upd = doing && !(test_T < T_THRESHOLD);
        const int pay = head.id[0];
        const float wgt = upd ? alpha0 * fp.T : 0.0f;
        fp.C[0] = fmaf(wgt, frF.x, fp.C[0]); fp.C[1] = fmaf(wgt, frF.y, fp.C[1]); fp.C[2] = fmaf(wgt, frF.z, fp.C[2]);
        if constexpr (DEPTHVIZ) depth_acc += upd ? head.depth[0] * alpha0 * fp.T : 0.0f; // reference resorted_render.cuh:107
        fp.T = upd ? test_T : fp.T;
        if constexpr (RECORD) log_append(upd, pay);
        else contrib = (doing && !upd) ? cfull : contrib;
        active = active && (upd || !doing);
    };

    // four candidates per quad, lane q brings candidate q (list position, -1 = none): the hierarchical head level's step
    auto feed4_from = [&](const int fid) __attribute__((always_inline)) {
        const int pf = min(max(fid, 0), list_last);
        float4 eAq = ent_row(eA, pf), eBq = ent_row(eB, pf), eCq = ent_row(eC, pf), eDq = ent_row(eD, pf);
#ifdef STP_KB_STATS
#define STP_KB_STAT(PASS, D)                                                                                            \
        {                                                                                                               \
            int disp = 0;                                                                                               \
            for (int s_ = 1; s_ < WIN; s_++) disp += (int)((D) < head.depth[s_]);                                       \
            disp = (PASS) ? disp : -1;                                                                                  \
            int mx = disp;                                                                                              \
            for (int o_ = 1; o_ < 64; o_ <<= 1) mx = max(mx, __shfl_xor(mx, o_));                                       \
            if (disp >= 0) atomicAdd(&g_kb_stats[min(disp, 15)], 1ull);                                                 \
            const int np_ = __popcll(__ballot(PASS)), na_ = __popcll(__ballot(active));                                 \
            if (lane == 0) { atomicAdd(&g_kb_stats[16 + min(max(mx, 0), 15)], 1ull); atomicAdd(&g_kb_stats[32], 1ull);  \
                             atomicAdd(&g_kb_stats[33], (unsigned long long)np_); atomicAdd(&g_kb_stats[34], (unsigned long long)na_); } \
        }
#else
#define STP_KB_STAT(PASS, D)
#endif
#define STP_KB_FEED(I)                                                                                                  \
    {                                                                                                                   \
        pop_forward();                                                                                                  \
        const int cid = quad_bcast_i<I>(pf);                                                                            \
        if constexpr (I == 0) {                                                                                         \
            eDq.w = fid < 0 ? 0.0f : eDq.w; /* no candidate: alpha 0 fails the tests */                                 \
            dpp_hazard_guard_on(eDq.w);                                                                                 \
        } else dpp_hazard_guard();                                                                                      \
        const float depth = depth_along_ray_quad_ent<I, FRCP>(eAq, eBq, eCq, pix_dir);                                  \
        const float dx = quad_sub<I>(eCq.y, (float)px), dy = quad_sub<I>(eCq.z, (float)py);                             \
        const float power = blend_power_quad<I>(dx, dy, eDq);                                                           \
        const float alpha = min_099(quad_mul<I>(eDq.w, exp_blend(power)));                                              \
        const bool pass = active && !(depth < 0.0f) && !(power > 0.0f) && !(alpha < ALPHA_THRESHOLD);                   \
        STP_KB_STAT(pass, depth)                                                                                        \
        head.replace_front(pass, pass ? depth : -FLT_MAX, cid, 0.0f);                                                   \
        fetch_front();                                                                                                  \
        if constexpr (!RECORD) cfull = pass ? cid + 1 : cfull;                                                          \
    }
        STP_KB_FEED(0) STP_KB_FEED(1) STP_KB_FEED(2) STP_KB_FEED(3)
#undef STP_KB_FEED
    };

    int* const hfifo = s_fifo + ((w * 4 + s) * 4 + m) * KB_CAP;
    int hf_head = 0, hf_cnt = 0; // (quad-uniform)
    auto head_round = [&](const bool force) __attribute__((always_inline)) -> bool { // false: nothing (more) to do now
        const unsigned long long act = __ballot(active);
        const bool qlive = ((act >> (lane & ~3)) & 0xFull) != 0ull;
        if (!qlive) { hf_head = (hf_head + hf_cnt) & (KB_CAP - 1); hf_cnt = 0; } // nobody left to show them to
        bool go;
        if (force) go = __any(hf_cnt > 0);
        else go = __any(hf_cnt > KB_CAP - 16) || (__all(hf_cnt >= 4 || !qlive) && __any(hf_cnt >= 4));
        if (!go) return false;
        const int n = min(hf_cnt, 4);
        wave_sync();
        const int fid = q < n ? hfifo[(hf_head + q) & (KB_CAP - 1)] : -1;
        hf_head = (hf_head + n) & (KB_CAP - 1);
        hf_cnt -= n;
        feed4_from(fid);
        return true;
    };
    auto head_rounds = [&](const bool force) __attribute__((always_inline)) {
#pragma unroll 1
        for (;;) {
            if (!head_round(force)) break;
            if constexpr (WIN <= 8) { if (!head_round(force)) break; } // (two copies of the group step: fewer register shuffles per step)
        }
    };

    // ---- main loop: batches of 32 list entries -----------------------------------------------------------------------
    const int half = lane >> 5, e = lane & 31;
    const float sxA = (float)(tile_x * TILE + 8 * half), sxB = sxA + 4.0f, syf = (float)cy;
    int* const stA = s_stage + (w * 4 + 2 * half) * 32; // my half's two sub-tiles: [0..32) and [32..64)
    const int* const st_row = s_stage + (w * 4 + s) * 32;
    const float qxc = (float)(px - (q & 1)) + 0.5f, qyc = (float)(py - (q >> 1)) + 0.5f; // centre of my 2x2 quad
#pragma unroll 1
    for (int base = 0; base < total; base += 32) {
        if (!__any(active)) break;
        // stage: lane = entry e of the batch x the sub-tile pair of my half
        const int ep = base + e;
        bool keepA = false, keepB = false;
        if (ep < total) {
#if STP_CULL_MASK
            // (the sixteen sub-tile verdicts of this entry were computed by the entry gather, stp_tilesort.hip: subtile_keep_mask_kbuffer)
            const uint32_t mask = __float_as_uint(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(eF) + ((uint32_t)ep << 4) + 12));
            const uint32_t mine = mask >> (4 * w + 2 * half);
            keepA = (mine & 1u) != 0u;
            keepB = (mine & 2u) != 0u;
#else
            const float4 C = ent_row(eC, ep), D = ent_row(eD, ep);
            const float x0A = sxA - C.y, x0B = sxB - C.y, y0 = syf - C.z;
            const float pA = min_power_rect(D, x0A, x0A + 3.0f, y0, y0 + 3.0f);
            const float pB = min_power_rect(D, x0B, x0B + 3.0f, y0, y0 + 3.0f);
            // rounding of the per-pixel exponent against this one: at most a few ulp of the form's terms, all below T * far^2
            const float T3 = fabsf(D.x) + fabsf(D.y) + fabsf(D.z);
            const float fy = fmaxf(fabsf(y0), fabsf(y0 + 3.0f));
            const float fA = fmaxf(fmaxf(fabsf(x0A), fabsf(x0A + 3.0f)), fy), fB = fmaxf(fmaxf(fabsf(x0B), fabsf(x0B + 3.0f)), fy);
            keepA = !(D.w * __builtin_amdgcn_exp2f(fmaf(T3 * fA * fA, 2.0e-6f, -pA) * 1.44269502162933349609375f) < ALPHA_THRESHOLD * 0.9999f);
            keepB = !(D.w * __builtin_amdgcn_exp2f(fmaf(T3 * fB * fB, 2.0e-6f, -pB) * 1.44269502162933349609375f) < ALPHA_THRESHOLD * 0.9999f);
#endif
        }
        const unsigned long long balA = __ballot(keepA), balB = __ballot(keepB);
        const unsigned int mA = (unsigned int)(balA >> (32 * half)), mB = (unsigned int)(balB >> (32 * half));
        const unsigned int below = (1u << e) - 1u;
        wave_sync(); // (the previous batch's readers are done)
        if (keepA) stA[__popc(mA & below)] = ep;
        if (keepB) stA[32 + __popc(mB & below)] = ep;
        wave_sync();
        const int n_s = __popc((unsigned int)(((s & 1) ? balB : balA) >> (32 * (s >> 1)))); // my sub-tile's survivors
        // feed: groups of four survivors per quad, head steps after every 16
        int n_max = n_s;
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) n_max = max(n_max, __shfl_xor(n_max, o));
#pragma unroll 1
        for (int g0 = 0; g0 < n_max; g0 += 16) {
#pragma unroll 1
            for (int g = g0; g < min(g0 + 16, n_max); g += 4) {
                const int i = g + q;
                int fid = -1;
                if (i < n_s) fid = st_row[i];
                bool keep = false;
                const unsigned long long act = __ballot(active);
                const bool qlive = ((act >> (lane & ~3)) & 0xFull) != 0ull;
                if (fid >= 0 && qlive) keep = kb_quad_can_blend(ent_row(eC, fid), ent_row(eD, fid), qxc, qyc);
                int bits = keep ? (1 << q) : 0;
                bits += __builtin_amdgcn_mov_dpp(bits, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
                bits += __builtin_amdgcn_mov_dpp(bits, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
                if (keep) hfifo[(hf_head + hf_cnt + __popc(bits & ((1 << q) - 1))) & (KB_CAP - 1)] = fid;
                hf_cnt += __popc(bits);
            }
            head_rounds(false);
        }
    }
    head_rounds(true);
    // drain: fillers that sort LAST push the remaining real entries to the front, one per step
#pragma unroll 1
    for (int it = 0; it < WIN; it++) {
        pop_forward();
        head.replace_front(false, FLT_MAX, 0, 0.0f);
        fetch_front();
        cfull = total; // (only the drain's first pop can be the one "in front of the next entry")
    }

    if constexpr (RECORD) log_finish();
    if (inside) {
        const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * py + px;
        a.final_T[pid] = fp.T;
        a.n_contrib[pid] = RECORD ? (uint32_t)log_records() : (uint32_t)contrib; // (recording forward: the pixel's number of log records)
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
        if (log_records() > a.log_depth || total > LOG_MAX_LIST) a.tile_flags[tile] = 1u; // log overflow: this tile's backward re-sorts
        report_log_need(a.log_need, log_records(), a.log_tag);
    }
}


// ---- the window as a per-lane RING in LDS ----------------------------------------------------------------------------------------------
// Where do candidates land in a pixel's window?  Measured on C3 (3M Gaussians, window 16; tools/kb_stats.py, profiles/r05_experiments/kb_stats_c3.txt):
// 92.7 % of the passing candidates are appended BEHIND every entry of the window, 6.6 % pass one entry, 0.7 % two or more; the LARGEST distance
// among the 64 lanes of a wave is 0 in 38 % of the candidate steps, 1 in 52 %, 2 in 9 %, above 2 in 1 %.  The register window above pays for the
// general case in every step: the front is consumed and all W slots move (15 compares, 14 tie terms, 30 payload selects, 16 medians at W = 16 --
// 75 half-rate instructions, 32 VGPRs, three waves per SIMD).  Here the window of a pixel is a ring of (depth, list position) records in the
// lane's own LDS column ([slot][thread]: the bank of an access is the lane, whatever the slot -- every lane may sit at its own ring position
// without a bank conflict), with a per-lane head and count:
//   * a candidate that fails its tests is a no-op for its pixel (the argument of this file's header: pop only in front of a passing candidate);
//   * a pop advances the head: nothing moves;
//   * an insertion walks in from the back while the entry in front of it is deeper (strict: a new entry goes behind equals, like the
//     reference's swap loop, resorted_render.cuh:187-196) -- one LDS round per entry passed, 0.74 rounds per step on average for the wave;
//   * the reference's swap loop is not a plain insertion when it carries an entry past EQUAL depths (the carried entry does not swap with
//     its equals: every run of equal depths among the displaced entries ends up rotated by one): such a step is detected while walking
//     (two consecutive displaced entries of equal depth) and its payloads are rotated afterwards, in a branch that is almost never entered.
// What the pop needs of the front entry is fetched when the entry becomes the front, as above; alpha is evaluated again at the pop.
constexpr int KBR_CAP = 24; // list positions a quad's FIFO may hold in the ring kernel (a round of 16 survivors adds up to 16; not a power of two: 40 KB of LDS = four workgroups per CU)
template <int WIN> constexpr int kb_ring_waves() { return WIN <= 16 ? 4 : 3; } // (LDS: 40 KB per workgroup at 16 entries, 48 / 56 KB at 20 / 24)
template <int WIN> constexpr size_t kb_ring_lds() { return (size_t)WIN * 256 * 8 + 16 * 32 * 4 + 64 * KBR_CAP * 4; }

template <int WIN, int MODE, bool FRCP>
__global__ void __launch_bounds__(256, kb_ring_waves<WIN>()) render_kbuffer_ring_kernel(const RenderArgs a)
{
    constexpr bool RECORD = MODE == KBW_RECORD;
    constexpr bool DEPTHVIZ = MODE == KBW_DEPTH;
    constexpr bool POW2 = (WIN & (WIN - 1)) == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;                                                  // (depth, list position) [WIN][256]: slot-major, 8 bytes per thread
    int* const s_stage = reinterpret_cast<int*>(smem + (size_t)WIN * 2048);   // [sub-tile][survivor]
    int* const s_fifo = s_stage + 16 * 32;                                    // [quad][slot]

    const int lane = (int)(threadIdx.x & 63);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = a.tile_order ? (int)a.tile_order[blockIdx.x] : kb_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);
    const int cx = tile_x * TILE + 4 * s, cy = tile_y * TILE + 4 * w;
    const int px = cx + 2 * (m & 1) + (q & 1), py = cy + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;
    bool active = inside;
    if (total <= 0) { // an empty tile is background (and its "entry 0" -- what pads and stand-ins read -- may not exist: stp_render_hier.inc)
        if (inside) {
            const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * py + px;
            a.final_T[pid] = 1.0f;
            a.n_contrib[pid] = 0u;
            if constexpr (DEPTHVIZ) { a.out_color[pid] = 0.0f; a.out_color[N + pid] = 1.0f; }
            else { a.out_color[pid] = a.bg[0]; a.out_color[N + pid] = a.bg[1]; a.out_color[2 * N + pid] = a.bg[2]; }
        }
        return;
    }

    const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    const float3 pix_dir = view_ray(a.inv_vp, cam, (float)px, (float)py, a.W, a.H);

    const float4* const eA = a.entA + range.x;
    const float4* const eB = a.entB + range.x;
    const float4* const eC = a.entC + range.x;
    const float4* const eD = a.entD + range.x;
    const float4* const eF = a.entF + range.x;
    const int list_last = max(total - 1, 0);
    auto ent_row = [&](const float4* base, int pos) __attribute__((always_inline)) -> float4 { // SGPR base + 32-bit offset
#if defined(STP_KB_ENT_NT) && STP_KB_ENT_NT   // experiment: the entry records as non-temporal loads (they stream through; the log's lines should stay in the L2)
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + ((uint32_t)pos << 4)));
        return make_float4(v.x, v.y, v.z, v.w);
#endif
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + ((uint32_t)pos << 4));
    };

    // blend log (recording forward): blocked layout, whole pieces from blending lanes only (stp_blend.h: BlockedLogCursor)
    BlockedLogCursor logc{RECORD ? log_wave_slice(a.blend_log, tile, w, a.log_depth) : nullptr, 2u * (uint32_t)a.log_depth, (uint32_t)lane << LOG_PIECE_SHIFT};
    auto log_append = [&](bool upd, int pay) __attribute__((always_inline)) {
#if defined(STP_KB_LOG_ABLATE) && STP_KB_LOG_ABLATE == 1   // timing experiment (log WRONG): every store of the wave into its first block -- same instructions, few lines
        *reinterpret_cast<log_t*>(logc.wave + logc.piece) = (log_t)pay; logc.j2 += upd ? 2u : 0u;
#elif defined(STP_KB_LOG_ABLATE) && STP_KB_LOG_ABLATE == 2 // ... no store at all
        logc.j2 += upd ? 2u : 0u;
#elif defined(STP_KB_LOG_ABLATE) && STP_KB_LOG_ABLATE == 4 // ... the unconditional 2-byte store of rounds 4-5 (into the slot of the lane's next record, or the spare block)
        *reinterpret_cast<log_t*>(logc.wave + log_record_offset<true>(min(logc.j2, logc.cap2), logc.piece)) = (log_t)pay; logc.j2 += upd ? 2u : 0u;
#else
        logc.append(upd, pay);
#endif
    };
    auto log_records = [&]() __attribute__((always_inline)) -> int { return logc.records(); };
    auto log_flush = [&]() __attribute__((always_inline)) { logc.flush(); };

    // the ring: logical entry k of my window lives in slot (rh + k) mod WIN of my column
    const uint32_t col = (uint32_t)threadIdx.x * 8u;
    int rn = 0, rh = 0;          // entries in my window, slot of its front
    float back_d = -FLT_MAX;     // my window's LAST entry (-FLT_MAX: the window is empty) and the one in front of it (-FLT_MAX: none): nine candidates
    int back_i = 0;              // in ten are placed against these two without an LDS access
    float back2_d = -FLT_MAX;
    int back2_i = 0;
    auto wrap = [&](int p) __attribute__((always_inline)) -> int { // p in [0, 2 WIN) -> [0, WIN)
        if constexpr (POW2) return p & (WIN - 1);
        else return p - (p >= WIN ? WIN : 0);
    };
    auto prev_slot = [&](int p) __attribute__((always_inline)) -> int { // p in [0, WIN) -> the slot in front of it
        if constexpr (POW2) return (p - 1) & (WIN - 1);
        else return (p == 0 ? WIN : p) - 1;
    };
    auto slot_addr = [&](int p) __attribute__((always_inline)) -> char* { return ring + (((uint32_t)p << 11) + col); };
    auto rd = [&](int p) __attribute__((always_inline)) -> float2 { return *reinterpret_cast<const float2*>(slot_addr(p)); }; // (.x depth, .y the position's bits)
    auto wr = [&](int p, float d, int i) __attribute__((always_inline)) { *reinterpret_cast<float2*>(slot_addr(p)) = make_float2(d, __int_as_float(i)); };

    FwdPixel fp;
    init_fwd_pixel(fp);
    float depth_acc = 0.0f;
    int contrib = total; // n_contrib of the plain / depth forward (see the header)
    int cfull = total;   // what it becomes if the NEXT pop saturates the pixel

    // the front entry's record rows (mean, conic + opacity, colour), fetched when an entry becomes the front
    float4 frC = make_float4(0, 0, 0, 0), frD = frC, frF = frC;
    int fr_id = 0;
    float fr_depth = 0.0f;
    auto fetch_front = [&]() __attribute__((always_inline)) { // (an empty window reads a stale slot: a harmless, clamped load)
        const float2 r = rd(rh);
        fr_id = min(max(__float_as_int(r.y), 0), list_last);
        if constexpr (DEPTHVIZ) fr_depth = r.x;
        frC = ent_row(eC, fr_id); frD = ent_row(eD, fr_id); frF = ent_row(eF, fr_id);
    };
    // blend my window's front where `popping` holds (reference blend_one, resorted_render.cuh:74-119) and advance the ring's head
    auto pop_front = [&](const bool popping) __attribute__((always_inline)) {
        const float alpha0 = fminf(0.99f, frD.w * exp_blend(blend_power(frC.y - (float)px, frC.z - (float)py, frD)));
        const float test_T = fp.T * (1.0f - alpha0);
        const bool upd = popping && !(test_T < T_THRESHOLD);
        const float wgt = upd ? alpha0 * fp.T : 0.0f;
        fp.C[0] = fmaf(wgt, frF.x, fp.C[0]); fp.C[1] = fmaf(wgt, frF.y, fp.C[1]); fp.C[2] = fmaf(wgt, frF.z, fp.C[2]);
        if constexpr (DEPTHVIZ) depth_acc += upd ? fr_depth * alpha0 * fp.T : 0.0f; // reference resorted_render.cuh:107
        fp.T = upd ? test_T : fp.T;
        if constexpr (RECORD) log_append(upd, fr_id);
        else contrib = (popping && !upd) ? cfull : contrib;
        active = active && (upd || !popping); // a saturated pixel retires
        rh = wrap(rh + (popping ? 1 : 0));
        rn -= popping ? 1 : 0;
        back_d = rn == 0 ? -FLT_MAX : back_d;
        back2_d = rn <= 1 ? -FLT_MAX : back2_d;
    };
    // insert (depth, cid) into my window where `ins` holds; returns the logical index it took
    auto ring_insert = [&](const bool ins, const float depth, const int cid) __attribute__((always_inline)) -> int {
        int j = rn;                       // the logical index the candidate takes: behind everything, for a start
        int p = wrap(rh + rn);            // ... and its slot
        const bool mv1 = ins && depth < back_d;   // the last entry is deeper than the candidate (strictly: a new entry goes behind its equals): it moves up
        const bool mv2 = mv1 && depth < back2_d;  // ... and so is the one in front of it
        bool tie = mv2 && back_d == back2_d;      // two displaced entries of equal depth: the reference's swap loop leaves them in another order
        if (__builtin_amdgcn_ballot_w64(mv1) != 0ull) {
            if (mv1) { wr(p, back_d, back_i); p = prev_slot(p); j--; }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mv2) != 0ull, 0)) { // one step in ten
                float prev_moved = back2_d;
                bool mv = mv2;
                if (mv) { wr(p, back2_d, back2_i); p = prev_slot(p); j--; }
                float cur_d = -FLT_MAX; int cur_i = 0;
                mv = mv && j > 0;
                if (mv) { const float2 r = rd(prev_slot(p)); cur_d = r.x; cur_i = __float_as_int(r.y); mv = depth < cur_d; }
                while (__builtin_amdgcn_ballot_w64(mv) != 0ull) {
                    if (mv) {
                        wr(p, cur_d, cur_i);
                        tie = tie || cur_d == prev_moved;
                        prev_moved = cur_d;
                        p = prev_slot(p);
                        j--;
                        mv = j > 0;
                        if (mv) { const float2 r = rd(prev_slot(p)); cur_d = r.x; cur_i = __float_as_int(r.y); mv = depth < cur_d; }
                    }
                }
            }
        }
        if (ins) {
            wr(p, depth, cid);
            // the window's last two entries afterwards: (back, candidate) when nothing moved, (candidate, back) when the last entry did, unchanged otherwise
            const bool none = !mv1, one = mv1 && !mv2;
            back2_d = none ? back_d : one ? depth : back2_d;
            back2_i = none ? back_i : one ? cid : back2_i;
            back_d = none ? depth : back_d;
            back_i = none ? cid : back_i;
            rn++;
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tie) != 0ull, 0)) {
            // reference resorted_render.cuh:187-196: the carried entry swaps only with strictly deeper ones, so it travels past its equals, which
            // keep their slots -- among the displaced entries [j + 1, rn) every run of equal depths ends up rotated left by one
            if (tie) {
                int k = j + 1;
                while (k < rn) {
                    const float2 rk = rd(wrap(rh + k));
                    int e = k;
                    while (e + 1 < rn && rd(wrap(rh + e + 1)).x == rk.x) e++;
                    if (e > k) {
                        for (int u = k; u < e; u++) wr(wrap(rh + u), rk.x, __float_as_int(rd(wrap(rh + u + 1)).y));
                        wr(wrap(rh + e), rk.x, __float_as_int(rk.y));
                    }
                    k = e + 1;
                }
                back_i = __float_as_int(rd(wrap(rh + rn - 1)).y);
                back2_i = __float_as_int(rd(wrap(rh + rn - 2)).y); // (a tie displaced at least two entries: rn >= 3)
            }
        }
        return j;
    };

    // four candidates per quad, lane q brings candidate q (list position, -1 = none)
    auto feed4_from = [&](const int fid) __attribute__((always_inline)) {
        const int pf = min(max(fid, 0), list_last);
        float4 eAq = ent_row(eA, pf), eBq = ent_row(eB, pf), eCq = ent_row(eC, pf), eDq = ent_row(eD, pf);
#define STP_KB_FEED(I)                                                                                                  \
    {                                                                                                                   \
        const int cid = quad_bcast_i<I>(pf);                                                                            \
        if constexpr (I == 0) {                                                                                         \
            eDq.w = fid < 0 ? 0.0f : eDq.w; /* no candidate: alpha 0 fails the tests */                                 \
            dpp_hazard_guard_on(eDq.w);                                                                                 \
        } else dpp_hazard_guard();                                                                                      \
        const float depth = depth_along_ray_quad_ent<I, FRCP>(eAq, eBq, eCq, pix_dir);                                  \
        const float dx = quad_sub<I>(eCq.y, (float)px), dy = quad_sub<I>(eCq.z, (float)py);                             \
        const float power = blend_power_quad<I>(dx, dy, eDq);                                                           \
        const float alpha = min_099(quad_mul<I>(eDq.w, exp_blend(power)));                                              \
        const bool pass = active && !(depth < 0.0f) && !(power > 0.0f) && !(alpha < ALPHA_THRESHOLD);                   \
        pop_front(pass && rn == WIN);   /* a full window gives up its front before the candidate goes in */             \
        wave_sync();                                                                                                    \
        fetch_front();                  /* the front behind it: needed at the next pop, a candidate step from now */    \
        const bool ins = pass && active; /* (a pixel that saturated at that pop is done) */                             \
        const int at = ring_insert(ins, depth, cid);                                                                    \
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ins && at == 0) != 0ull, 0)) { /* the candidate IS the front now (a window that was empty, mostly) */ \
            wave_sync();                                                                                                \
            fetch_front();                                                                                              \
        }                                                                                                               \
        if constexpr (!RECORD) cfull = ins ? cid + 1 : cfull;                                                           \
    }
        STP_KB_FEED(0) STP_KB_FEED(1) STP_KB_FEED(2) STP_KB_FEED(3)
#undef STP_KB_FEED
    };

    int* const hfifo = s_fifo + ((w * 4 + s) * 4 + m) * KBR_CAP;
    int hf_head = 0, hf_cnt = 0; // (quad-uniform)
    auto fwrap = [&](int v) __attribute__((always_inline)) -> int { // v in [0, 3 KBR_CAP) -> [0, KBR_CAP)
        v -= v >= 2 * KBR_CAP ? 2 * KBR_CAP : 0;
        return v - (v >= KBR_CAP ? KBR_CAP : 0);
    };
    auto head_round = [&](const bool force) __attribute__((always_inline)) -> bool {
        const unsigned long long act = __ballot(active);
        const bool qlive = ((act >> (lane & ~3)) & 0xFull) != 0ull;
        if (!qlive) { hf_head = fwrap(hf_head + hf_cnt); hf_cnt = 0; }
        bool go;
        if (force) go = __any(hf_cnt > 0);
        else go = __any(hf_cnt > KBR_CAP - 16) || (__all(hf_cnt >= 4 || !qlive) && __any(hf_cnt >= 4));
        if (!go) return false;
        const int n = min(hf_cnt, 4);
        wave_sync();
        const int fid = q < n ? hfifo[fwrap(hf_head + q)] : -1;
        hf_head = fwrap(hf_head + n);
        hf_cnt -= n;
        feed4_from(fid);
        return true;
    };
    auto head_rounds = [&](const bool force) __attribute__((always_inline)) {
#pragma unroll 1
        for (;;) {
            if (!head_round(force)) break;
        }
    };

    // ---- main loop: batches of 32 list entries (as in the kernel above) -----------------------------------------------
    const int half = lane >> 5, e = lane & 31;
    int* const stA = s_stage + (w * 4 + 2 * half) * 32;
    const int* const st_row = s_stage + (w * 4 + s) * 32;
    const float qxc = (float)(px - (q & 1)) + 0.5f, qyc = (float)(py - (q >> 1)) + 0.5f;
#ifndef STP_KB_SYNC
#define STP_KB_SYNC 0 // experiment: the four waves of the tile meet at a workgroup barrier every STP_KB_SYNC entries of the list (every wave the same number
                      // of times, also one that has left the loop): keeps their reads of the entry records inside one stretch of the list
#endif
    int sync_at = 0; // (list position of my next barrier)
#pragma unroll 1
    for (int base = 0; base < total; base += 32) {
        if (!__any(active)) break;
        if (STP_KB_SYNC && base >= sync_at) { __builtin_amdgcn_s_barrier(); sync_at += STP_KB_SYNC; }
        const int ep = base + e;
        bool keepA = false, keepB = false;
        if (ep < total) {
            const uint32_t mask = __float_as_uint(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(eF) + ((uint32_t)ep << 4) + 12));
            const uint32_t mine = mask >> (4 * w + 2 * half);
            keepA = (mine & 1u) != 0u;
            keepB = (mine & 2u) != 0u;
        }
        const unsigned long long balA = __ballot(keepA), balB = __ballot(keepB);
        const unsigned int mA = (unsigned int)(balA >> (32 * half)), mB = (unsigned int)(balB >> (32 * half));
        const unsigned int below = (1u << e) - 1u;
        wave_sync();
        if (keepA) stA[__popc(mA & below)] = ep;
        if (keepB) stA[32 + __popc(mB & below)] = ep;
        wave_sync();
        const int n_s = __popc((unsigned int)(((s & 1) ? balB : balA) >> (32 * (s >> 1))));
        int n_max = n_s;
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) n_max = max(n_max, __shfl_xor(n_max, o));
#pragma unroll 1
        for (int g0 = 0; g0 < n_max; g0 += 16) {
#pragma unroll 1
            for (int g = g0; g < min(g0 + 16, n_max); g += 4) {
                const int i = g + q;
                int fid = -1;
                if (i < n_s) fid = st_row[i];
                bool keep = false;
                const unsigned long long act = __ballot(active);
                const bool qlive = ((act >> (lane & ~3)) & 0xFull) != 0ull;
                if (fid >= 0 && qlive) keep = kb_quad_can_blend(ent_row(eC, fid), ent_row(eD, fid), qxc, qyc);
                int bits = keep ? (1 << q) : 0;
                bits += __builtin_amdgcn_mov_dpp(bits, 0xB1, 0xF, 0xF, true);
                bits += __builtin_amdgcn_mov_dpp(bits, 0x4E, 0xF, 0xF, true);
                if (keep) hfifo[fwrap(hf_head + hf_cnt + __popc(bits & ((1 << q) - 1)))] = fid;
                hf_cnt += __popc(bits);
            }
            head_rounds(false);
        }
    }
    if (STP_KB_SYNC) for (; sync_at < total; sync_at += STP_KB_SYNC) __builtin_amdgcn_s_barrier(); // (the barriers I did not reach: my siblings count on them)
    head_rounds(true);
    // drain: what is left in the window, front first.  Only a pop of a FULL window can be the reference's pop "in front of the next entry"
    if (rn != WIN) cfull = total;
#pragma unroll 1
    for (int it = 0; it < WIN; it++) {
        if (!__any(active && rn > 0)) break;
        pop_front(active && rn > 0);
        wave_sync();
        fetch_front();
        cfull = total;
    }

    if constexpr (RECORD) log_flush();
    if (inside) {
        const size_t N = (size_t)a.W * a.H, pid = (size_t)a.W * py + px;
        a.final_T[pid] = fp.T;
        a.n_contrib[pid] = RECORD ? (uint32_t)log_records() : (uint32_t)contrib;
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
        if (log_records() > a.log_depth || total > LOG_MAX_LIST) a.tile_flags[tile] = 1u;
        report_log_need(a.log_need, log_records(), a.log_tag);
    }
}

template <int WIN, int MODE> hipError_t launch_kb_ring(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    const dim3 grid(f.gx * (f.ty1 - f.ty0)), block(256);
    constexpr size_t lds = kb_ring_lds<WIN>();
    if (f.wild_cov) hipLaunchKernelGGL((render_kbuffer_ring_kernel<WIN, MODE, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((render_kbuffer_ring_kernel<WIN, MODE, true>), grid, block, lds, st, a);
    return hipGetLastError();
}

template <int WIN, int MODE> hipError_t launch_kb_win(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    const dim3 grid(f.gx * (f.ty1 - f.ty0)), block(256);
    if (f.wild_cov) hipLaunchKernelGGL((render_kbuffer_wave_kernel<WIN, MODE, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((render_kbuffer_wave_kernel<WIN, MODE, true>), grid, block, 0, st, a);
    return hipGetLastError();
}

} // namespace

// mode: 0 forward, 2 recording forward, 3 depth visualisation.  Every window size of the reference's ladder (forward.cu:409-425)
// has a kernel here: four waves per SIMD up to 4 entries, three up to 16, two for 20 and 24 (164-187 VGPRs, no scratch)
hipError_t launch_kbuffer_wave(int mode, const FrameParams& f, const RenderArgs& a, hipStream_t st, bool* handled)
{
    const int w = f.s.queue_per_pixel; // reference forward.cu:409-425: the next supported window
    *handled = true;
    // windows of 8 .. 16 entries: the ring-in-LDS kernel (STP_KBUFFER=wave keeps the register window for them too)
    static const char* const kb_env = std::getenv("STP_KBUFFER");
    static const bool ring = !STP_LOG_PACK && !(kb_env && std::strcmp(kb_env, "wave") == 0); // (the ring kernel writes the plain log layout only)
#define STP_KBW(WIN) return mode == KBW_RECORD ? launch_kb_win<WIN, KBW_RECORD>(f, a, st) : mode == KBW_DEPTH ? launch_kb_win<WIN, KBW_DEPTH>(f, a, st) : launch_kb_win<WIN, KBW_FWD>(f, a, st)
#define STP_KBR(WIN) return mode == KBW_RECORD ? launch_kb_ring<WIN, KBW_RECORD>(f, a, st) : mode == KBW_DEPTH ? launch_kb_ring<WIN, KBW_DEPTH>(f, a, st) : launch_kb_ring<WIN, KBW_FWD>(f, a, st)
    if (w <= 1) STP_KBW(1);
    if (w <= 2) STP_KBW(2);
    if (w <= 4) STP_KBW(4);
    if (ring) {
        if (w <= 8) STP_KBR(8);
        if (w <= 12) STP_KBR(12);
        if (w <= 16) STP_KBR(16);
        if (w <= 20) STP_KBR(20);
        STP_KBR(24);
    }
    if (w <= 8) STP_KBW(8);
    if (w <= 12) STP_KBW(12);
    if (w <= 16) STP_KBW(16);
    if (w <= 20) STP_KBW(20);
    STP_KBW(24);
#undef STP_KBW
#undef STP_KBR
}

} // namespace stp
