// stp_render_replay.hip -- backward of the per-pixel-sort modes by REPLAYING the forward's blend log.
//
// No counterpart in the reference: its backward (hierarchical_render.cuh:1038-1175) re-runs the complete
// three-level resort to rediscover the order in which every pixel blended its Gaussians.  MI355X has 288 GB of
// HBM, so the training forward (render_hier_kernel<..., MODE_FWD_RECORD>, render_kbuffer_kernel<WIN, KB_FWD_RECORD>)
// simply writes that order down -- 2 bytes (the tile-list position) per blended (pixel, Gaussian) pair,
// as many records per pixel as the frames before it needed (RenderArgs::log_depth; 192 + eight spare rows = 400 B per pixel for a frame nothing
// is known about, 304 B per pixel = 0.63 GB at 1080p once C2's 114 blends per pixel are known) -- and this kernel walks each pixel's log
// front to back.  The gradient maths per pair is the reference's (blend_backward_terms); the result is the same sum
// in a different order.  Tiles whose log overflowed (a pixel with more than BLEND_LOG_DEPTH blended entries, a list longer than
// 65535) are flagged by the forward and left to the re-sorting backward kernels, which then run only on those tiles.
//
// Layout: one 256-thread workgroup per tile, thread -> pixel mapping identical to the forward (wave = row of four
// 4x4 sub-tiles), log laid out as the forward's mode wrote it (stp_blend.h): [tile][wave][k][lane] in hierarchical mode (the 64 lanes of a wave
// read record k with one 128-byte load as long as they walk in step), [tile][wave][k / 4][lane][k % 4] in k-buffer mode (a wave's load of "record
// k of every lane" touches the four lines of one 512-byte block, which the next three steps find in the L1).  The nine gradient terms of a blend are summed on chip as 64-bit fixed point (see
// stp_render_hier.inc for why not fp32 LDS atomics) in ONE set of sums per LIST POSITION, shared by the workgroup:
// acc[term][position - window start], 512 positions = 36 KB.  A tile whose list fits (all of C2-full) is one window:
// a blend is nine ds_add_u64 and nothing else, the sums leave the chip once at the end (16-lane group = one position,
// nine lanes = nine sums, one atomic instruction into the Gaussian's 64-byte gradient record).  Longer lists are
// walked with a window that slides in half steps (position p lives in slot p mod 512): a lane pauses at its first
// record beyond the window, and when every lane has left the window's lower half the workgroup meets at a barrier,
// writes that half out and moves on -- every pixel visits the list in (nearly) increasing position, so only the
// few records that the re-sort moved behind the window fall back to global atomics.  Before the LDS adds,
// lanes that hold the same position merge their terms pairwise with DPP (the adds serialise on equal addresses).
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

#ifdef STP_REPLAY_STATS
__device__ unsigned long long g_replay_stats[16];
#endif

namespace {

#ifndef STP_REPLAY_PAIRMERGE
#define STP_REPLAY_PAIRMERGE 1 // merge levels: 1 = inside 2x2 quads (lane^1, lane^2), 2 = + mirror in the 8-lane half, 3 = + mirror in the 16-lane row
                               // (C2-full, lanes in step: 1.38 ms without, 0.99 / 0.95 / 0.94 ms with 1 / 2 / 3; de-phased lanes, see
                               // STP_REPLAY_DEPHASE: 1.01 ms without, 0.92 / 0.95 with 1 / 2)
#endif
#ifndef STP_REPLAY_F64
#define STP_REPLAY_F64 0 // 1: on-chip sums as doubles through ds_add_f64 instead of 64-bit fixed point (one conversion per term instead of
                         // four instructions, no range check).  MEASURED SLOWER on MI355X: ds_add_f64 costs 9 cycles per wave
                         // instruction on distinct addresses like ds_add_u64, but 42 / 181 cycles when 4 / 16 lanes share an address
                         // (u64: 27 / 119; tools/lds_atomic_bench.hip), and this kernel lives on shared addresses: C2-full replay
                         // 0.98 -> 1.36 ms.  Kept as a switch for the record.
#endif
#ifndef STP_REPLAY_OCC
#define STP_REPLAY_OCC 4
#endif
#ifndef STP_REPLAY_COPIES
#define STP_REPLAY_COPIES 1 // 2: two sets of sums per workgroup, one for sub-tile rows 0-1 and one for rows 2-3 of every wave -- the round-2 verdict's
                            // experiment against same-address serialisation of the LDS adds.  MEASURED (round 3, C2-full, one box): 72 KB of LDS
                            // = two workgroups per CU 1.374 ms, windows of 256 positions at four workgroups per CU 1.155 ms, against 0.946 ms:
                            // the kernel lives on its four waves per SIMD (latency), and an average C2 list (321 entries) needs the 512 window.
#endif
#ifndef STP_REPLAY_COLOR32
#define STP_REPLAY_COLOR32 0 // 1: the three colour sums as 32-bit fixed point (ds_add_u32 costs half of ds_add_u64, also on shared addresses), the
                             // six geometric ones stay 64-bit.  MEASURED (round 3, one box, alternating): C2-full 0.947 against 0.928 ms,
                             // C2-min 1.056 against 1.008, C3 1.87 against 1.72 -- no faster anywhere; kept as a switch for the record.
#endif
#ifndef STP_REPLAY_WINDOW
#define STP_REPLAY_WINDOW 512
#endif
#ifndef STP_REPLAY_RAWADD
#define STP_REPLAY_RAWADD 0 // (MEASURED, round 4, off) The fixed-point conversion is fma(g, scale, 1.5 * 2^52): the double's low mantissa bits then hold round(g * scale) in two's
                            // complement, and rounds 1-3 subtracted the bit pattern of 1.5 * 2^52 before the LDS add -- a 64-bit integer subtraction,
                            // two half-rate VALU instructions per term, eighteen per blend.  1: the RAW bits are added.  Every add then carries the
                            // constant 0x4338 << 48 along, which only ever touches the sum's top 16 bits: with |sum| < 2^47 the low 48 bits ARE the
                            // sum in two's complement, and the flush sign-extends them (it never needs to know how many adds a slot received).
                            // The price is range: |q| < 2^39 per add (256 pixels per tile), so the scale is 2^27 / M instead of 2^31 / M and a term
                            // of 2^11 M or more takes the global-atomic path (2^20 M before); resolution 7.5e-9 M per add.
                            // MEASURED (one box, alternating, profiles/r04_replay_rawadd_ab.txt): 18 of a step's ~125 VALU instructions gone
                            // (isa_cost: 517 -> 437 SIMD cycles) and C2-full replay 0.856 against 0.859 ms -- nothing; C3 1.645 against 1.561 and
                            // C5 1.629 against 1.530 ms -- WORSE, their larger terms now miss the cap and go to global atomics.  The kernel
                            // is not bound by its VALU stream.
#endif
#ifndef STP_REPLAY_PACK2
#define STP_REPLAY_PACK2 0 // 1: the red and green colour sums share ONE 64-bit LDS add (two 32-bit fixed-point fields, scale 2^22 / M: the colour terms
                           // are bounded by 16 M by construction, their sums over a tile by 256 M) -- eight ds_add_u64 per blend instead of nine and
                           // two v_cvt_i32_f32 instead of two double conversions (round-3 verdict, item 2b).
#endif
#ifndef STP_REPLAY_ABLATE
#define STP_REPLAY_ABLATE 0 // timing experiments (results are WRONG): 1 = no LDS adds (conversions kept), 2 = no DPP merge levels, 3 = every entry
                            // record read from list position 0 (no gather), 4 = 1 + 3
#endif
#ifndef STP_REPLAY_FASTEXP
#define STP_REPLAY_FASTEXP 1 // the Gaussian weight of a replayed blend with a plain v_exp_f32 (see blend_terms)
#endif
#ifndef STP_REPLAY_FOLD
#define STP_REPLAY_FOLD 1 // constant factors of the geometric terms applied to the sums at the flush instead of to every pair (needs STP_REPLAY_STRAIGHT)
#endif
#ifndef STP_REPLAY_STRAIGHT
#define STP_REPLAY_STRAIGHT 1 // the gradient terms of a step as straight-line code (see blend_terms); 0: the branchy form of rounds 1-3
#endif
// (checked BELOW the defaults: a -DSTP_REPLAY_F64=1 build must say -DSTP_REPLAY_FOLD=0 too -- its flush writes the sums without the folded factors)
#if STP_REPLAY_FOLD && (STP_REPLAY_F64 || STP_REPLAY_COLOR32 || !STP_REPLAY_STRAIGHT)
#error "STP_REPLAY_FOLD is written for the 64-bit fixed-point sums of the straight-line step"
#endif
#if STP_REPLAY_COLOR32 && STP_REPLAY_F64
#error "STP_REPLAY_F64 keeps all nine sums as doubles"
#endif
constexpr int WINDOW = STP_REPLAY_WINDOW; // list positions per window (9 x 512 x 8 B = 36 KB of LDS: four workgroups per CU)
constexpr int EXHAUSTED = 0x7fffffff; // "position" of a lane that has no record left

__device__ __forceinline__ int replay_remap_tile(int wg, int n_wg)
{
    const int q = n_wg >> 3, r = n_wg & 7;
    const int xcd = wg & 7, k = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// (One kernel for both kinds of tile: as two launches the mixed case -- C2-min -- loses more to the half-empty grids
// than the lean loop gains.)
// LOG_BLOCKED: the forward's log layout (stp_blend.h: rows in hierarchical mode, blocked in k-buffer mode)
template <bool LOG_BLOCKED>
__global__ void __launch_bounds__(256, STP_REPLAY_OCC) render_replay_kernel(const RenderArgs a)
{
#if STP_REPLAY_COLOR32
    __shared__ unsigned long long s_acc[6 * WINDOW]; // [term 3..8][position - window start]: the six geometric sums, 64-bit fixed point
    __shared__ unsigned int s_acc32[3 * WINDOW];     // [term 0..2][position - window start]: the three colour sums, 32-bit fixed point
    constexpr int ACC64_FIRST = 3;
#else
    __shared__ unsigned long long s_acc[STP_REPLAY_COPIES * 9 * WINDOW]; // [copy][term][position - window start]
    constexpr int ACC64_FIRST = 0;
#endif
    double* const s_accd = reinterpret_cast<double*>(s_acc); // STP_REPLAY_F64: the same sums as doubles (ds_add_f64)
    __shared__ float s_md[4];

    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const int s = lane >> 4, x = lane & 15, m = x >> 2, q = x & 3;
    const int rows = a.ty1 - a.ty0;
    const int t = a.tile_order ? (int)a.tile_order[blockIdx.x] : replay_remap_tile((int)blockIdx.x, a.gx * rows);
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx, tile = tile_y * a.gx + tile_x;
    if (a.tile_flags[tile] != 0u) return; // log overflow: the re-sorting backward takes this tile
    const uint2 range = a.ranges[tile];
    const int px = tile_x * TILE + 4 * s + 2 * (m & 1) + (q & 1), py = tile_y * TILE + 4 * w + 2 * (m >> 1) + (q >> 1);
    const bool inside = px < a.W && py < a.H;
    const int list_len = (int)(range.y - range.x);
    if (list_len <= 0) return;

    for (int i = (int)threadIdx.x; i < STP_REPLAY_COPIES * (9 - ACC64_FIRST) * WINDOW; i += 256) s_acc[i] = 0ull;
    const int acc_copy = (STP_REPLAY_COPIES == 2 ? (lane >> 5) : 0) * 9 * WINDOW; // (two copies: sub-tile rows 0-1 / 2-3 of the wave add to their own)
#if STP_REPLAY_COLOR32
    for (int i = (int)threadIdx.x; i < 3 * WINDOW; i += 256) s_acc32[i] = 0u;
#endif

    BwdPixel bp;
    init_bwd_pixel(bp, a, inside, px, py);
    // (STP_REPLAY_FOLD) channel sums of the pixel: final . dL, C . dL (running), -T_final (bg . dL)
    const float FD = fmaf(bp.final_color[2], bp.dL_dpix[2], fmaf(bp.final_color[1], bp.dL_dpix[1], bp.final_color[0] * bp.dL_dpix[0]));
    float CD = 0.0f;
    const float tfbg = -bp.T_final * bp.bg_dot;
    int n = inside ? (int)a.n_contrib[(size_t)a.W * py + px] : 0;
    float md = fmaxf(fmaxf(fabsf(bp.dL_dpix[0]), fabsf(bp.dL_dpix[1])), fabsf(bp.dL_dpix[2]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmaxf(md, __shfl_xor(md, off));
    // fixed-point scale of the sums (stp_render_hier.inc: "on-chip gradient window"): one for the workgroup
    if (lane == 0) s_md[w] = md;
    __syncthreads(); // (also: the accumulators are zeroed)
    md = fmaxf(fmaxf(s_md[0], s_md[1]), fmaxf(s_md[2], s_md[3]));
    int md_exp = 0;
    const bool md_ok = md > 0.0f && md < 3.0e38f;
    if (md_ok) (void)frexpf(md, &md_exp);
    constexpr int FX_BITS = STP_REPLAY_RAWADD ? 27 : 31, FX_CAP_BITS = STP_REPLAY_RAWADD ? 11 : 20;
    const double fx_scale = ldexp(1.0, FX_BITS - md_exp), fx_inv = ldexp(1.0, md_exp - FX_BITS);
    // factor of term k that the blend step leaves out (STP_REPLAY_FOLD): applied once per sum
    auto term_scale = [&](int k) __attribute__((always_inline)) -> float {
#if STP_REPLAY_FOLD && STP_REPLAY_STRAIGHT
        return k == 3 ? -0.5f * (float)a.W : k == 4 ? -0.5f * (float)a.H : (k >= 5 && k <= 7) ? -0.5f : 1.0f;
#else
        return 1.0f;
#endif
    };
    const double fx_inv_term = fx_inv * (double)term_scale(lane & 15); // (flush: lane & 15 is the term a lane writes back)
    const float fx_cap = (md_ok || md == 0.0f) ? ldexpf(1.0f, min(md_exp + FX_CAP_BITS, 126)) : 0.0f; // (a tile whose M is not finite: nothing fits, every term goes to memory)
    // Colour terms: |alpha T dL/dpixel| < M = 2^md_exp by construction (M >= max |dL/dpixel| of the tile), and a lane that the
    // DPP merge has loaded with its partners' terms carries at most 16 of them: round(t 2^22 / M) fits 27 bits, the sum over
    // the tile's 256 pixels 31 -- 32-bit LDS adds, which cost half of the 64-bit ones, also where lanes share an address
    // (tools/lds_atomic_bench.hip: 5.3 / 13.7 against 8.3 / 27.3 cycles at 1 / 4 lanes per address).  Resolution M 2^-23.
    const float fx_scale32 = ldexpf(1.0f, 22 - max(md_exp, -100)), fx_inv32 = ldexpf(1.0f, max(md_exp, -100) - 22);

    // Addressing: wave-uniform bases (SGPR pairs) + one 32-bit byte offset per load, so that the loop's loads are
    // `global_load ... v_off, s[base]` without 64-bit address arithmetic (v_lshl_add_u64 issues at half the rate of a
    // 32-bit add on gfx950, tools/valu_rate_bench.hip).
    const char* const log_wave = log_wave_slice(a.blend_log, tile, __builtin_amdgcn_readfirstlane(w), a.log_depth);
    // (record indices are clamped to the slice: log_last_row = the last record index that has storage -- with STP_LOG_UNCOND the spare block's --,
    // log_last_rec = the last one that can hold a record; a clamped read is readable garbage that is never used)
    const uint32_t log_last_row = (uint32_t)(a.log_depth + BLEND_LOG_SPARE - 1), log_last_rec = (uint32_t)(a.log_depth - 1);
    const uint32_t lane16 = (uint32_t)lane << LOG_PIECE_SHIFT;
    auto log_at = [&](uint32_t k) __attribute__((always_inline)) -> int { // record k of this lane
        return (int)*reinterpret_cast<const log_t*>(log_wave + log_record_offset<LOG_BLOCKED>(2u * k, lane16));
    };
    const float pxf = (float)px, pyf = (float)py;
    const float4* const eC = a.entC + range.x; // list-ordered entry records: mean + Gaussian id, conic + opacity, colour
    const float4* const eD = a.entD + range.x;
    const float4* const eF = a.entF + range.x;
    struct Entry { float4 c, d, f; };
    auto entry_at = [&](int p) __attribute__((always_inline)) { // (a harmless read of entry 0 where there is no record: no branch)
        const uint32_t off = (uint32_t)(p < list_len ? p : 0) << 4;
        auto at = [&](const float4* base) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off); };
        return Entry{at(eC), at(eD), at(eF)};
    };

    const uint32_t list_last = (uint32_t)(list_len - 1);
    auto entry_at_clamped = [&](uint32_t p) __attribute__((always_inline)) { // (p beyond the list -- a corrupt log word -- reads the last entry)
#if STP_REPLAY_ABLATE == 3 || STP_REPLAY_ABLATE == 4
        const uint32_t off = min(p, 0u) << 4;
#else
        const uint32_t off = min(p, list_last) << 4;
#endif
        auto at = [&](const float4* base) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off); };
        return Entry{at(eC), at(eD), at(eF)};
    };

    // the gradient terms of one record (reference maths); false = nothing to add (no record, or the pixel saturates here)
    auto blend_terms = [&](bool act, const Entry& cur, float (&g)[9]) __attribute__((always_inline)) -> bool {
#if STP_REPLAY_STRAIGHT
        // Straight-line form: every lane evaluates its (possibly stand-in) entry; a lane without a record, or whose pixel saturates
        // here, is switched off through its FACTORS -- all nine terms are linear in (T, T_final), so T := 0 and T_final := 0 make them
        // exact zeros (every other factor is finite: alpha <= 0.99, test_T >= 1e-6 where it is used, the stand-in is entry 0 of the
        // list) -- three selects instead of two branches and eighteen zeroing moves per step.
        {
            const float4 co = cur.d;
            const float dx = cur.c.y - pxf, dy = cur.c.z - pyf;
#if STP_REPLAY_FASTEXP
            // the exponent with contracted products (7 instructions for the forward's 9) and 2^(x log2 e) without the forward's first-order
            // correction of the product's rounding (2 for 6): |relative error of G| < 3e-7 for exponents above -5.6, where a blend can be.
            // The blend SET is the log's; G only weighs gradient terms here, which are compared with a tolerance anyway.
            const float e2 = fmaf(co.y * dx, dy, 0.5f * fmaf(co.z * dy, dy, co.x * dx * dx));
            const float G = __builtin_amdgcn_exp2f(fmaxf(e2, 0.0f) * -1.44269502162933349609375f);
#else
            const float G = exp_blend(fminf(blend_power(dx, dy, co), 0.0f)); // (a recorded blend has power <= 0: the clamp only keeps a stand-in's G finite)
#endif
            const float alpha = fminf(0.99f, co.w * G);
            const float test_T = bp.T * (1.0f - alpha);
            const bool ok = act && !(test_T < T_THRESHOLD);
#if STP_REPLAY_FOLD
            // dL/dalpha = sum_ch (c_ch - (final_ch - C_ch) / test_T) dL_ch  with the channel sums taken first: cd = c . dL, FD = final . dL (a
            // constant of the pixel), CD = C . dL (a running scalar, CD += alpha T cd) -- six instructions instead of fifteen, and one
            // accumulated scalar instead of three colours; 1 / (1 - alpha) = T / test_T costs a multiply instead of a second reciprocal
            const float Tm = ok ? bp.T : 0.0f, tfbgm = ok ? tfbg : 0.0f;
            const float dchannel_dcolor = alpha * Tm;
            const float rcp_test_T = __builtin_amdgcn_rcpf(test_T);
            const float rcp_1ma = rcp_test_T * bp.T;
            const float cd = fmaf(cur.f.z, bp.dL_dpix[2], fmaf(cur.f.y, bp.dL_dpix[1], cur.f.x * bp.dL_dpix[0]));
            CD = fmaf(dchannel_dcolor, cd, CD);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) g[ch] = dchannel_dcolor * bp.dL_dpix[ch];
            float dL_dalpha = fmaf(-rcp_test_T, FD - CD, cd) * Tm;
            dL_dalpha = fmaf(tfbgm, rcp_1ma, dL_dalpha);
#else
            const float Tm = ok ? bp.T : 0.0f, tfm = ok ? bp.T_final : 0.0f;
            const float dchannel_dcolor = alpha * Tm;
            const float rcp_test_T = __builtin_amdgcn_rcpf(test_T);
            const float rcp_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
            const float col[3] = {cur.f.x, cur.f.y, cur.f.z};
            float dL_dalpha = 0.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                bp.C[ch] += col[ch] * alpha * Tm;
                const float accum_rec = (bp.final_color[ch] - bp.C[ch]) * rcp_test_T;
                dL_dalpha += (col[ch] - accum_rec) * bp.dL_dpix[ch];
                g[ch] = dchannel_dcolor * bp.dL_dpix[ch];
            }
            dL_dalpha *= Tm;
            dL_dalpha += (-tfm * rcp_1ma) * bp.bg_dot;
#endif
            const float dL_dG = co.w * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
#if STP_REPLAY_FOLD
            // the frame's constant factors of the five geometric terms (-W/2, -H/2, -1/2 three times) are applied to the SUMS when they
            // leave the chip (term_scale), not to every pair: with u = G dx dL/dG, v = G dy dL/dG the five terms are eleven instructions
            const float u = gdx * dL_dG, v = gdy * dL_dG;
            g[3] = fmaf(v, co.y, u * co.x);
            g[4] = fmaf(u, co.y, v * co.z);
            g[5] = u * dx;
            g[6] = u * dy;
            g[7] = v * dy;
#else
            const float dG_ddelx = -gdx * co.x - gdy * co.y;
            const float dG_ddely = -gdy * co.z - gdx * co.y;
            g[3] = dL_dG * dG_ddelx * (0.5f * (float)a.W);
            g[4] = dL_dG * dG_ddely * (0.5f * (float)a.H);
            g[5] = -0.5f * gdx * dx * dL_dG;
            g[6] = -0.5f * gdx * dy * dL_dG;
            g[7] = -0.5f * gdy * dy * dL_dG;
#endif
            g[8] = G * dL_dalpha;
            bp.T = ok ? test_T : bp.T;
            return ok;
        }
#endif
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
        return ok;
    };
    // merge lanes on the same position, then add to the window's sums (lo = first position of the window)
    // deep (wave-uniform): also the two mirror levels inside the 16-lane row
    // one_window (a literal at both call sites): the list fits the window -- no position lies in front of it, a position IS its slot
    unsigned long long ablate_sink = 0ull;
    auto merge_and_add = [&](bool ok, int cur_pos, int cur_id, float (&g)[9], int lo, const bool deep, const bool one_window) __attribute__((always_inline)) {
#if STP_REPLAY_PAIRMERGE && STP_REPLAY_ABLATE != 2
        // Pairwise merge (DPP): a lane and its partner -- lane^1, lane^2, then the mirror lanes of its 8-lane half and
        // row -- that hold the same list position sum their terms in registers and only one of them goes to LDS.  Per
        // step 55 lanes blend on 17.5 distinct positions (C2); the LDS atomics serialise on equal addresses and were the
        // limiter (1.38 ms), the VALU had headroom: 0.94 ms.  The partner's value enters as the DPP operand of one
        // v_fmac per term.  (A pre-reduction that needs a whole quad on one position fires for one quad in ten.)
        {
            int key = ok ? cur_pos : -2 - lane; // unique when not blending
#define STP_MERGE_LEVEL(CTRL, LOWER)                                                                                    \
            {                                                                                                       \
                const int pk = __builtin_amdgcn_mov_dpp(key, CTRL, 0xF, 0xF, true);                                 \
                const bool match = pk == key;                                                                       \
                const float mf = (match && (LOWER)) ? 1.0f : 0.0f;                                                  \
                dpp_hazard_guard(); /* g[] was written by ordinary VALU instructions a moment ago */                \
                _Pragma("unroll") for (int kk = 0; kk < 9; kk++) g[kk] = partner_fma<CTRL>(g[kk], mf, g[kk]);       \
                if (match && !(LOWER)) key = -2 - lane; /* the upper lane of a matching pair has handed its terms over */ \
            }
            STP_MERGE_LEVEL(0xB1, (q & 1) == 0) // partner lane ^ 1 (quad_perm [1,0,3,2])
            STP_MERGE_LEVEL(0x4E, (q & 2) == 0) // partner lane ^ 2 (quad_perm [2,3,0,1])
            if (deep) {
                STP_MERGE_LEVEL(0x141, (x & 7) < 4)  // partner 7 - i inside each 8-lane half (row_half_mirror)
                STP_MERGE_LEVEL(0x140, x < 8)        // partner 15 - i inside the 16-lane row (row_mirror)
            }
#undef STP_MERGE_LEVEL
            ok = key >= 0; // (a lane that blends holds a list position, every other one its negative stand-in: no separate flag to carry through the levels)
        }
#endif
#ifdef STP_REPLAY_STATS
        {
            const int nw = __popcll(__ballot(ok)), ns = __popcll(__ballot(ok && cur_pos < lo));
            // lanes whose position a LOWER adding lane also holds (same row / another row): what the LDS serialises
            bool dup_row = false, dup_any = false;
            int mult = 1;
            for (int l = 0; l < 64; l++) {
                const int pl = __shfl(ok ? cur_pos : -1 - lane, l);
                if (ok && pl == cur_pos && l != lane) { mult++; if (l < lane) { dup_any = true; if ((l >> 4) == (lane >> 4)) dup_row = true; } }
            }
            const int n_dup = __popcll(__ballot(ok && dup_any)), n_dup_row = __popcll(__ballot(ok && dup_row));
            int mmax = ok ? mult : 0;
            for (int off = 32; off > 0; off >>= 1) mmax = max(mmax, __shfl_xor(mmax, off));
            if (lane == 0) { atomicAdd(&g_replay_stats[0], 1ull); atomicAdd(&g_replay_stats[2], (unsigned long long)nw); atomicAdd(&g_replay_stats[3], (unsigned long long)ns);
                             atomicAdd(&g_replay_stats[4], (unsigned long long)n_dup); atomicAdd(&g_replay_stats[5], (unsigned long long)n_dup_row); atomicAdd(&g_replay_stats[6], (unsigned long long)mmax); }
        }
#endif
        if (ok) {
#if STP_REPLAY_F64
            if (cur_pos >= lo) { // nine ds_add_f64, nothing else (one v_cvt per term; no range check, no fixed-point scale)
#pragma unroll
                for (int kk = 0; kk < 9; kk++) atomicAdd(&s_accd[kk * WINDOW + (cur_pos & (WINDOW - 1))], (double)g[kk]);
            } else {
#else
            // (the three colour terms are alpha T dL/dpixel: below M each, below 16 M after the merge levels -- they cannot reach the
            // fixed point's cap of 2^20 M and stay out of the range check)
            float gmax = fabsf(g[3]);
#pragma unroll
            for (int kk = 4; kk < 9; kk++) gmax = fmaxf(gmax, fabsf(g[kk]));
            const int slot = one_window ? cur_pos : (cur_pos & (WINDOW - 1));
            if ((one_window || cur_pos >= lo) && gmax < fx_cap) { // nine adds, nothing else
#if STP_REPLAY_COLOR32
#pragma unroll
                for (int kk = 0; kk < 3; kk++) atomicAdd(&s_acc32[kk * WINDOW + (cur_pos & (WINDOW - 1))], (unsigned int)__float2int_rn(g[kk] * fx_scale32));
#endif
#if STP_REPLAY_PACK2
                {   // red | green: sum(q1) * 2^32 + sum(q0), both signed -- the high field takes the low field's borrow along
                    const int q0 = __float2int_rn(g[0] * fx_scale32), q1 = __float2int_rn(g[1] * fx_scale32);
                    const unsigned long long packed = ((unsigned long long)(unsigned int)(q1 + (q0 >> 31)) << 32) | (unsigned long long)(unsigned int)q0;
                    atomicAdd(&s_acc[acc_copy + slot], packed);
                }
#endif
#pragma unroll
                for (int kk = (STP_REPLAY_PACK2 ? 2 : ACC64_FIRST); kk < 9; kk++) {
                    // round-to-nearest integer of g*scale through the 1.5*2^52 trick (|g*scale| < 2^51 + margin)
                    const double tq = fma((double)g[kk], fx_scale, 6755399441055744.0);
#if STP_REPLAY_RAWADD
                    atomicAdd(&s_acc[acc_copy + (kk - ACC64_FIRST) * WINDOW + slot], (unsigned long long)__double_as_longlong(tq));
#else
                    const long long qv = __double_as_longlong(tq) - 0x4338000000000000ll;
#if STP_REPLAY_ABLATE == 1 || STP_REPLAY_ABLATE == 4
                    ablate_sink ^= (unsigned long long)qv + (unsigned long long)slot;
#else
                    atomicAdd(&s_acc[acc_copy + (kk - ACC64_FIRST) * WINDOW + slot], (unsigned long long)qv);
#endif
#endif
                }
            } else { // a record the re-sort moved across a window boundary, or a term too large for the fixed point
#endif
#pragma unroll
                for (int kk = 0; kk < 9; kk++) atomicAdd(grad_slot(a, cur_id, kk), g[kk] * term_scale(kk));
            }
        }
    };
    // the window's sums leave the chip: 16-lane group = one position, nine lanes = its nine sums, one atomic
    // instruction (one request) into the Gaussian's 64-byte gradient record
    // (positions f0 .. f1 - 1, at most WINDOW of them; position p lives in slot p mod WINDOW)
    auto flush_range = [&](int f0, int f1) __attribute__((always_inline)) {
        __syncthreads();
        const int term = lane & 15;
        for (int pp = f0 + (int)(threadIdx.x >> 4); pp < f1; pp += 16) {
            const int p = pp & (WINDOW - 1);
            if (term < 9) {
#if STP_REPLAY_F64
                const double v = s_accd[term * WINDOW + p];
                if (v != 0.0) {
                    s_accd[term * WINDOW + p] = 0.0;
                    atomicAdd(grad_slot(a, __float_as_int(eC[pp].w), term), (float)v);
                }
#else
#if STP_REPLAY_COLOR32
                if (term < 3) {
                    const int v32 = (int)s_acc32[term * WINDOW + p];
                    if (v32 != 0) {
                        s_acc32[term * WINDOW + p] = 0u;
                        atomicAdd(grad_slot(a, __float_as_int(eC[pp].w), term), (float)v32 * fx_inv32);
                    }
                    continue;
                }
#endif
#if STP_REPLAY_PACK2
                if (term < 2) { // the two colour fields of one word (both lanes read it in this instruction; lane 0 clears it afterwards)
                    const long long w = (long long)s_acc[p];
                    const int lo = (int)(unsigned int)(unsigned long long)w;
                    const int field = term == 0 ? lo : (int)((w - (long long)lo) >> 32);
                    if (term == 0 && w != 0) s_acc[p] = 0ull;
                    if (field != 0) atomicAdd(grad_slot(a, __float_as_int(eC[pp].w), term), (float)field * fx_inv32);
                    continue;
                }
#endif
                long long v = (long long)s_acc[(term - ACC64_FIRST) * WINDOW + p];
#if STP_REPLAY_RAWADD
                static_assert(STP_REPLAY_COPIES == 1, "the raw-bits sums are decoded per slot");
                if (v != 0) { s_acc[(term - ACC64_FIRST) * WINDOW + p] = 0ull; v = (long long)((unsigned long long)v << 16) >> 16; } // the low 48 bits, sign-extended
                if (v != 0) atomicAdd(grad_slot(a, __float_as_int(eC[pp].w), term), (float)((double)v * fx_inv_term));
                continue;
#endif
                if (STP_REPLAY_COPIES == 2) { v += (long long)s_acc[9 * WINDOW + (term - ACC64_FIRST) * WINDOW + p]; s_acc[9 * WINDOW + (term - ACC64_FIRST) * WINDOW + p] = 0ull; }
                if (v != 0) {
                    s_acc[(term - ACC64_FIRST) * WINDOW + p] = 0ull;
                    atomicAdd(grad_slot(a, __float_as_int(eC[pp].w), term), (float)((double)v * fx_inv_term));
                }
#endif
            }
        }
    };

    // Where the splats are larger than the wave's 16x4 pixels every lane blends the same entries whatever its phase: de-phasing
    // buys nothing there and the two mirror levels of the merge are what keeps the LDS adds apart (workload L1: 2.8 ms with
    // them, 3.1 ms without).  Decided once per wave: do most of its pixels START on the same entry?
    bool same_start;
    {
        const int p0 = n > 0 ? log_at(0) : 0x7fffffff;
        int pmin = p0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pmin = min(pmin, __shfl_xor(pmin, o));
        same_start = __popcll(__ballot(p0 == pmin && n > 0)) >= 40;
    }
#ifndef STP_REPLAY_HOIST
#define STP_REPLAY_HOIST 0 // 1: do not re-zero the terms of a lane that does not blend in a step (round 2).  Its partner in the DPP merge
                           // multiplies them by zero -- and 0 x Inf is NaN: a lane whose last blend overflowed would poison the sums of
                           // OTHER Gaussians.  Re-zeroing is nine full-rate v_mov per step: 0.951 against 0.929 ms on the final code of round 3
                           // (three alternating runs; the first measurement, beside another change, had shown no difference).  Zeroing only
                           // in the step in which a lane stops blending (ballot difference + branch) was built too: 1.046 ms -- de-phased
                           // lanes stop in different steps, the branch is taken most of the time and splits the loop body.  Safety wins.
#endif
    // (STP_REPLAY_HOIST: the terms of a lane that does not blend in a step are not zeroed -- they keep the lane's last,
    // finite, values; the merge multiplies such a partner by zero and the lane itself adds nothing)
    float g[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // Two dependent loads lead to a blend: log record (list position) -> the entry's record.  They are software
    // pipelined one step apart: `pos` / `en` hold the lane's next record and its entry, `pos1` the position of the one
    // after, each loaded an iteration before it is needed.
    if (list_len <= WINDOW) {
        // ---- the list fits one window (all of C2-full): the lanes walk their logs in step, record k in iteration k ----
        int nmax = n;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
#ifndef STP_REPLAY_DEPHASE
#define STP_REPLAY_DEPHASE 1
#endif
        // De-phasing.  The kernel is bound by the LDS adds, and those by lanes that hit one address in one instruction:
        // neighbouring pixels blend the same entries in the same order, so lanes that walk their logs in step sit on the
        // same list position all the time (measured, C2-full, after the four merge levels: 41 adding lanes on 17.5 distinct
        // positions, the largest group 6 lanes; tools/replay_stats.py).  Lane x of every 16-lane row therefore starts x
        // iterations late: 15 % more iterations, but 42 adding lanes on 24 positions with the quad merge alone (half the
        // merge's VALU work), and 0.98 -> 0.92 ms.  (Other patterns measured: by quad 0.96-1.01, by row and lane 1.07, all
        // 64 lanes apart 1.64 ms.)
        const bool dense = STP_REPLAY_DEPHASE == 0 || same_start; // (wave-uniform)
        const int off = dense ? 0 : x;
        if (STP_REPLAY_DEPHASE) {
            int nn = n + off;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nn = max(nn, __shfl_xor(nn, o));
            nmax = nn;
        }
        // Pipeline: `pos` = list position of my record of this step (-1: none), `raw1` = the log word of the record after it, read a step
        // ago and checked against the record count only now (a count can still shrink, see below); the entry of the NEXT step is
        // requested at the top of a step, the log word of the one after next right behind it -- unconditionally (a row of the log that
        // holds no record of mine is readable garbage): nothing in a step waits for a load issued in the same step.
        int pos = (0 < n && off == 0) ? log_at(0) : -1;
        int raw1 = log_at(min((uint32_t)max(1 - off, 0), log_last_rec));
        Entry en = entry_at(max(pos, 0));
#ifndef STP_REPLAY_UNROLL2
#define STP_REPLAY_UNROLL2 1 // two copies of the step, the entry registers alternating between them (no copy of the ten entry words at the back edge)
#endif
        // one step: blends `cur` (loaded an iteration ago), loads the entry of the next step into `nxt`
        auto one_step = [&](const int k, const Entry& cur, Entry& nxt) __attribute__((always_inline)) {
            const int kr = k - off; // my record index
            // (index checks as one unsigned compare each -- n >= 0; the entry offset clamped with one unsigned minimum; the log row of a
            // record index outside the log is clamped to the last row of my slice -- the spare row --, whose word is read and never used)
            const bool have = (uint32_t)kr < (uint32_t)n;
            const bool have1 = (uint32_t)(kr + 1) < (uint32_t)n;
            const int cur_pos = pos, cur_id = __float_as_int(cur.c.w);
            const int pos1 = have1 ? raw1 : -1;
            nxt = entry_at_clamped(have1 ? (uint32_t)raw1 : 0u);
            raw1 = log_at(min((uint32_t)(kr + 2), log_last_row));
            pos = pos1;
#if !STP_REPLAY_HOIST && !STP_REPLAY_STRAIGHT
            for (int kk = 0; kk < 9; kk++) g[kk] = 0.0f;
#endif
            const bool ok = blend_terms(have, cur, g);
            if (have && !ok) n = kr; // (an ulp of difference against the forward's transmittance: stop where it says so)
            merge_and_add(ok, cur_pos, cur_id, g, 0, dense, true);
        };
#if STP_REPLAY_UNROLL2
        Entry en2 = en;
#pragma unroll 1
        for (int k = 0; k < nmax; k += 2) {
            one_step(k, en, en2);
            if (k + 1 >= nmax) break;
            one_step(k + 1, en2, en);
        }
#else
        for (int k = 0; k < nmax; k++) {
            const Entry cur = en;
            one_step(k, cur, en);
        }
#endif
        flush_range(0, list_len);
        if (ablate_sink == 0x123456789abcull) a.grad_rec[0] = 1.0f; // (STP_REPLAY_ABLATE: keeps the conversions alive)
    } else {
    // ---- longer lists, window by window: every lane pauses at its first record beyond the window ----
    int k = 0; // records consumed by this lane
    int pos = (0 < n) ? log_at(0) : EXHAUSTED;
    int pos1 = (1 < n) ? log_at(1) : EXHAUSTED;
    Entry en = entry_at(pos);
#ifndef STP_REPLAY_DEPHASE_WIN
#define STP_REPLAY_DEPHASE_WIN 0
#endif
#ifndef STP_REPLAY_RING
#define STP_REPLAY_RING 1
#endif
    // De-phasing as in the one-window walk (lane x of every 16-lane row sits out the first x iterations) was built and MEASURED SLOWER
    // in round 3, twice: with hard windows, once per window (C2-min replay 1.015 -> 1.06 ms, C3 1.74 -> 2.08, C5 1.72 -> 1.97), and with
    // the sliding window, once at the start (C2-min 1.007 -> 1.016 ms, C3 1.65 -> 1.90, C5 1.65 -> 1.86; one box, alternating).  Kept as
    // a switch, off.
    const int dephase = (STP_REPLAY_DEPHASE_WIN && !same_start) ? x : 0;
    // The window slides in HALF steps (STP_REPLAY_RING): a phase covers positions [lo, lo + WINDOW), position p lives in slot p mod WINDOW,
    // and the phase ends when every lane has left its LOWER half -- lanes that are ahead keep blending in the upper half meanwhile and
    // stop only at lo + WINDOW.  Then the lower half's sums leave the chip and its slots become the next phase's upper half.  With hard
    // windows (STEP = WINDOW, round 2) every lane idled from its last record of a window until the slowest lane of the slowest wave had
    // finished it: the iterations of a wave were the SUM over the windows of the busiest pixel's records in each.
    // MEASURED (round 3, one box, alternating, -DSTP_REPLAY_RING=0 = hard windows): C3 replay 1.734 -> 1.645 ms, C5 1.714 -> 1.636,
    // C2-min 1.010 -> 1.005, L1 2.856 -> 2.836; C2-full (one window per tile) unchanged.
#ifndef STP_REPLAY_RING_DIV
#define STP_REPLAY_RING_DIV 2 // steps of WINDOW / 2.  Quarter / eighth steps (more, smaller flushes and barriers): C3 replay 1.655 -> 1.669 / 1.708 ms,
                              // C5 1.637 -> 1.694 / 1.752
#endif
    constexpr int STEP = STP_REPLAY_RING ? WINDOW / STP_REPLAY_RING_DIV : WINDOW;
    static_assert((WINDOW & (WINDOW - 1)) == 0, "slots are addressed by position mod WINDOW");
    int wait = dephase; // (sliding window: the lanes are not re-aligned at a phase's end, one offset at the start lasts)
    for (int lo = 0;; lo += STEP) {
        const int hi = lo + WINDOW;
        const bool last = hi >= list_len;                  // (workgroup-uniform)
        const int leave = last ? EXHAUSTED : lo + STEP;    // the phase is over when every lane's next record is at or beyond this position
        if (!STP_REPLAY_RING) wait = dephase;
        for (;;) {
            if (!__any(pos < leave)) break;
            const bool mine = pos < hi; // my next record belongs to this phase (or to an earlier one: a straggler)
            const bool act = mine && wait <= 0;
            wait--;
            const Entry cur = en;
            const int cur_pos = pos, cur_id = __float_as_int(cur.c.w);
            // issue the next round of loads before touching this step's data
            k += (int)act;
            const int rec = log_at(min((uint32_t)(k + 1), log_last_rec));
            pos = act ? pos1 : pos;
            pos1 = act ? (k + 1 < n ? rec : EXHAUSTED) : pos1;
            en = entry_at(pos);
#if !STP_REPLAY_HOIST && !STP_REPLAY_STRAIGHT
            for (int kk = 0; kk < 9; kk++) g[kk] = 0.0f;
#endif
            const bool ok = blend_terms(act, cur, g);
            if (act && !ok) { n = k; pos = EXHAUSTED; pos1 = EXHAUSTED; } // (saturated one record earlier than the forward said)
            merge_and_add(ok, cur_pos, cur_id, g, lo, same_start, false);
        }
        flush_range(lo, last ? list_len : lo + STEP);
        if (last) break;
        __syncthreads();
    }
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

int blend_log_rows(int depth) { return depth + BLEND_LOG_SPARE; }
int blend_log_default_depth() { return BLEND_LOG_DEPTH; }
int blend_log_clamp_depth(int d) // (a multiple of the block: a lane's records come in pieces)
{
    constexpr int Q = LOG_BLOCK < 8 ? 8 : LOG_BLOCK;
    static_assert(BLEND_LOG_DEPTH_MIN % Q == 0 && BLEND_LOG_DEPTH_MAX % Q == 0 && BLEND_LOG_DEPTH % Q == 0, "log depths are multiples of the block");
    d = d > BLEND_LOG_DEPTH_MAX ? BLEND_LOG_DEPTH_MAX : (d + Q - 1) / Q * Q;
    return d < BLEND_LOG_DEPTH_MIN ? BLEND_LOG_DEPTH_MIN : d;
}

hipError_t launch_hier_replay(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    if (log_blocked(f.s)) hipLaunchKernelGGL(render_replay_kernel<true>, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(render_replay_kernel<false>, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
