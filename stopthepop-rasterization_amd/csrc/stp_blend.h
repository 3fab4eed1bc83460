// stp_blend.h -- per-pixel blend state and the front-to-back gradient of one blended
// (pixel, Gaussian) pair, shared by the k-buffer and hierarchical backward kernels.
// Replaces the blend lambdas of reference hierarchical_render.cuh:1094-1166 and
// resorted_render.cuh:312-392 (identical maths in both).
#pragma once

#include "stp_device.h"

namespace stp {

// Pointers every render kernel needs (by value in the kernarg segment).
struct RenderArgs {
    int W, H, gx, ty0, ty1;
    const uint2* ranges;
    const uint32_t* point_list;
    const float2* means2D;
    const float4* conic_opacity;
    const float4* cov3D_inv;
    const float* features; // colours, P x 3
    const float4* entA;    // per-entry data in list order (BinningState::entA..entF), per-pixel-sort modes only
    const float4* entB;
    const float4* entC;
    const float4* entD;
    const float4* entF;
    const float* inv_vp;
    const float* cam;
    const float* bg;
    // forward outputs
    float* final_T;
    uint32_t* n_contrib;
    float* out_color;
    // backward inputs / outputs
    const float* pixel_colors;
    const float* dL_dpix;
    float* grad_rec;   // P x grad_stride floats: the nine sums of a Gaussian in one record (stp_raster.h, stp_backward)
    int grad_stride;   // floats per record: STP_GRAD_RECORD_FLOATS (one 64-byte line), or 9 = compact (tile-row sharding: what crosses xGMI)
    // blend log (training forward -> replay backward): per (tile, wave, k, lane) the list position of the k-th
    // entry that lane's pixel blended; tile_flags[tile] != 0 marks a tile whose log overflowed
    uint32_t* blend_log;   // (storage; the records are log_t)
    int log_depth;         // records per pixel this frame's log holds (host: log_depth_for; the backward gets the forward's value)
    uint32_t* log_need;    // recording forwards: where the frame's largest blend count per pixel is reported (report_log_need), or nullptr
    uint32_t log_tag;      // 16 bits that identify the frame's KIND in that word (the words are shared by the kinds that map to one guess slot)
    uint32_t* tile_flags;
    const uint32_t* tile_order; // nullptr, or workgroup j of the render kernels takes tile tile_order[j] of the frame's window: longest list first (tile_order_kernel)
    int flag_mode; // resorting backward: 0 = all tiles, 1 = only tiles with tile_flags != 0
    // debug depth visualisation (StpSettings::debug_visualization == STP_DEBUG_DEPTH): the forward kernels write
    // sum(depth * alpha * T) to channel 0 and T to channel 1 of out_color instead of the colour
    int debug_depth;
    const float* means3D; // GLOBAL mode's visualised depth is |cam - mean| (reference forward.cu:337-341)
};

// A log record is a 16-bit list position (measured on C2: 2-byte records cost the forward 0.05 ms less than 4-byte
// ones and halve the log); a tile whose list is longer than LOG_MAX_LIST is flagged like a log overflow.
typedef uint16_t log_t;
constexpr int LOG_MAX_LIST = 65535;
#ifndef STP_LOG_PACK
#define STP_LOG_PACK 0 // 1: two records per 32-bit store, layout [tile][wave][record / 2][lane] of u32 (measured in round 2, see profiles/EXPERIMENTS.md;
                       // written for the [record][lane] layout of rounds 1-5 and not carried over to the blocked layout)
#endif
#if STP_LOG_PACK
#error "STP_LOG_PACK addressed the [record][lane] log of rounds 1-5; the blocked layout (LOG_BLOCK) keeps a lane's records side by side already"
#endif
// Depth of the log = records per pixel it can hold (2 B each; + one spare row, STP_LOG_UNCOND): a RUN-TIME value since round 4
// (RenderArgs::log_depth), chosen per frame by the host from the blends per pixel the previous recording forwards of the same kind
// needed (stp_api.hip: log_depth_for) -- C2 blends at most 114 entries per pixel, C3 155, C5 195 (profiles/r03_log_depth_stats.txt), so
// one fixed depth either wastes memory or sends tiles to the re-sorting fallback, and ONE such tile costs 1.25 ms (profiles/r04_log_depth_ab.txt).
// BLEND_LOG_DEPTH is the depth of a frame nothing is known about yet; a pixel that blends more flags its tile as before.
#ifndef STP_LOG_DEPTH
#define STP_LOG_DEPTH 192
#endif
constexpr int BLEND_LOG_DEPTH = STP_LOG_DEPTH, BLEND_LOG_DEPTH_MIN = 32, BLEND_LOG_DEPTH_MAX = 512;
#ifndef STP_LOG_UNCOND
#define STP_LOG_UNCOND 1 // 1: the hierarchical recording forward stores a record in EVERY head step, without a branch -- a step that does
                         // not blend writes into the slot of the lane's next record, which the next blend overwrites -- and only the
                         // cursor's advance is conditional.  Needs one spare row per wave for the stores behind the last record.
#endif
// Layout of one wave's slice.  TWO layouts since round 6, by sort mode (log_blocked()):
//   rows    [record][lane]                 hierarchical mode (rounds 1-5: every mode).  Its lanes blend nearly in step: a store instruction
//                                          of the wave fills (most of) ONE 128-byte row.
//   blocked [record / 4][lane][record % 4] k-buffer mode.  A lane's four consecutive records lie side by side in ONE 8-byte piece of a
//                                          512-byte block; a 128-byte line belongs to sixteen neighbouring lanes = one 4x4 sub-tile.  The k-buffer kernel's lanes do NOT
//                                          blend in step (a pixel pops when ITS window is full): with rows every store instruction touched as
//                                          many lines as its lanes' record counts were apart (thirty and more at C3), a line stayed open until
//                                          the SLOWEST of 64 lanes had passed it, and the L2 wrote partial lines back again and again --
//                                          0.67 ms of the C3 forward's 2.45 (the same stores aimed at one row: 1.78 ms), and the reason why
//                                          that kernel's speed followed the physical placement of the image buffer (profiles/EXPERIMENTS.md,
//                                          round 6).  Blocked: a line is filled by one sub-tile's 16 pixels x 4 consecutive records, open for a
//                                          few steps whatever the other 48 lanes do: C3 forward 2.28-2.36 (rows, conditional stores) -> 2.14 ms.
// MEASURED the other way round too (one box, alternating, profiles/r06_experiments/log_layout_ab.txt): blocked in the hierarchical kernel costs
// three more address instructions per head step and eight lines per store instead of one -- C2-full forward 0.861 -> 0.884 ms, C5 unchanged,
// replay +1 % everywhere: rows stay there.  The replay reads record k of lane l through the same function (template argument by mode).
// A slice holds depth records per lane (a multiple of 8) + 8 spare rows (the hierarchical kernel's unconditional stores behind the last
// record land in the first of them).
#ifndef STP_LOG_BLOCK
#define STP_LOG_BLOCK 4 // MEASURED (one box, alternating, C3 forward / replay ms): 2: 2.237 / 1.554, 4: 2.135 / 1.544, 8: 2.162 / 1.561, 16: 2.39 / 1.66, 32: 3.08 / 1.86
#endif
constexpr int LOG_BLOCK = STP_LOG_BLOCK;                       // blocked layout: records of one lane side by side
static_assert(LOG_BLOCK == 2 || LOG_BLOCK == 4 || LOG_BLOCK == 8 || LOG_BLOCK == 16 || LOG_BLOCK == 32, "blocks of 2 .. 32 two-byte records");
constexpr int LOG_PIECE_SHIFT = LOG_BLOCK == 2 ? 2 : LOG_BLOCK == 4 ? 3 : LOG_BLOCK == 8 ? 4 : LOG_BLOCK == 16 ? 5 : 6; // a lane's piece of a block starts at lane << LOG_PIECE_SHIFT
constexpr int BLEND_LOG_SPARE = LOG_BLOCK < 8 ? 8 : LOG_BLOCK;                     // record rows (64 records = 128 B) of one wave's slice = depth + BLEND_LOG_SPARE
__host__ __device__ __forceinline__ size_t log_wave_bytes(int depth) { return (size_t)(depth + BLEND_LOG_SPARE) * 64 * sizeof(uint16_t); }
__device__ __forceinline__ char* log_wave_slice(uint32_t* blend_log, int tile, int wave, int depth)
{
    return reinterpret_cast<char*>(blend_log) + (size_t)(tile * 4 + wave) * log_wave_bytes(depth);
}
inline bool log_blocked(const StpSettings& s) { return s.sort_mode == 2; } // (MODE_KBUFFER, stp_internal.h)
#ifndef STP_LOG_LAYOUT
#define STP_LOG_LAYOUT 0 // A/B builds: 1 = rows everywhere (rounds 1-5), 2 = blocked everywhere
#endif
// byte offset, inside the wave's slice, of record k of the lane whose blocked piece starts at lane16 = 16 * lane; j2 = 2 * k (what the
// recording forwards keep as their cursor).  Blocked: ((j2 << 6) & ~(block bytes - 1)) | piece | (j2 & (piece bytes - 2)) -- v_lshlrev, v_and_or, v_and_or.
template <bool BLOCKED> __device__ __forceinline__ uint32_t log_record_offset(uint32_t j2, uint32_t lane16) // lane16 = lane << LOG_PIECE_SHIFT
{
    if constexpr ((BLOCKED && STP_LOG_LAYOUT != 1) || STP_LOG_LAYOUT == 2) return (((j2 << 6) & ~(128u * LOG_BLOCK - 1u)) | lane16) | (j2 & (2u * LOG_BLOCK - 2u));
    else return (j2 << 6) | (lane16 >> (LOG_PIECE_SHIFT - 1));
}
// The k-buffer forwards' log cursor (blocked layout).  A lane's four consecutive records are ONE 8-byte piece: they are collected in two registers -- a
// 64-bit shift register, the newest record enters at the top -- and stored as a whole piece when the fourth has arrived: a quarter of the store
// instructions, every one a complete piece, a 128-byte line complete after sixteen of them.  MEASURED (round 6, one box, alternating, C3): forward
// 2.02-2.11 -> 1.91-1.94 ms, 1.78 -> 1.24 GB written per launch (2-byte stores from the blending lanes: the line of a piece was dirtied four times).
#ifndef STP_KB_LOG_PIECES
#define STP_KB_LOG_PIECES (STP_LOG_BLOCK == 4 && STP_LOG_LAYOUT != 1)
#endif
struct BlockedLogCursor {
    char* wave;          // the wave's slice
    uint32_t cap2;       // 2 * depth: cursor of the first record that does not fit
    uint32_t piece;      // lane << LOG_PIECE_SHIFT
    uint32_t j2 = 0u;    // 2 * records so far (also beyond the depth)
    uint32_t lo = 0u, hi = 0u;
    __device__ __forceinline__ void append(bool upd, int pay)
    {
#if STP_KB_LOG_PIECES
        if (upd) {
            lo = __builtin_amdgcn_alignbit(hi, lo, 16);            // {hi, lo} >> 16
            hi = __builtin_amdgcn_alignbit((uint32_t)pay, hi, 16); // ... and the record into the top 16 bits
            if ((j2 & 6u) == 6u && j2 < cap2) *reinterpret_cast<uint2*>(wave + log_record_offset<true>(j2 & ~6u, piece)) = make_uint2(lo, hi);
        }
#else
        if (upd && j2 < cap2) *reinterpret_cast<log_t*>(wave + log_record_offset<true>(j2, piece)) = (log_t)pay;
#endif
        j2 += upd ? 2u : 0u;
    }
    __device__ __forceinline__ int records() const { return (int)(j2 >> 1); }
    __device__ __forceinline__ void flush() // the last, incomplete piece: its records sit at the TOP of the shift register
    {
#if STP_KB_LOG_PIECES
        const uint32_t have = (j2 >> 1) & 3u;
        if (have != 0u && (j2 & ~6u) < cap2) { // (the piece starts below the cap: depths are multiples of eight, a piece never straddles it)
            const unsigned long long acc = ((((unsigned long long)hi) << 32) | lo) >> (16u * (4u - have));
            *reinterpret_cast<uint2*>(wave + log_record_offset<true>(j2 & ~6u, piece)) = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32));
        }
#endif
    }
};

// what a recording forward reports back: the largest number of blends of any of its pixels (one compare per wave, an atomic only while
// the maximum still rises), collected per device and kind and handed to the host with the next forward's num_rendered
// (the word carries the reporting kind's tag in its upper half: a forward of another kind that shares the slot does not take the report for its own)
__device__ __forceinline__ void report_log_need(uint32_t* word, int nrec, uint32_t tag)
{
    int m = nrec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if (word != nullptr && (threadIdx.x & 63) == 0) {
        const uint32_t v = (tag << 16) | (uint32_t)min(m, 0xFFFF);
        uint32_t old = *reinterpret_cast<volatile uint32_t*>(word);
        // another kind's (or no) report: ours replaces it; our own kind's: the maximum.  One compare-and-swap loop decides both, so that of
        // several waves that meet the frame's zeroed word at once the largest report stays (an exchange let a smaller one land last: an
        // under-sized log for the next frame, i.e. a tile in the re-sorting fallback).  A wave that cannot raise the word issues no atomic.
        while ((old >> 16) != tag || v > old) {
            const uint32_t seen = atomicCAS(word, old, v);
            if (seen == old) break;
            old = seen;
        }
    }
}

struct FwdPixel {
    float T;
    float C[3];
};

struct BwdPixel {
    float T_final;
    float dL_dpix[3];
    float final_color[3];
    float bg_dot; // sum_ch bg[ch] * dL_dpix[ch]
    float T;
    float C[3];
};

__device__ __forceinline__ void init_fwd_pixel(FwdPixel& p)
{
    p.T = 1.0f;
    p.C[0] = p.C[1] = p.C[2] = 0.0f;
}

__device__ __forceinline__ void init_bwd_pixel(BwdPixel& b, const RenderArgs& a, bool inside, int px, int py)
{
    const size_t N = (size_t)a.W * a.H;
    const size_t pid = (size_t)a.W * py + px;
    b.T = 1.0f;
    b.T_final = inside ? a.final_T[pid] : 0.0f;
    b.bg_dot = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        b.C[ch] = 0.0f;
        b.dL_dpix[ch] = inside ? a.dL_dpix[ch * N + pid] : 0.0f;
        b.final_color[ch] = inside ? (a.pixel_colors[ch * N + pid] - b.T_final * a.bg[ch]) : 0.0f;
        b.bg_dot += a.bg[ch] * b.dL_dpix[ch];
    }
}

// Forward blend of the head entry; false = pixel saturated (nothing accumulated).
__device__ __forceinline__ bool blend_forward(FwdPixel& p, const float* __restrict__ features, int id, float alpha)
{
    const float test_T = p.T * (1.0f - alpha);
    if (test_T < T_THRESHOLD) return false;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) p.C[ch] += features[3 * (size_t)id + ch] * alpha * p.T;
    p.T = test_T;
    return true;
}

// same, with the colour already in registers (prefetched when the entry became the queue front)
__device__ __forceinline__ bool blend_forward_c(FwdPixel& p, const float (&c)[3], float alpha)
{
    const float test_T = p.T * (1.0f - alpha);
    if (test_T < T_THRESHOLD) return false;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) p.C[ch] += c[ch] * alpha * p.T;
    p.T = test_T;
    return true;
}

// Gradient of one blended pair, front-to-back formulation: the colour behind the current entry is
// reconstructed from the forward image, accum_rec = (final_colour - C_so_far) / T_after.
// Produces the nine per-Gaussian terms g[0..2] = dL/dcolour, g[3..4] = dL/dmean2D (x,y),
// g[5..7] = dL/dconic (xx, xy, yy), g[8] = dL/dopacity; the caller decides how they are accumulated.
// Returns false (and leaves g untouched) when the pixel saturates.
struct FrontData { float4 co; float2 xy; float c[3]; }; // what a backward blend reads of its Gaussian

__device__ __forceinline__ FrontData load_front(const RenderArgs& a, int id)
{
    FrontData f;
    f.co = a.conic_opacity[id];
    f.xy = a.means2D[id];
    f.c[0] = a.features[3 * (size_t)id + 0]; f.c[1] = a.features[3 * (size_t)id + 1]; f.c[2] = a.features[3 * (size_t)id + 2];
    return f;
}

__device__ __forceinline__ bool blend_backward_terms(BwdPixel& b, const RenderArgs& a, int px, int py, const FrontData& fd, float G, float (&g)[9])
{
    const float4 co = fd.co;
    const float alpha = fminf(0.99f, co.w * G);
    const float test_T = b.T * (1.0f - alpha);
    if (test_T < T_THRESHOLD) return false;
    const float2 xy = fd.xy;
    const float dx = xy.x - (float)px, dy = xy.y - (float)py;
    const float dchannel_dcolor = alpha * b.T;
    // gradients are compared with a relative tolerance (summation order already differs from the reference's
    // atomics), so the two quotients use the hardware reciprocal (v_rcp_f32, 1 ulp) instead of IEEE division
    const float rcp_test_T = __builtin_amdgcn_rcpf(test_T);
    const float rcp_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
    float dL_dalpha = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float c = fd.c[ch];
        b.C[ch] += c * alpha * b.T;
        const float accum_rec = (b.final_color[ch] - b.C[ch]) * rcp_test_T;
        dL_dalpha += (c - accum_rec) * b.dL_dpix[ch];
        g[ch] = dchannel_dcolor * b.dL_dpix[ch];
    }
    dL_dalpha *= b.T;
    dL_dalpha += (-b.T_final * rcp_1ma) * b.bg_dot;
    const float dL_dG = co.w * dL_dalpha;
    const float gdx = G * dx, gdy = G * dy;
    const float dG_ddelx = -gdx * co.x - gdy * co.y;
    const float dG_ddely = -gdy * co.z - gdx * co.y;
    g[3] = dL_dG * dG_ddelx * (0.5f * (float)a.W);
    g[4] = dL_dG * dG_ddely * (0.5f * (float)a.H);
    g[5] = -0.5f * gdx * dx * dL_dG;
    g[6] = -0.5f * gdx * dy * dL_dG;
    g[7] = -0.5f * gdy * dy * dL_dG;
    g[8] = G * dL_dalpha;
    b.T = test_T;
    return true;
}

// destination of term k of Gaussian id: k = 0..2 dL/dcolour, 3..4 dL/dmean2D, 5..7 dL/dconic (xx, xy, yy),
// 8 dL/dopacity -- consecutive floats of the Gaussian's 64-byte gradient record, so that nine lanes can hand
// over all nine sums with ONE atomic instruction that the memory pipeline treats as one request.
constexpr int GRAD_REC = STP_GRAD_RECORD_FLOATS;
__device__ __forceinline__ float* grad_slot(const RenderArgs& a, int id, int k)
{
    return a.grad_rec + (size_t)a.grad_stride * id + k;
}

// Straightforward accumulation: nine hardware fp32 atomics (global_atomic_add_f32; build with
// -munsafe-fp-atomics) per blended pair -- what the reference does (hierarchical_render.cuh:1131-1161).
__device__ __forceinline__ bool blend_backward(BwdPixel& b, const RenderArgs& a, int px, int py, int id, float G)
{
    float g[9];
    const FrontData fd = load_front(a, id);
    if (!blend_backward_terms(b, a, px, py, fd, G, g)) return false;
#pragma unroll
    for (int k = 0; k < 9; k++) atomicAdd(grad_slot(a, id, k), g[k]);
    return true;
}

// Per-pixel sorted insertion window in registers (reference resorted_render.cuh:74-119,186-197;
// hierarchical_render.cuh:386-417,509-522).  All indexing is compile-time (fully unrolled) so the
// arrays stay in VGPRs.
template <int CAP> struct Window {
    float depth[CAP];
    float store[CAP];
    int id[CAP];
    int num;
    __device__ __forceinline__ void init()
    {
        num = 0;
#pragma unroll
        for (int i = 0; i < CAP; i++) { depth[i] = FLT_MAX; store[i] = 0.0f; id[i] = -1; }
    }
    // strict '<': a new entry goes behind old entries of equal depth
    __device__ __forceinline__ void insert(float d, int gid, float st)
    {
#pragma unroll
        for (int s = 0; s < CAP; s++) {
            const bool sw = d < depth[s];
            const float td = depth[s]; const int ti = id[s]; const float ts = store[s];
            depth[s] = sw ? d : td; id[s] = sw ? gid : ti; store[s] = sw ? st : ts;
            d = sw ? td : d; gid = sw ? ti : gid; st = sw ? ts : st;
        }
        num++;
    }
    // Branch-free form of insert() for the hot paths: inserts only where `pass` holds.  It reproduces the swap
    // loop above exactly, ties included: the value carried through the loop is always max(candidate, depth[s-1]),
    // so slot s swaps iff  d < depth[s]  and not (d < depth[s-1] and depth[s-1] == depth[s]); the payload follows
    // the same chain, the depths themselves are one median-of-three per slot.
    __device__ __forceinline__ void insert_if(bool pass, float d, int gid, float st)
    {
        const float c = pass ? d : FLT_MAX; // FLT_MAX is smaller than nothing: no slot changes
        bool sw[CAP];
        sw[0] = c < depth[0];
#pragma unroll
        for (int s = 1; s < CAP; s++) // (bitwise operators: no short-circuit control flow)
            sw[s] = (bool)((int)(c < depth[s]) & ~((int)(c < depth[s - 1]) & (int)(depth[s - 1] == depth[s])) & 1);
#pragma unroll
        for (int s = 0; s < CAP; s++) {
            const int oi = id[s];
            const float os = store[s];
            id[s] = sw[s] ? gid : oi;
            store[s] = sw[s] ? st : os;
            gid = sw[s] ? oi : gid;
            st = sw[s] ? os : st;
        }
#pragma unroll
        for (int s = CAP - 1; s > 0; s--) depth[s] = __builtin_amdgcn_fmed3f(depth[s - 1], depth[s], c);
        depth[0] = fminf(depth[0], c);
        num += (int)pass;
    }
    // ---- the always-full form used by the forward head level -------------------------------------------------------
    // The queue always holds CAP slots: its k real entries in ascending order, preceded by CAP - k PADS of depth
    // -FLT_MAX (store 0).  One step = the reference's "pop the front if the queue is full, then insert the candidate if
    // it passes": the front slot is consumed by the caller -- a pad when the queue was not full, i.e. exactly when the
    // reference does not pop -- and replace_front() puts the candidate (or a new pad, when it did not pass) into the
    // remaining CAP - 1 slots.  No separate shift, no element count: slot s - 1 receives what the reference's swap loop
    // leaves in slot s of the popped queue (same tie rule as insert_if), the last slot the element carried out.
    __device__ __forceinline__ void init_padded()
    {
        num = 0;
#pragma unroll
        for (int i = 0; i < CAP; i++) { depth[i] = -FLT_MAX; store[i] = 0.0f; id[i] = 0; }
    }
    // `real`: c is a candidate (tie rule of the swap loop applies); otherwise c is a pad (-FLT_MAX, goes in front of
    // every real entry WITHOUT disturbing the order of equal-depth entries -- the reference does not insert at all) or
    // the drain filler (FLT_MAX, goes last).
    __device__ __forceinline__ void replace_front(bool real, float c, int gid, float st)
    {
        if constexpr (CAP > 1) {
            bool sw[CAP];
            sw[1] = c < depth[1];
#pragma unroll
            for (int s = 2; s < CAP; s++)
                sw[s] = (bool)((int)(c < depth[s]) & ~((int)real & (int)(c < depth[s - 1]) & (int)(depth[s - 1] == depth[s])) & 1);
#pragma unroll
            for (int s = 1; s < CAP; s++) {
                const int oi = id[s];
                const float os = store[s];
                id[s - 1] = sw[s] ? gid : oi;
                store[s - 1] = sw[s] ? st : os;
                gid = sw[s] ? oi : gid;
                st = sw[s] ? os : st;
            }
            float nd[CAP];
            // (min / max as bare instructions: fminf / fmaxf cost a canonicalising v_max per operand first, and the compiler
            // folds a median with an infinity back into exactly that -- three instructions where one does)
            asm("v_min_f32 %0, %1, %2" : "=v"(nd[0]) : "v"(depth[1]), "v"(c));
#pragma unroll
            for (int s = 1; s < CAP - 1; s++) nd[s] = __builtin_amdgcn_fmed3f(depth[s], depth[s + 1], c);
            asm("v_max_f32 %0, %1, %2" : "=v"(nd[CAP - 1]) : "v"(depth[CAP - 1]), "v"(c));
#pragma unroll
            for (int s = 0; s < CAP; s++) depth[s] = nd[s];
        } else depth[0] = c;
        id[CAP - 1] = gid;
        store[CAP - 1] = st;
    }
    // pop() where `need` holds, as selects
    __device__ __forceinline__ void pop_if(bool need)
    {
#pragma unroll
        for (int i = 1; i < CAP; i++) {
            depth[i - 1] = need ? depth[i] : depth[i - 1];
            store[i - 1] = need ? store[i] : store[i - 1];
            id[i - 1] = need ? id[i] : id[i - 1];
        }
        depth[CAP - 1] = need ? FLT_MAX : depth[CAP - 1];
        num -= (int)need;
    }
    __device__ __forceinline__ void pop()
    {
#pragma unroll
        for (int i = 1; i < CAP; i++) { depth[i - 1] = depth[i]; store[i - 1] = store[i]; id[i - 1] = id[i]; }
        depth[CAP - 1] = FLT_MAX;
        num--;
    }
};

} // namespace stp
