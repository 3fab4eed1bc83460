// stp_tilesort.hip -- second half of the (tile, depth) sort, fused with the entry gather.
//
// The reference sorts the (tile << 32 | depth) keys with one device-wide stable radix sort over 32 + log2(tiles) bits
// (rasterizer_impl.cu:344-352): six 8-bit passes over all R pairs at C2.  A tile's list is a few hundred entries long:
// once the duplicates are grouped by tile (a radix sort on the tile bits only -- two passes; or, STP_SORT=counters, through
// per-tile counters without any sort pass, stp_binning.hip), the depth order inside a segment is established by the tile's own workgroup in
// LDS -- bitonic network on (depth bits, Gaussian id).  The reference's stable sort leaves equal (tile, depth) keys in
// the order duplicateWithKeys emitted them, which is the order of the Gaussian index: the id as the minor key gives
// exactly that list whatever order the segment arrived in.  The same workgroup writes the sorted keys, the sorted id
// list AND the list-ordered entry records (stp_preprocess.hip: gather_entries_kernel), which it would otherwise take
// another pass over the list to build.  Segments longer than TS_CAP entries are sorted by the workgroup with stable
// 8-bit counting passes (id bytes first when the segment is not in id order yet, then the four depth bytes) through the
// otherwise unused unsorted arrays.  Same sorted list, bit for bit.
#include "stp_internal.h"
#include "stp_device.h"
#include <rocprim/block/block_radix_sort.hpp>

namespace stp {

namespace {

// Entries a workgroup sorts in LDS: two instantiations, launched back to back -- SMALL (8 KB of LDS: the latency-bound
// gather wants many workgroups per CU) takes the tiles with up to TS_SMALL entries and leaves at once on the others,
// LARGE (32 KB) takes the rest, up to TS_CAP in LDS and beyond that through the counting passes.
constexpr int TS_SMALL = 1024, TS_CAP = 4096;
#ifndef STP_GATHER_WAVES
#define STP_GATHER_WAVES 6 // waves per SIMD the small instantiation is compiled for: 80 VGPRs as the compiler likes it = 6.  MEASURED (round 5, after the
                           // hierarchical forward gained 7 % from a fifth wave): 8 waves = 64 VGPRs + 32 B of scratch: sort stage 0.3157 / 0.3151 / 0.3161 -> 0.3180 / 0.3201 / 0.3204 ms
#endif

struct TileSortArgs {
    const uint2* ranges;
    uint64_t* keys;           // in: grouped by tile, out: sorted
    uint32_t* point_list;     // likewise
    uint64_t* keys_scratch;   // the unsorted arrays: scratch of the long-segment path
    uint32_t* list_scratch;
    const float4* gpack;      // nullptr: no entry records (GLOBAL mode)
    const float* features;
    int id_passes;            // long segments: counting passes on the id bytes before the depth passes (0: already in id order)
    int gx;                   // tiles per row
    int tile0;                // first tile of the frame's tile-row window (the grid covers the window's tiles)
    const uint32_t* tile_order; // nullptr, or workgroup j takes tile tile0 + tile_order[j] (longest list first: tile_order_kernel)
    int cull_mask;            // leave every entry's 16-bit sub-tile mask in entF.w (see write_entry): 1 = hierarchical mode's 4x4 culling, 2 = the k-buffer kernel's sub-tile pre-test
    float4* entA; float4* entB; float4* entC; float4* entD; float4* entF;
};

#ifndef STP_BITONIC_WAVE
#define STP_BITONIC_WAVE 1 // 0: a workgroup barrier behind every stage of the bitonic network
#endif
#ifndef STP_GATHER_ABLATE
#define STP_GATHER_ABLATE 0 // timing experiments (results WRONG): 1 = no sub-tile masks, 2 = no colour read, 3 = no gpack read (constants), 4 = no entry stores, 5 = no sort network.
                           // MEASURED (round 4, C2-full, sort stage 0.330 ms, two alternating rounds): without the masks -15 us, without the colour read -22,
                           // without the entry stores -48 (240 MB: the HBM floor of that part), without the bitonic network -37; the rest is key / list / gpack IO.
#endif
__device__ __forceinline__ void write_entry(const TileSortArgs& a, size_t i, int id, int tile)
{
#if STP_GATHER_ABLATE == 3
    const float4 pa = make_float4(1, 0, 0, 1), pb = make_float4(0, 1, 0, 0), pc = make_float4(0, 8.f + id * 1e-9f, 8, 0), pd = make_float4(1, 0, 1, 0.5f);
    const float3 col = make_float3(a.features[3 * (size_t)id], a.features[3 * (size_t)id + 1], a.features[3 * (size_t)id + 2]);
#elif STP_GATHER_ABLATE == 2
    const float4* __restrict__ gp = a.gpack + 4 * (size_t)id;
    const float4 pa = gp[0], pb = gp[1], pc = gp[2], pd = gp[3];
    const float3 col = make_float3(pa.x, pa.y, pa.z);
#else
    const float4* __restrict__ gp = a.gpack + 4 * (size_t)id; // one 64-byte line written by preprocess_kernel
    const float4 pa = gp[0], pb = gp[1], pc = gp[2], pd = gp[3];
    const float3 col = make_float3(a.features[3 * (size_t)id], a.features[3 * (size_t)id + 1], a.features[3 * (size_t)id + 2]);
#endif
#if STP_GATHER_ABLATE == 4
    if (pa.x + pb.y + pc.z + pd.w + col.x == 1.2345e-30f) a.entA[i] = pa;
    return;
#endif
    a.entA[i] = pa;
    a.entB[i] = pb;
    a.entC[i] = make_float4(pc.x, pc.y, pc.z, __int_as_float(id));
    a.entD[i] = pd;
    float spare = 0.0f;
    if (a.cull_mask && STP_GATHER_ABLATE != 1) spare = __uint_as_float(subtile_mask(a.cull_mask, pd, make_float2(pc.y, pc.z), tile % a.gx, tile / a.gx)); // (stp_device.h)
    a.entF[i] = make_float4(col.x, col.y, col.z, spare);
}

// Segments of 1025 .. 4096 entries that arrive in Gaussian-id order (the tile-bit radix sort is stable): a stable sort on the
// 32 depth bits alone leaves equal depths in id order, i.e. IS the (depth, id) order -- rocPRIM's workgroup radix sort, four
// 8-bit passes over IPT keys per thread, instead of the bitonic network's 66 / 78 barrier-separated stages on 64-bit keys
// (which cost 0.58 ms per C3 frame and 0.78 ms per C5 frame, the third-largest kernel there).  Blocked arrangement in
// (thread t holds entries t * IPT ...: that is the order the sort is stable in), striped out (coalesced stores).
template <int IPT> struct TileRadix {
    using Sort = rocprim::block_radix_sort<uint32_t, 256, IPT, uint32_t>;
    template <class WriteEntry>
    static __device__ __forceinline__ void run(typename Sort::storage_type& st, uint64_t* keys, uint32_t* list, int n, int tid, WriteEntry&& write_entry_at)
    {
        const uint64_t tile_bits = keys[0] & 0xFFFFFFFF00000000ull;
        uint32_t k[IPT], v[IPT];
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const int i = tid * IPT + j;
            k[j] = i < n ? (uint32_t)keys[i] : 0xFFFFFFFFu; // (a depth is a non-negative float: its bits stay below the padding)
            v[j] = i < n ? list[i] : 0xFFFFFFFFu;
        }
        __syncthreads(); // (every thread has read its part of the segment before anybody overwrites it)
        Sort().sort_to_striped(k, v, st, 0, 32);
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            const int i = j * 256 + tid;
            if (i < n) {
                keys[i] = tile_bits | k[j];
                list[i] = v[j];
                write_entry_at(i, (int)v[j]);
            }
        }
    }
};

template <int CAP, int MIN_N>
__global__ void __launch_bounds__(256, (CAP == TS_SMALL ? STP_GATHER_WAVES : 4)) tile_sort_gather_kernel(const TileSortArgs a)
{
    using Radix8 = TileRadix<8>;
    using Radix16 = TileRadix<16>;
    constexpr size_t RADIX_BYTES = CAP == TS_CAP ? (sizeof(typename Radix16::Sort::storage_type) > sizeof(typename Radix8::Sort::storage_type) ? sizeof(typename Radix16::Sort::storage_type) : sizeof(typename Radix8::Sort::storage_type)) : 0;
    constexpr size_t RAW_BYTES = RADIX_BYTES > sizeof(uint64_t) * CAP ? RADIX_BYTES : sizeof(uint64_t) * CAP;
    __shared__ __attribute__((aligned(16))) char s_raw[RAW_BYTES]; // the bitonic network's keys, or the radix sort's exchange area
    uint64_t* const s_key = reinterpret_cast<uint64_t*>(s_raw); // (depth bits << 32) | Gaussian id
    static_assert(CAP != TS_CAP || RAW_BYTES >= sizeof(int) * 13 * 256, "the long-segment path keeps its counters in the sort's LDS area");
    const int tid = (int)threadIdx.x;
    // XCD-aware tile order (same map as the render kernels): workgroup ids are dealt round-robin to the 8 XCDs, so each XCD gets
    // a contiguous run of tiles and the 64-byte lines of the Gaussians that neighbouring tiles share hit in that XCD's L2
    const int n_wg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int xq = n_wg >> 3, xr = n_wg & 7, xcd = wg & 7;
    const int tile = a.tile0 + (a.tile_order ? (int)a.tile_order[wg] : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wg >> 3));
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n <= MIN_N || (CAP == TS_SMALL && n > TS_SMALL)) return; // empty, or the other instantiation's tile
    uint64_t* const keys = a.keys + range.x;
    uint32_t* const list = a.point_list + range.x;

    if constexpr (CAP == TS_CAP) {
        if (n <= CAP && a.id_passes == 0) { // (segments in id order: always, unless the list was binned through atomic cursors)
            auto we = [&](int i, int id) __attribute__((always_inline)) { if (a.gpack) write_entry(a, (size_t)range.x + i, id, tile); };
            if (n <= 2048) Radix8::run(*reinterpret_cast<typename Radix8::Sort::storage_type*>(s_raw), keys, list, n, tid, we);
            else Radix16::run(*reinterpret_cast<typename Radix16::Sort::storage_type*>(s_raw), keys, list, n, tid, we);
            return;
        }
    }
    if (n <= CAP) {
        int m = 2;
        while (m < n) m <<= 1;
        const uint64_t tile_bits = keys[0] & 0xFFFFFFFF00000000ull;
        for (int i = tid; i < m; i += 256) {
            s_key[i] = i < n ? ((keys[i] << 32) | list[i]) : ~0ull;
        }
        __syncthreads();
        // A stage with partner distance j <= 64 keeps every wave inside its own 128 keys (the 64 consecutive comparators c of a wave cover keys
        // [128 (c / 64), 128 (c / 64) + 128)): between two such stages the wave's own LDS order is all the synchronisation there is to need --
        // a workgroup barrier only around the stages that cross waves (3 of the 45 stages of a 512-key network, 6 of 55 at 1024 keys).
        for (int k = 2; k <= (STP_GATHER_ABLATE == 5 ? 0 : m); k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int c = tid; c < (m >> 1); c += 256) {
                    const int lo = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                    const int hi = lo | j;
                    const bool up = (lo & k) == 0;
                    const uint64_t x = s_key[lo], y = s_key[hi];
                    if ((x > y) == up) { s_key[lo] = y; s_key[hi] = x; }
                }
                const int j_next = j > 1 ? (j >> 1) : k; // (the first distance of the next merge; behind the last stage: the read-out, which crosses waves)
                const bool last = j == 1 && k == m;
                if (!STP_BITONIC_WAVE || j > 64 || j_next > 64 || last) __syncthreads();
                else wave_sync();
            }
        for (int i = tid; i < n; i += 256) {
            const uint64_t k = s_key[i];
            const int id = (int)(uint32_t)k;
            keys[i] = tile_bits | (k >> 32);
            list[i] = (uint32_t)id;
            if (a.gpack) write_entry(a, (size_t)range.x + i, id, tile);
        }
        return;
    }

    // ---- long segment: stable counting passes (LSD: id bytes if needed, then the four depth bytes), keys/list <-> scratch ----
    // Round 6: ranks by wave ballots.  The lanes of a wave that hold the SAME digit find each other with eight ballots (one per digit bit), a
    // lane's rank among them is a popcount, the waves' counts per digit meet in LDS and the thread that owns a digit advances its base --
    // three barriers per chunk of 1024 and no loop over the chunk.  (Rounds 1-5 counted "how many of the
    // threads below me hold my digit" with a 256-iteration loop per element and pass: 5.1 ms of sort stage on C2H, whose clusters' tiles
    // hold up to 28 000 entries -- now 0.99: what is left is ONE workgroup walking the longest tile; a pass skips itself when every key
    // has the same digit, and all passes' histograms come from one sweep.)
    if constexpr (CAP == TS_SMALL) return; // (not reached: those tiles belong to the large instantiation)
    // (counters in the LDS area the in-LDS sorts of shorter segments use)
    int* const s_hist = reinterpret_cast<int*>(s_raw); // [256] the running base of every digit in the current pass
    int* const s_wcnt = s_hist + 256;                  // [4 rounds][256 digits]: the four waves' counts of the digit in that round of the current chunk, one 8-bit field per wave
    int* const s_hall = s_hist + 5 * 256;              // [8 passes][256]: every pass's digit histogram, from ONE sweep over the segment (the keys do not change between passes)
    uint64_t* src_k = keys; uint32_t* src_v = list;
    uint64_t* dst_k = a.keys_scratch + range.x; uint32_t* dst_v = a.list_scratch + range.x;
    const int n_pass = a.id_passes + 4;
    const int wv = tid >> 6, ln = tid & 63;
    const unsigned long long lt_mask = (1ull << ln) - 1ull;
    int done_passes = 0;
    for (int i = tid; i < 8 * 256; i += 256) s_hall[i] = 0;
    __syncthreads();
    {   // four elements per thread in flight
        const int n4 = n & ~1023;
        for (int i0 = 0; i0 < n; i0 += 1024) {
            uint64_t k[4]; uint32_t v[4]; bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { const int i = i0 + 256 * r + tid; ok[r] = i0 < n4 || i < n; k[r] = ok[r] ? src_k[i] : 0ull; v[r] = (ok[r] && a.id_passes) ? src_v[i] : 0u; }
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (ok[r]) {
                    for (int p = 0; p < a.id_passes; p++) atomicAdd(&s_hall[256 * p + (int)((v[r] >> (8 * p)) & 0xFF)], 1);
#pragma unroll
                    for (int p = 0; p < 4; p++) atomicAdd(&s_hall[256 * (a.id_passes + p) + (int)((k[r] >> (8 * p)) & 0xFF)], 1);
                }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < n_pass; pass++) {
        const bool on_id = pass < a.id_passes;
        const int shift = 8 * (on_id ? pass : pass - a.id_passes);
        auto digit = [&](uint64_t k, uint32_t v) __attribute__((always_inline)) { return (int)(((on_id ? (uint64_t)v : k) >> shift) & 0xFF); };
        s_hist[tid] = s_hall[256 * pass + tid];
#pragma unroll
        for (int r = 0; r < 4; r++) s_wcnt[256 * r + tid] = 0;
        __syncthreads();
        const bool same = s_hist[digit(src_k[0], src_v[0])] == n; // (workgroup-uniform: every thread reads the same word) nothing to order in this pass
        __syncthreads();
        if (same) continue;
        if (tid < 64) { // exclusive scan of the 256 counters by one wave, four per lane
            int c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { c[k] = s_hist[4 * tid + k]; sum += c[k]; }
            int inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (tid >= o) inc += v; }
            int run = inc - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) { s_hist[4 * tid + k] = run; run += c[k]; }
        }
        __syncthreads();
        // chunks of 1024 in order (the pass is stable): element c0 + 256 r + tid is thread tid's r-th -- four independent loads in flight per thread,
        // three barriers per 1024 elements (a 28 000-entry tile of C2H is ONE workgroup's job: its chunks are the stage's critical path)
        for (int c0 = 0; c0 < n; c0 += 4 * 256) {
            uint64_t k[4]; uint32_t v[4]; int d[4], rank[4]; bool valid[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = c0 + 256 * r + tid;
                valid[r] = i < n;
                k[r] = 0; v[r] = 0;
                if (valid[r]) { k[r] = src_k[i]; v[r] = src_v[i]; }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                d[r] = valid[r] ? digit(k[r], v[r]) : 0;
                unsigned long long peers = __ballot(valid[r]);
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    const bool one = ((d[r] >> bit) & 1) != 0;
                    const unsigned long long bal = __ballot(one);
                    peers &= one ? bal : ~bal;
                }
                rank[r] = __popcll(peers & lt_mask);
                // my wave's count of the digit in round r, in its 8-bit field of the (round, digit) word (at most 64 per wave and round)
                if (valid[r] && rank[r] == 0) atomicAdd(&s_wcnt[256 * r + d[r]], __popcll(peers) << (8 * wv));
            }
            __syncthreads();
            auto fields = [](unsigned int wc) __attribute__((always_inline)) { return (int)((wc & 0xFFu) + ((wc >> 8) & 0xFFu) + ((wc >> 16) & 0xFFu) + (wc >> 24)); };
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (valid[r]) {
                    int below = 0;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        if (rr < r) below += fields((unsigned int)s_wcnt[256 * rr + d[r]]);
                    const unsigned int wc = (unsigned int)s_wcnt[256 * r + d[r]];
                    below += (int)((wv > 0 ? (wc & 0xFFu) : 0u) + (wv > 1 ? ((wc >> 8) & 0xFFu) : 0u) + (wv > 2 ? ((wc >> 16) & 0xFFu) : 0u));
                    const int at = s_hist[d[r]] + below + rank[r];
                    dst_k[at] = k[r];
                    dst_v[at] = v[r];
                }
            }
            __syncthreads();
            { // thread t owns digit t: advance its base by the chunk's count of it, clear the rounds' and waves' counts
                int tot = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned int wc = (unsigned int)s_wcnt[256 * r + tid];
                    if (wc != 0u) { tot += fields(wc); s_wcnt[256 * r + tid] = 0; }
                }
                s_hist[tid] += tot;
            }
            __syncthreads();
        }
        uint64_t* tk = src_k; src_k = dst_k; dst_k = tk;
        uint32_t* tv = src_v; src_v = dst_v; dst_v = tv;
        done_passes++;
        __threadfence_block();
        __syncthreads();
    }
    if (done_passes & 1) { // an odd number of passes leaves the result in the scratch arrays
        for (int i = tid; i < n; i += 256) { keys[i] = src_k[i]; list[i] = src_v[i]; }
        __threadfence_block();
        __syncthreads();
    }
    if (a.gpack) {
#pragma unroll 4
        for (int i = tid; i < n; i += 256) write_entry(a, (size_t)range.x + i, (int)list[i], tile);
    }
}

} // namespace

hipError_t launch_tile_sort_gather(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img, int R, bool unordered, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    TileSortArgs a{};
    a.id_passes = 0;
    if (unordered) // segments filled through atomic cursors: not in id order
        for (unsigned int top = (unsigned int)(f.P > 1 ? f.P - 1 : 1); top; top >>= 8) a.id_passes++;
    a.ranges = img.ranges; a.keys = b.keys; a.point_list = b.point_list; a.keys_scratch = b.keys_unsorted; a.list_scratch = b.point_list_unsorted;
    const bool entries = f.s.sort_mode == MODE_HIER || f.s.sort_mode == MODE_KBUFFER;
    a.gpack = entries ? g.gpack : nullptr;
    a.features = f.colors_precomp ? f.colors_precomp : g.rgb;
    a.gx = f.gx;
    a.tile0 = f.gx * f.ty0;
    a.cull_mask = subtile_mask_kind(f.s);
    // (STP_GATHER_ORDER: 0 = default: spatial, 1: longest first inside every XCD's contiguous run, 2: longest first over the frame -- tile_order_kernel, measured there)
    const int gmode = (tile_order_used(f) && !unordered) ? gather_order_mode() : 0;
    a.tile_order = gmode == 1 ? img.tile_counts + f.gx * f.ty0 : gmode == 2 ? img.tile_cursor + f.gx * f.ty0 : nullptr;
    a.entA = b.entA; a.entB = b.entB; a.entC = b.entC; a.entD = b.entD; a.entF = b.entF;
    const int n_tiles = f.gx * (f.ty1 - f.ty0);
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL((tile_sort_gather_kernel<TS_SMALL, 0>), dim3(n_tiles), dim3(256), 0, st, a);
    hipLaunchKernelGGL((tile_sort_gather_kernel<TS_CAP, TS_SMALL>), dim3(n_tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
