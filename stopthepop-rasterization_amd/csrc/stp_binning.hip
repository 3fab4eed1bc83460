// stp_binning.hip -- prefix sum of tile counts and the stable (tile, depth) radix sort.
//
// Replaces cub::DeviceScan::InclusiveSum (reference rasterizer_impl.cu:189,313) and
// cub::DeviceRadixSort::SortPairs on key bits [0, 32+bit) (rasterizer_impl.cu:211-214,344-352) with
// rocPRIM's device-wide primitives (both are stable LSD radix sorts, so the sorted list is the
// same list).
#include <cstdlib>
#include <cstring> // rocPRIM 7.2's texture_cache_iterator.hpp uses unqualified memset
#include <rocprim/rocprim.hpp>

#include "stp_internal.h"
#include "stp_device.h"

namespace stp {

uint32_t higher_msb(uint32_t n) // reference rasterizer_impl.cu:37-52: bisect for the bit above the MSB
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t scan_temp_bytes(size_t P)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, P, rocprim::plus<uint32_t>());
    return bytes;
}

// ---- the tile-bit sort: rocPRIM's onesweep kernels under our own driver ---------------------------------------------------------------
// rocprim::radix_sort_pairs on the 13 tile bits is two onesweep passes (80 us at C2) -- and, per call, FIVE hipMemsetAsync of a few KB
// (the digit histogram; per pass the look-back states and the ordered-block-id counter), each a launch-bound __amd_rocclr_fillBufferAligned
// of 8 us on the GPU: 41 us per frame for nothing (profiles/r03_full/kernel_stats.csv: 5.1 fills per frame).  The library's host function
// cannot be told to leave them out, but its device code is all in headers: the three kernels below are rocPRIM's own device functions
// (onesweep_histograms / onesweep_scan_histograms / onesweep_iteration, its gfx950 tuning parameters, its look-back states), launched by
// launch_sort() with every pass owning its OWN look-back states and block counter, so that all of them are zeroed ONCE -- by trailing
// workgroups of duplicate_kernel, which runs in front of the sort anyway (BinningState::sort_zero).  Same passes, same stable order,
// same sorted list bit for bit (STP_TILE_SORT=rocprim keeps the library call for comparison).
// Per frame at C2 under the kernel trace (profiles/r04_sortprof.txt): the library 2 x 42.5 (passes) + 16.2 + 12.5 + five fills of 7.6 = 152 us,
// this driver 2 x 56.7 + 13.2 + 11.4 = 138 us.  The passes themselves look slower here only because they START earlier: the SH colour kernel runs on
// the side stream beside the sort, and without the fills in front of them the passes overlap more of it (clearing the states closer to a pass --
// by the scan kernel, by the pass before -- or with a memset in front changed the passes' durations exactly as far as it delayed them).  What
// counts is the stage: sort stage -10 .. -25 us on every workload (profiles/r04_tilesort_driver_ab.txt).
// rocprim::detail is private, unversioned API: this driver is written against rocPRIM 4.2 (ROCm 7.2).  Another version must be looked at before it is
// trusted: there the driver is NOT compiled (a build warning says so), own_onesweep_driver() answers false and launch_sort() takes the library's own
// host function -- the version-proof path that STP_TILE_SORT=rocprim selects here -- instead of mis-tuning or mis-launching silently.
#include <rocprim/rocprim_version.hpp>
#if ROCPRIM_VERSION_MAJOR == 4 && ROCPRIM_VERSION_MINOR == 2
#define STP_OWN_ONESWEEP 1
#else
#define STP_OWN_ONESWEEP 0
#warning "stp_binning.hip: the own onesweep driver is written against rocPRIM 4.2; with this rocPRIM the tile-bit sort uses rocprim::radix_sort_pairs (check rocprim::detail::onesweep_* against this version, then extend the #if)"
#endif
namespace {
#if STP_OWN_ONESWEEP
namespace rpd = rocprim::detail;
using OsConfig = rpd::wrapped_radix_sort_onesweep_config<rocprim::default_config, uint64_t, uint32_t>;
constexpr rpd::radix_sort_onesweep_config_params os_params()
{
    rpd::radix_sort_onesweep_config_params p = OsConfig::architecture_config<rpd::target_arch::gfx950>::params; // 512 threads x 16 keys, 8 bits per place
    // The library's 8192 keys per workgroup leave a C2 frame (2.6 M entries) with 320 workgroups = 2.5 waves per SIMD for a scatter that lives on
    // memory parallelism.  MEASURED in the frame (round 5, two boxes, three alternating rounds each, profiles/r05_experiments/onesweep_tuning.txt):
    // 512 x 8 sort stage 0.318 -> 0.300 and 0.330 -> 0.314 ms at C2-full, C3 0.987 -> 0.971, C5 1.211 -> 1.197, C2-min / C4 unchanged; 512 x 12 -16 / -7 us,
    // 512 x 10 -7, 512 x 6 -5, 256 x 16 -10, 256 x 12 -8, 256 x 8 +4, 1024 x 4 -6, 1024 x 6 -2.  (Round 4 had tried 512 x 12 under the library's own
    // driver, whose fills in front of every pass hid the difference.)
    p.sort.items_per_thread = 8;
#ifdef STP_OS_SORT_BLOCK
    p.sort.block_size = STP_OS_SORT_BLOCK;
#endif
#ifdef STP_OS_SORT_IPT
    p.sort.items_per_thread = STP_OS_SORT_IPT;
#endif
#ifdef STP_OS_HIST_BLOCK
    p.histogram.block_size = STP_OS_HIST_BLOCK;
#endif
#ifdef STP_OS_HIST_IPT
    p.histogram.items_per_thread = STP_OS_HIST_IPT;
#endif
    return p;
}
constexpr rpd::radix_sort_onesweep_config_params OS = os_params();
constexpr unsigned OS_RADIX = 1u << OS.radix_bits_per_place;
constexpr unsigned OS_HIST_ITEMS = OS.histogram.block_size * OS.histogram.items_per_thread;
constexpr unsigned OS_SORT_ITEMS = OS.sort.block_size * OS.sort.items_per_thread;
constexpr unsigned OS_MAX_PLACES = 4; // 32 tile bits at most
using OsBlockId = rpd::block_id_wrapper<unsigned int, true>;

struct OsLayout { // inside BinningState::sort_temp
    uint32_t* offsets;                                  // [places][radix] digit histograms -> exclusive offsets   } zeroed once per frame
    unsigned int* block_ids;                            // [places]                                                   } (sort_zero)
    rpd::onesweep_lookback_state* lookback;             // [places][radix * blocks]                                   }
    uint32_t* offsets_tmp;                              // [radix] (next-batch offsets: written, never read with one batch)
    uint64_t* keys_tmp;                                 // [R]
    uint32_t* values_tmp;                               // [R]
    size_t zero_bytes, total, lookback_offset;          // zero_bytes: all places' states; a frame clears [0, lookback_offset + its places' states)
    uint32_t sort_blocks;
};
OsLayout os_layout(char* base, size_t R)
{
    OsLayout L{};
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L.sort_blocks = (uint32_t)((R + OS_SORT_ITEMS - 1) / OS_SORT_ITEMS);
    size_t off = 0;
    // (the look-back states last: a frame's sort uses the first `places` of them -- two at 1080p -- and only those are cleared, sort_zero_region)
    L.offsets = reinterpret_cast<uint32_t*>(base + off); off += up(sizeof(uint32_t) * OS_MAX_PLACES * OS_RADIX);
    L.block_ids = reinterpret_cast<unsigned int*>(base + off); off += up(sizeof(unsigned int) * OS_MAX_PLACES);
    L.lookback = reinterpret_cast<rpd::onesweep_lookback_state*>(base + off);
    L.lookback_offset = off;
    off += up(sizeof(rpd::onesweep_lookback_state) * OS_MAX_PLACES * OS_RADIX * (size_t)L.sort_blocks);
    L.zero_bytes = off;
    L.offsets_tmp = reinterpret_cast<uint32_t*>(base + off); off += up(sizeof(uint32_t) * OS_RADIX);
    L.keys_tmp = reinterpret_cast<uint64_t*>(base + off); off += up(sizeof(uint64_t) * R);
    L.values_tmp = reinterpret_cast<uint32_t*>(base + off); off += up(sizeof(uint32_t) * R);
    L.total = off;
    return L;
}

__global__ void __launch_bounds__(OS.histogram.block_size) os_histograms_kernel(const uint64_t* keys, uint32_t* counts, uint32_t size, uint32_t full_blocks, unsigned begin_bit, unsigned end_bit)
{
    rpd::onesweep_histograms<OS.histogram.block_size, OS.histogram.items_per_thread, OS.radix_bits_per_place, false>(
        keys, counts, size, full_blocks, rocprim::identity_decomposer{}, begin_bit, end_bit);
}
__global__ void __launch_bounds__(OS.histogram.block_size) os_scan_histograms_kernel(uint32_t* offsets)
{
    rpd::onesweep_scan_histograms<OS.histogram.block_size, OS.radix_bits_per_place>(offsets);
}
__global__ void __launch_bounds__(OS.sort.block_size) os_iteration_kernel(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* values_in, uint32_t* values_out, unsigned size,
                                                                         uint32_t* offsets_in, uint32_t* offsets_out, rpd::onesweep_lookback_state* lookback, unsigned bit,
                                                                         unsigned radix_bits, unsigned full_blocks, OsBlockId ordered_bid)
{
    rpd::onesweep_iteration<OS.sort.block_size, OS.sort.items_per_thread, OS.radix_bits_per_place, false, OS.radix_rank_algorithm>(
        keys_in, keys_out, values_in, values_out, size, offsets_in, offsets_out, lookback, rocprim::identity_decomposer{}, bit, radix_bits, full_blocks, ordered_bid);
}

// (One kernel for the histograms AND their scans -- workgroup histograms in LDS, the last workgroup by ticket scans in place -- was built and
// measured in round 5: sort stage equal to these two library kernels within 3 us in four A/B runs; removed.  What it taught: a device-scope
// __threadfence() behind duplicate_kernel's 31 MB of keys writes the XCD's L2 back and cost 200 us; profiles/EXPERIMENTS.md, round 5.)
// (below OWN_MIN entries the library sorts with ONE workgroup-sort kernel: nothing to gain from four launches of our own)
constexpr uint32_t OWN_MIN = 1u << 16, OWN_MAX = 1u << 30;
bool own_onesweep_driver(size_t R)
{
    static const char* const env = std::getenv("STP_TILE_SORT");
    static const bool lib = env && std::strcmp(env, "rocprim") == 0;
    static const bool always = env && std::strcmp(env, "own") == 0; // (tests: also below OWN_MIN)
    return !lib && (R >= OWN_MIN || (always && R > 0)) && R < OWN_MAX;
}
#else  // another rocPRIM: the library's host function only
struct OsLayout { size_t total; };
OsLayout os_layout(char*, size_t) { return OsLayout{0}; }
bool own_onesweep_driver(size_t) { return false; }
#endif
} // namespace

size_t sort_temp_bytes(size_t R)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    R, 0u, 64u);
    const size_t own = os_layout(nullptr, R).total;
    return bytes > own ? bytes : own;
}

// what duplicate_kernel's trailing workgroups clear in front of the tile-bit sort (nothing with the library's own driver)
void sort_zero_region(const BinningState& b, size_t R, uint32_t tiles, uint32_t** ptr, size_t* words)
{
    *ptr = nullptr; *words = 0;
    if (!own_onesweep_driver(R)) return;
#if STP_OWN_ONESWEEP
    const OsLayout L = os_layout(b.sort_temp, R);
    const uint32_t bit = higher_msb(tiles);
    const size_t places = (bit + OS.radix_bits_per_place - 1) / OS.radix_bits_per_place; // (what launch_sort will run: two at 1080p, of OS_MAX_PLACES)
    const size_t bytes = L.lookback_offset + ((sizeof(rpd::onesweep_lookback_state) * places * OS_RADIX * (size_t)L.sort_blocks + 255) & ~(size_t)255);
    *ptr = reinterpret_cast<uint32_t*>(b.sort_temp);
    *words = (bytes < L.zero_bytes ? bytes : L.zero_bytes) / sizeof(uint32_t);
#else
    (void)b; (void)tiles;
#endif
}

hipError_t launch_scan(const FrameParams& f, const GeometryState& g, hipStream_t st)
{
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, (size_t)f.P, rocprim::plus<uint32_t>(), st);
}

// tile_bits_only: sort on the tile bits alone (two radix passes; the depth order inside every tile's segment is then
// established by launch_tile_sort_gather, stp_tilesort.hip); otherwise the reference's full sort on bits [0, 32 + bit).
// zeroed: duplicate_kernel has cleared sort_zero_region() for this R (the own driver's precondition)
hipError_t launch_sort(const FrameParams& f, const BinningState& b, int R, bool tile_bits_only, bool zeroed, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    const uint32_t bit = higher_msb((uint32_t)(f.gx * f.gy));
    if (!(tile_bits_only && zeroed && own_onesweep_driver((size_t)R))) {
        size_t bytes = b.sort_temp_bytes;
        return rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.point_list_unsorted, b.point_list, (size_t)R,
                                         tile_bits_only ? 32u : 0u, 32u + bit, st);
    }
#if STP_OWN_ONESWEEP
    const OsLayout L = os_layout(b.sort_temp, (size_t)R);
    const unsigned begin_bit = 32u, end_bit = 32u + bit;
    const unsigned places = (bit + OS.radix_bits_per_place - 1) / OS.radix_bits_per_place;
    const uint32_t size = (uint32_t)R;
    {   // digit histograms of all places in one pass over the keys, then one exclusive scan per place (rocPRIM: radix_sort_onesweep_global_offsets)
        const uint32_t blocks = (size + OS_HIST_ITEMS - 1) / OS_HIST_ITEMS, full_blocks = size % OS_HIST_ITEMS == 0 ? blocks : blocks - 1;
        hipLaunchKernelGGL(os_histograms_kernel, dim3(blocks), dim3(OS.histogram.block_size), 0, st, b.keys_unsorted, L.offsets, size, full_blocks, begin_bit, end_bit);
        hipLaunchKernelGGL(os_scan_histograms_kernel, dim3(places), dim3(OS.histogram.block_size), 0, st, L.offsets);
    }
    const uint32_t blocks = L.sort_blocks, full_blocks = size % OS_SORT_ITEMS == 0 ? blocks : blocks - 1;
    bool to_output = (places - 1) % 2 == 0, from_input = true; // (ping-pong through the temporaries so that the last pass lands in the output arrays)
    for (unsigned place = 0, pbit = begin_bit; place < places; place++, pbit += OS.radix_bits_per_place) {
        const uint64_t* kin = from_input ? b.keys_unsorted : (to_output ? L.keys_tmp : b.keys);
        const uint32_t* vin = from_input ? b.point_list_unsorted : (to_output ? L.values_tmp : b.point_list);
        uint64_t* kout = to_output ? b.keys : L.keys_tmp;
        uint32_t* vout = to_output ? b.point_list : L.values_tmp;
        const unsigned radix_bits = end_bit - pbit < OS.radix_bits_per_place ? end_bit - pbit : OS.radix_bits_per_place;
        hipLaunchKernelGGL(os_iteration_kernel, dim3(blocks), dim3(OS.sort.block_size), 0, st, kin, kout, vin, vout, size, L.offsets + place * OS_RADIX, L.offsets_tmp,
                           L.lookback + (size_t)place * OS_RADIX * blocks, pbit, radix_bits, full_blocks, OsBlockId::create(L.block_ids + place));
        from_input = false;
        to_output = !to_output;
    }
#endif
    return hipGetLastError();
}

// ---- binning by tile counters (STP_SORT=counters; not the default, see stp_api.hip) ----------------------------------
// The reference (and the default path) brings the duplicates into tile order with a device-wide radix sort of all R
// (key, id) pairs.  The tile of every duplicate is known when it is emitted, and a tile's segment length is known once
// every Gaussian has been preprocessed: preprocess_kernel counts entries per tile (one fire-and-forget atomic per
// duplicate), tile_scan_kernel turns the counters into the tile ranges (an exclusive prefix sum over T <= a few 10^4
// tiles: one workgroup) and duplicate_kernel writes each duplicate straight into its tile's segment through an atomic
// cursor.  The order INSIDE a segment is whatever the atomics gave; the tile's own workgroup then sorts it by
// (depth, Gaussian id), which is the order a stable sort of duplicates emitted in Gaussian order produces
// (stp_tilesort.hip) -- same sorted list, bit for bit, no device-wide sort pass, no identifyTileRanges pass.
namespace {

constexpr int SCAN_THREADS = 1024;

__global__ void __launch_bounds__(SCAN_THREADS) tile_scan_kernel(int tile0, int T, const uint32_t* __restrict__ counts_, uint2* __restrict__ ranges_,
                                                                 uint32_t* __restrict__ cursor_, uint32_t* __restrict__ total)
{
    const uint32_t* __restrict__ counts = counts_ + tile0; // the T tiles of the frame's tile-row window
    uint2* __restrict__ ranges = ranges_ + tile0;
    uint32_t* __restrict__ cursor = cursor_ + tile0;
    __shared__ uint32_t s_part[SCAN_THREADS];
    const int tid = (int)threadIdx.x;
    const int per = (T + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(tid * per, T), hi = min(lo + per, T);
    uint32_t sum = 0;
    for (int t = lo; t < hi; t++) sum += counts[t];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) { // inclusive scan of the per-thread sums
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int t = lo; t < hi; t++) {
        const uint32_t c = counts[t];
        ranges[t] = c ? make_uint2(run, run + c) : make_uint2(0u, 0u); // (an empty tile keeps the (0, 0) of the reference's memset)
        cursor[t] = run;
        run += c;
    }
    if (tid == SCAN_THREADS - 1) total[0] = s_part[tid];
}

// entries [total, R): duplicates that preprocess counted and culling then dropped (reference stopthepop_common.cuh:503-508)
__global__ void __launch_bounds__(256) bin_pad_kernel(int R, const uint32_t* __restrict__ total, uint64_t* __restrict__ keys, uint32_t* __restrict__ list)
{
    for (int i = (int)total[0] + (int)threadIdx.x; i < R; i += 256) {
        keys[i] = make_sort_key(INVALID_TILE_ID, FLT_MAX);
        list[i] = 0xFFFFFFFFu;
    }
}

} // namespace

hipError_t launch_tile_scan(const FrameParams& f, const ImageState& img, hipStream_t st)
{
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, f.gx * f.ty0, f.gx * (f.ty1 - f.ty0), img.tile_counts, img.ranges, img.tile_cursor, img.bin_total);
    return hipGetLastError();
}

hipError_t launch_bin_pad(const BinningState& b, const ImageState& img, int R, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(bin_pad_kernel, dim3(1), dim3(256), 0, st, R, img.bin_total, b.keys, b.point_list);
    return hipGetLastError();
}

} // namespace stp
