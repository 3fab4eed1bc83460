// stp_binning.hip -- prefix sum of tile counts and the stable (tile, depth) radix sort.
//
// Replaces cub::DeviceScan::InclusiveSum (reference rasterizer_impl.cu:189,313) and
// cub::DeviceRadixSort::SortPairs on key bits [0, 32+bit) (rasterizer_impl.cu:211-214,344-352) with
// rocPRIM's device-wide primitives (both are stable LSD radix sorts, so the sorted list is the
// same list).
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

size_t sort_temp_bytes(size_t R)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    R, 0u, 64u);
    return bytes;
}

hipError_t launch_scan(const FrameParams& f, const GeometryState& g, hipStream_t st)
{
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, (size_t)f.P, rocprim::plus<uint32_t>(), st);
}

// tile_bits_only: sort on the tile bits alone (two radix passes; the depth order inside every tile's segment is then
// established by launch_tile_sort_gather, stp_tilesort.hip); otherwise the reference's full sort on bits [0, 32 + bit).
hipError_t launch_sort(const FrameParams& f, const BinningState& b, int R, bool tile_bits_only, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    const uint32_t bit = higher_msb((uint32_t)(f.gx * f.gy));
    size_t bytes = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.point_list_unsorted, b.point_list, (size_t)R,
                                     tile_bits_only ? 32u : 0u, 32u + bit, st);
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
