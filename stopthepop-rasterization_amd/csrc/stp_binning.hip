// stp_binning.hip -- prefix sum of tile counts and the stable (tile, depth) radix sort.
//
// Replaces cub::DeviceScan::InclusiveSum (reference rasterizer_impl.cu:189,313) and
// cub::DeviceRadixSort::SortPairs on key bits [0, 32+bit) (rasterizer_impl.cu:211-214,344-352) with
// rocPRIM's device-wide primitives (both are stable LSD radix sorts, so the sorted list is the
// same list).
#include <cstring> // rocPRIM 7.2's texture_cache_iterator.hpp uses unqualified memset
#include <rocprim/rocprim.hpp>

#include "stp_internal.h"

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

} // namespace stp
