// stp_internal.h -- host-side internals of libstp_raster.so: scratch-buffer carving and the
// launcher interface between the C ABI (stp_api.hip) and the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdlib>
#include <string>

#include "../../include/stp_raster.h"

namespace stp {

constexpr int TILE = 16;
#ifndef STP_CULL_MASK
#define STP_CULL_MASK 1 // hierarchical 4x4 culling: the per-(entry, sub-tile) decisions are made by the entry gather (stp_tilesort.hip) and
                        // travel in the entry records; 0: by the render kernel's batch staging (rounds 1-2)
#endif
constexpr size_t ALIGN = 256; // sub-array alignment inside the scratch buffers

enum SortMode { MODE_GLOBAL = 0, MODE_FULL = 1, MODE_KBUFFER = 2, MODE_HIER = 3 };
enum SortOrder { ORDER_Z = 0, ORDER_DISTANCE = 1, ORDER_PTD_CENTER = 2, ORDER_PTD_MAX = 3 };

inline bool uses_blend_log(const StpSettings& s)
{
    return s.record_blend_log != 0 && s.debug_visualization == 0 && (s.sort_mode == MODE_HIER || s.sort_mode == MODE_KBUFFER);
}

// which 16-bit sub-tile mask the entry gather leaves in the spare word of an entry's colour record (stp_device.h: subtile_mask):
// 1 = the hierarchical mode's 4x4 culling decisions, 2 = the k-buffer kernel's sub-tile pre-test, 0 = none
inline int subtile_mask_kind(const StpSettings& s)
{
    if (!STP_CULL_MASK) return 0;
    if (s.sort_mode == MODE_HIER && s.hierarchical_4x4_culling) return 1;
    return s.sort_mode == MODE_KBUFFER ? 2 : 0;
}

inline bool requires_depth_along_ray(const StpSettings& s) // reference rasterizer.h:66-71
{
    return s.sort_mode != MODE_GLOBAL || s.sort_order == ORDER_PTD_CENTER || s.sort_order == ORDER_PTD_MAX;
}

// Bump allocator over one byte buffer: the counterpart of the reference's obtain()/required()
// (rasterizer_impl.h:21-27,68-75) with our own layout.  With base == nullptr it only measures.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    // STP_CARVE_SKEW=n (experiment, profiles/EXPERIMENTS.md round 6: HBM channel aliasing between the SoA arrays a kernel streams side by side):
    // n extra bytes (a multiple of ALIGN) in front of every sub-array but the first
    static size_t skew()
    {
        static const size_t v = [] { const char* e = std::getenv("STP_CARVE_SKEW"); return e ? ((size_t)std::strtoull(e, nullptr, 0) & ~(ALIGN - 1)) : (size_t)0; }();
        return v;
    }
    template <typename T> T* take(size_t count, size_t* off_out = nullptr)
    {
        off = (off + ALIGN - 1) & ~(ALIGN - 1);
        if (off) off += skew();
        if (off_out) *off_out = off;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return ((off + ALIGN - 1) & ~(ALIGN - 1)) + ALIGN; }
};

// Per-Gaussian state (reference GeometryState, rasterizer_impl.cu:175-193).  SoA, HBM-resident
// between forward and backward.
struct GeometryState {
    uint32_t* status;        // [0] num_rendered, [1] error flags   (read back once per forward)
    float* depths;           // P
    uint8_t* clamped;        // 3P
    int32_t* internal_radii; // P
    float2* rects2D;         // P
    float2* means2D;         // P
    float* cov3D;            // 6P
    float4* cov3D_inv;       // 3P, only if requires_depth_along_ray
    float4* gpack;           // 4P, only if requires_depth_along_ray: A, B, C (id slot unused), D of the entry records, packed per
                             //     Gaussian so that gather_entries_kernel reads one contiguous 64-byte line per entry
    float4* conic_opacity;   // P
    float* rgb;              // 3P
    uint32_t* tiles_touched; // P
    uint32_t* point_offsets; // P   (inclusive scan of tiles_touched; between preprocess_kernel and duplicate_kernel: inclusive inside each block of 256)
    uint32_t* block_sums;    // ceil(P / 256): entries of each preprocess workgroup, and their exclusive scan (two-level scan, stp_preprocess.hip)
    uint32_t* block_prefix;  // ceil(P / 256)
    char* scan_temp;
    size_t scan_temp_bytes;
};

struct ImageState { // reference ImageState, rasterizer_impl.cu:195-202 (ranges sized per tile, not per pixel)
    float* final_T;      // N
    uint32_t* n_contrib; // N
    uint2* ranges;       // T
    uint32_t* dbg_minmax; // 2 (debug depth visualisation: the frame's extrema, as order-preserving integers)
    // Binning by tile counters (stp_binning.hip: "atomic binning"): entries per tile counted by preprocess_kernel, the
    // next free slot of every tile's segment while duplicate_kernel fills it, and [0] = entries over all tiles.
    uint32_t* tile_counts; // T
    uint32_t* tile_cursor; // T
    uint32_t* bin_total;   // 2
    uint32_t* header;      // 4: {STP_HEADER_MAGIC_IMAGE, log_depth, ~log_depth, 0} -- the blend log's depth travels WITH the buffer (stp_api.hip: buffer headers)
    uint32_t* tile_flags; // T   0 = this tile's log is valid, 1 = its log overflowed, 0xFFFFFFFF = the forward recorded no log
    uint32_t* blend_log;  // T * 4 waves * (log_depth + spare) rows * 64 lanes of u16 (only with the blend log)
    int log_depth;        // records per pixel the log holds (0: none)
};

struct BinningState { // reference BinningState, rasterizer_impl.cu:204-217
    uint32_t* header;      // first 256 bytes of the buffer: {STP_HEADER_MAGIC_BINNING, entries the buffer was carved for, ~that, 0} (stp_api.hip: buffer headers)
    uint32_t* point_list;
    uint32_t* point_list_unsorted;
    uint64_t* keys;
    uint64_t* keys_unsorted;
    char* sort_temp;
    size_t sort_temp_bytes;
    // Per-entry data in LIST ORDER (no counterpart in the reference, which gathers by Gaussian id wherever it needs
    // an entry's data): written once per frame by gather_entries_kernel, read with unit stride by the per-pixel-sort
    // render kernels.  A = (S00 S01 S02 S11), B = (S12 S22 q.x q.y), C = (q.z mean2D.x mean2D.y id), D = conic + opacity,
    // F = (r g b .), with S = Sigma^-1 and q = Sigma^-1 (mu - cam).
    float4* entA;
    float4* entB;
    float4* entC;
    float4* entD;
    float4* entF;
};

constexpr uint32_t STP_HEADER_MAGIC_BINNING = 0x42505453u, STP_HEADER_MAGIC_IMAGE = 0x49505453u; // "STPB", "STPI"
struct NamedOffset { const char* name; size_t offset; size_t count; };

GeometryState carve_geometry(char* base, size_t P, bool with_inv, size_t* total, NamedOffset* names = nullptr, int* n_names = nullptr);
ImageState carve_image(char* base, int W, int H, int ty0, int ty1, int log_depth /* 0: no blend log */, size_t* total, NamedOffset* names = nullptr, int* n_names = nullptr); // the tile-row window's share, frame-coordinate indexing
BinningState carve_binning(char* base, size_t R, size_t* total, NamedOffset* names = nullptr, int* n_names = nullptr);

int blend_log_rows(int depth); // rows of 64 records per wave in a blend log of that depth (stp_render_replay.hip / stp_blend.h)
int blend_log_default_depth(); // depth of a frame nothing is known about
int blend_log_clamp_depth(int d);
size_t scan_temp_bytes(size_t P);
size_t sort_temp_bytes(size_t R);

// Everything a kernel launcher needs about one frame.
struct FrameParams {
    int P, D, M, W, H, gx, gy;
    int ty0, ty1; // tile-row window
    float focal_x, focal_y, tan_fovx, tan_fovy, scale_modifier;
    StpSettings s;
    const float* background;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* inv_viewprojmatrix;
    const float* cam_pos;
    int prefiltered;
    int log_depth;       // depth of this frame's blend log (forward: chosen by log_depth_for; backward: the forward's)
    uint32_t* log_need;  // forward: device word the recording kernels report their largest blend count per pixel to (or nullptr)
    uint32_t log_tag;    // ... tagged with 16 bits of the frame's kind
    int split_launch = 0; // this FrameParams describes ONE of the two render launches of a split forward (stp_set_forward_split): no tile order of the whole window
    int wild_cov; // forward, after the status read-back: some visible Gaussian has a Sigma^-1 entry >= 1e36 or not finite (depth keys then take the reciprocal with its domain check)
};

struct BackwardParams {
    const float* pixel_colors;
    const float* dL_dpix;
    float* grad_rec;    // P x grad_stride: written by the render half, read by the per-Gaussian half
    int grad_stride;    // STP_GRAD_RECORD_FLOATS, or STP_GRAD_RECORD_USED with phases bit 2 (compact records)
    int clear_rec;      // phases bit 3: the per-Gaussian half zero-fills the records again
    int chunk, chunks;  // the per-Gaussian half on chunk `chunk` of `chunks` equal ranges of 256-Gaussian blocks (chunks <= 1: all Gaussians)
    float* dL_dmean2D;  // outputs of the per-Gaussian half from here on
    float* dL_dopacity;
    float* dL_dcolor;
    float* dL_dmean3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscale;
    float* dL_drot;
};

// ---- launchers (one per stage; each returns hipSuccess or the launch error) ----
hipError_t launch_preprocess(const FrameParams& f, const GeometryState& g, int* radii, uint32_t* tile_counts, hipStream_t st); // tile_counts: nullptr = do not count per tile
hipError_t launch_frame_init(const GeometryState& g, const ImageState& img, int tile0, int n_tiles, bool with_log, bool tile_counters, hipStream_t st); // status, ranges, tile flags (+ tile counters) of the window's tiles
hipError_t launch_scan(const FrameParams& f, const GeometryState& g, hipStream_t st);
// Longest-list-first order of the window's tiles for the render kernels' workgroups (into ImageState::tile_cursor, free outside STP_SORT=counters)
bool tile_order_enabled();
// ... and does THIS frame use it: a window of up to 1024 tiles is resident on the chip all at once (1024-1280 workgroup slots), an order of its
// workgroups changes nothing and the 7 us kernel is left out (C1: 256 tiles)
inline bool tile_order_used(const FrameParams& f) { return tile_order_enabled() && f.gx * (f.ty1 - f.ty0) > 1024; }
bool gather_order_enabled();
int gather_order_mode();
hipError_t launch_tile_order(const FrameParams& f, const ImageState& img, hipStream_t st);
hipError_t launch_sh_color(const FrameParams& f, const GeometryState& g, const int* radii, hipStream_t st); // SH -> RGB of the visible Gaussians (after preprocess)
hipError_t launch_mailbox(const uint32_t* last_offset, const uint32_t* status, uint32_t* mailbox_dev, uint32_t ticket, uint32_t* log_need, hipStream_t st);
hipError_t launch_block_prefix_mailbox(const FrameParams& f, const GeometryState& g, uint32_t* mailbox_dev, uint32_t ticket, uint32_t* log_need, hipStream_t st); // second level of the scan + the hand-over
hipError_t launch_duplicate(const FrameParams& f, const GeometryState& g, const int* radii, const BinningState& b, uint32_t* tile_cursor, uint32_t cap, uint32_t header_cap, uint32_t* zero_ptr, size_t zero_words, hipStream_t st); // header_cap: entries the buffer was carved for (its header); tile_cursor: nullptr = by point_offsets into the unsorted arrays; cap: slots of a run-ahead launch (0xFFFFFFFF = exact)
hipError_t launch_tile_scan(const FrameParams& f, const ImageState& img, hipStream_t st);
hipError_t launch_bin_pad(const BinningState& b, const ImageState& img, int R, hipStream_t st);
hipError_t launch_sort(const FrameParams& f, const BinningState& b, int R, bool tile_bits_only, bool zeroed, hipStream_t st); // zeroed: duplicate_kernel cleared sort_zero_region()
void sort_zero_region(const BinningState& b, size_t R, uint32_t tiles, uint32_t** ptr, size_t* words); // what the tile-bit sort's own driver needs cleared in front of it
hipError_t launch_tile_sort_gather(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img, int R, bool unordered, hipStream_t st);
hipError_t launch_ranges(const FrameParams& f, const BinningState& b, const ImageState& img, int R, hipStream_t st);
hipError_t launch_gather_entries(const FrameParams& f, const GeometryState& g, const BinningState& b, int R, hipStream_t st);
hipError_t launch_render_forward(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img,
                                 float* out_color, hipStream_t st, std::string* err);
hipError_t launch_render_debug_finish(const FrameParams& f, const ImageState& img, float* out_color, hipStream_t st);
hipError_t launch_render_backward(const FrameParams& f, const GeometryState& g, const BinningState& b, const ImageState& img,
                                  const BackwardParams& bw, hipStream_t st, std::string* err);
hipError_t launch_preprocess_backward(const FrameParams& f, const GeometryState& g, const int* radii, const BackwardParams& bw, hipStream_t st);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t st);

uint32_t higher_msb(uint32_t n);

} // namespace stp
