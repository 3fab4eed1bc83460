// stp_rasterizer.hpp -- C++ face of libstp_raster.so for code written against the reference's static C++ API
// (the SIBR viewer and other C++ users of `CudaRasterizer::Rasterizer`, reference cuda_rasterizer/rasterizer.h:184-258,
// CMakeLists.txt:22-36).  Header-only, C++17, no HIP or torch headers: everything goes through the C ABI of
// stp_raster.h.  Link with -lstp_raster.
//
// What maps to what (reference file:line):
//   StpRasterizer::SortMode / GlobalSortOrder / SortQueueSizes / SortSettings / CullingSettings / SplattingSettings
//                                            <- rasterizer.h:27-135 (same member names, same defaults, same helpers)
//   StpRasterizer::DebugVisualization(Data)  <- stopthepop/rasterizer_debug.h:11-57 (Depth and Disabled are implemented;
//                                               the viewer-only types throw)
//   StpRasterizer::Rasterizer::markVisible   <- rasterizer.h:188-193
//   StpRasterizer::Rasterizer::forward       <- rasterizer.h:195-220 (same argument order; one trailing `stream`)
//   StpRasterizer::Rasterizer::backward      <- rasterizer.h:222-257 (same argument order; the dL_dconic scratch tensor
//                                               is replaced by grad_records, P x 16 floats, zero-filled by the caller;
//                                               one trailing `stream`)
// Errors: the reference throws std::runtime_error from its CHECK_CUDA macro and from the dispatchers
// ("Not supported head/mid queue size", "Backward not supported for full per-pixel sort", ...); so does this header,
// with the message of stp_last_error().
#ifndef STP_RASTERIZER_HPP_INCLUDED
#define STP_RASTERIZER_HPP_INCLUDED

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "stp_raster.h"

namespace StpRasterizer {

enum SortMode { GLOBAL = 0, PER_PIXEL_FULL = 1, PER_PIXEL_KBUFFER = 2, HIERARCHICAL = 3 };

enum GlobalSortOrder { VIEWSPACE_Z = 0, DISTANCE = 1, PER_TILE_DEPTH_CENTER = 2, PER_TILE_DEPTH_MAXPOS = 3 };

struct SortQueueSizes {
    int tile_4x4 = 64;
    int tile_2x2 = 8;
    int per_pixel = 4;
};

// the queue sizes the library is built with (rasterizer.h:51-60)
static const std::vector<int> per_pixel_queue_sizes{1, 2, 4, 8, 12, 16, 20, 24};
static const std::vector<int> twobytwo_tile_queue_sizes{8, 12, 20};
static const std::vector<int> per_pixel_queue_sizes_hier{4, 8, 16};

struct SortSettings {
    SortMode sort_mode = SortMode::GLOBAL;
    GlobalSortOrder sort_order = GlobalSortOrder::VIEWSPACE_Z;
    SortQueueSizes queue_sizes;

    bool requiresDepthAlongRay() const
    {
        return sort_mode != SortMode::GLOBAL || sort_order == PER_TILE_DEPTH_CENTER || sort_order == PER_TILE_DEPTH_MAXPOS;
    }
    bool hasModifiableWindowSize() const { return sort_mode == HIERARCHICAL || sort_mode == PER_PIXEL_KBUFFER; }
};

struct CullingSettings {
    bool rect_bounding = false;
    bool tight_opacity_bounding = false;
    bool tile_based_culling = false;
    bool hierarchical_4x4_culling = false;
};

struct SplattingSettings {
    SortSettings sort_settings;
    CullingSettings culling_settings;
    bool load_balancing = false;
    bool proper_ewa_scaling = false;
};

inline std::string toString(SortMode m)
{
    static const char* const names[] = {"GLOBAL", "FULL SORT", "KBUFFER", "HIERARCHICAL"};
    return (m >= GLOBAL && m <= HIERARCHICAL) ? names[m] : "";
}
inline std::string toString(GlobalSortOrder m)
{
    static const char* const names[] = {"VIEWSPACE_Z", "DISTANCE", "PER_TILE_DEPTH_CENTER", "PER_TILE_DEPTH_MAXPOS"};
    return (m >= VIEWSPACE_Z && m <= PER_TILE_DEPTH_MAXPOS) ? names[m] : "";
}
inline bool isInvalidSortMode(int m) { return m < GLOBAL || m > HIERARCHICAL; }
inline bool isInvalidSortOrder(int m) { return m < VIEWSPACE_Z || m > PER_TILE_DEPTH_MAXPOS; }

enum class DebugVisualization { SortErrorOpacity, SortErrorDistance, GaussianCountPerTile, GaussianCountPerPixel, Depth, Transmittance, Disabled };

inline std::string toString(DebugVisualization m)
{
    switch (m) {
    case DebugVisualization::SortErrorOpacity: return "Sort Error: Opacity";
    case DebugVisualization::SortErrorDistance: return "Sort Error: Distance";
    case DebugVisualization::GaussianCountPerTile: return "Gaussian Count Per Tile";
    case DebugVisualization::GaussianCountPerPixel: return "Gaussian Count Per Pixel";
    case DebugVisualization::Depth: return "Depth";
    case DebugVisualization::Transmittance: return "Transmittance";
    default: return "Disabled";
    }
}

// rasterizer_debug.h:43-57.  Honoured: `type` (Depth / Disabled) and `timing_enabled` (-> `timings_text` after every
// forward).  debugPixel / dataCallback / minMax / debug_normalize belong to the viewer-only visualisations and are ignored.
struct DebugVisualizationData {
    DebugVisualization type{DebugVisualization::Disabled};
    int debugPixel[2] = {};
    std::function<void(const DebugVisualizationData&, float, float, float, float, float)> dataCallback{
        [](const DebugVisualizationData&, float, float, float, float, float) {}};
    float minMax[2] = {0.f, 10000.f};
    bool debug_normalize = false;
    std::string timings_text = "";
    bool timing_enabled = false;
};

// SplattingSettings -> the POD the C ABI takes (the job of from_json, rasterizer.h:160-182, one level down)
inline StpSettings toPod(const SortSettings& sort, const CullingSettings& cull, bool load_balancing, bool proper_ewa_scaling)
{
    StpSettings s{};
    s.sort_mode = (int32_t)sort.sort_mode;
    s.sort_order = (int32_t)sort.sort_order;
    s.queue_tile_4x4 = sort.queue_sizes.tile_4x4;
    s.queue_tile_2x2 = sort.queue_sizes.tile_2x2;
    s.queue_per_pixel = sort.queue_sizes.per_pixel;
    s.rect_bounding = cull.rect_bounding;
    s.tight_opacity_bounding = cull.tight_opacity_bounding;
    s.tile_based_culling = cull.tile_based_culling;
    s.hierarchical_4x4_culling = cull.hierarchical_4x4_culling;
    s.load_balancing = load_balancing;
    s.proper_ewa_scaling = proper_ewa_scaling;
    return s;
}
inline StpSettings toPod(const SplattingSettings& s)
{
    return toPod(s.sort_settings, s.culling_settings, s.load_balancing, s.proper_ewa_scaling);
}

class Rasterizer {
    static void* allocTrampoline(void* user, size_t bytes)
    {
        return (*static_cast<std::function<char*(size_t)>*>(user))(bytes);
    }
    static void check(int rc)
    {
        if (rc < 0) throw std::runtime_error(stp_last_error());
    }

public:
    static void markVisible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present, void* stream = nullptr)
    {
        static_assert(sizeof(bool) == 1, "present is P bytes");
        check(stp_mark_visible(P, means3D, viewmatrix, projmatrix, reinterpret_cast<uint8_t*>(present), stream));
    }

    // Returns num_rendered.  `recordBlendLog`: set it when a backward with the same settings will follow (training) --
    // the forward then writes the per-pixel blend order into the image buffer and the backward replays it (DESIGN.md
    // section 3.1); the buffers returned by the three callbacks must stay valid until that backward has run.
    static int forward(std::function<char*(size_t)> geometryBuffer, std::function<char*(size_t)> binningBuffer,
                       std::function<char*(size_t)> imageBuffer, const int P, int D, int M, const float* background,
                       const int width, int height, const SplattingSettings splatting_settings,
                       DebugVisualizationData& debugVisualization, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales, const float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* inv_viewprojmatrix, const float* cam_pos, const float tan_fovx, float tan_fovy,
                       const bool prefiltered, float* out_color, int* radii = nullptr, bool debug = false,
                       void* stream = nullptr, bool recordBlendLog = false)
    {
        StpSettings s = toPod(splatting_settings);
        s.record_blend_log = recordBlendLog ? 1 : 0;
        switch (debugVisualization.type) {
        case DebugVisualization::Disabled: break;
        case DebugVisualization::Depth: s.debug_visualization = STP_DEBUG_DEPTH; s.record_blend_log = 0; break;
        default: throw std::runtime_error("Debug visualization '" + toString(debugVisualization.type) + "' is not supported by libstp_raster");
        }
        if (debugVisualization.timing_enabled) stp_timing_enable(1); // (restarts the running means: the text is per call, rasterizer_impl.cu:391-399)
        const int rendered = stp_forward(allocTrampoline, &geometryBuffer, allocTrampoline, &binningBuffer, allocTrampoline,
                                         &imageBuffer, P, D, M, background, width, height, &s, means3D, shs, colors_precomp,
                                         opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                         inv_viewprojmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color, radii,
                                         debug ? 1 : 0, stream);
        if (debugVisualization.timing_enabled) {
            std::string text(stp_timing_text(nullptr, 0) + 1, '\0');
            text.resize(stp_timing_text(text.data(), text.size()));
            debugVisualization.timings_text = text;
            stp_timing_enable(0);
        }
        check(rendered);
        return rendered;
    }

    // `grad_records` (P x STP_GRAD_RECORD_FLOATS floats, zero-filled) stands where the reference takes dL_dconic.
    // `replayBlendLog` must equal the forward's recordBlendLog.
    static void backward(const int P, int D, int M, int R, const float* background, const int width, int height,
                         const SortSettings sort_settings, const CullingSettings culling_settings,
                         const bool proper_ewa_scaling, const float* means3D, const float* shs, const float* opacities,
                         const float* colors_precomp, const float* scales, const float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* inv_viewprojmatrix, const float* cam_pos, const float tan_fovx, float tan_fovy,
                         const float* pixel_colors, const int* radii, char* geom_buffer, char* binning_buffer,
                         char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* grad_records, float* dL_dopacity,
                         float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                         float* dL_drot, bool debug, void* stream = nullptr, bool replayBlendLog = false)
    {
        StpSettings s = toPod(sort_settings, culling_settings, false, proper_ewa_scaling);
        s.record_blend_log = replayBlendLog ? 1 : 0;
        check(stp_backward(P, D, M, R, background, width, height, &s, means3D, shs, opacities, colors_precomp, scales,
                           scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, cam_pos,
                           tan_fovx, tan_fovy, pixel_colors, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                           dL_dmean2D, grad_records, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                           debug ? 1 : 0, stream));
    }
};

} // namespace StpRasterizer

#endif
