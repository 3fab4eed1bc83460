/*
 * oracle/stp_oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported host baseline.  The product (libstp_raster.so) never links,
 * loads or falls back to it.
 *
 * PARITY PIN: the reference ships no tests, golden vectors or fixtures for this path, and it is CUDA.  It is pinned
 * against THE REFERENCE ITSELF compiled for gfx950: oracle/ref_build/build_ref.sh runs the ROCm image's own hipify-perl
 * over the reference's cuda_rasterizer sources (where they lie under /root/reference) and compiles them with hipcc
 * (-ffp-contract=off) into oracle/_ref/libstp_ref_ieee.so; tests/golden/ref/ holds that library's outputs (generated on
 * an MI355X by tests/golden/make_ref_golden.py) and tests/test_reference_golden.py holds this oracle to them BIT FOR BIT
 * in every integer / index result, per-Gaussian state array, sort key, sorted list and tile range, in every sort mode.
 * What the pin is NOT: a run of the reference under nvcc on NVIDIA hardware (none exists here).  PER_PIXEL_FULL is not
 * covered (the reference's kernel faults on gfx950).  See DESIGN.md section 5.
 */
#ifndef STP_ORACLE_H_INCLUDED
#define STP_ORACLE_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors CudaRasterizer::SplattingSettings (reference cuda_rasterizer/rasterizer.h:27-135). */
typedef struct OrcSettings {
    int32_t sort_mode;               /* 0 GLOBAL, 1 PPX_FULL, 2 PPX_KBUFFER, 3 HIER */
    int32_t sort_order;              /* 0 Z_DEPTH, 1 DISTANCE, 2 PTD_CENTER, 3 PTD_MAX */
    int32_t queue_tile_4x4;          /* parsed, unused (reference: tail is hard-wired to 64) */
    int32_t queue_tile_2x2;          /* MID queue size (8, 12, 20) */
    int32_t queue_per_pixel;         /* HEAD queue size / k-buffer window */
    int32_t rect_bounding;
    int32_t tight_opacity_bounding;
    int32_t tile_based_culling;
    int32_t hierarchical_4x4_culling;
    int32_t load_balancing;          /* performance hint only: results identical */
    int32_t proper_ewa_scaling;
    /* our own extension (north_star tile-row sharding): only tile rows [tile_y0, tile_y1) are
       binned and rendered.  tile_y1 <= 0 means "all rows". */
    int32_t tile_y0;
    int32_t tile_y1;
    /* 0 = normal image; 1 = DebugVisualization::Depth (what `render_depth=True` selects, ref: rasterize_points.cu:104-107):
       sum(depth * alpha * T) per pixel, normalised by the frame's extrema, Turbo colormap (forward only) */
    int32_t debug_visualization;
} OrcSettings;

typedef struct OrcFrame OrcFrame; /* opaque: forward state kept for backward / inspection */

/* Forward.  Returns num_rendered (>= 0) or a negative error code.  All pointers are host memory.
   shs/colors_precomp/scales/rotations/cov3D_precomp may be NULL as in the reference
   (rasterize_points.cu:109-135).  *frame_out receives a handle to be released with
   orc_frame_free. */
int orc_forward(int P, int D, int M, const float* background, int W, int H,
                const OrcSettings* settings,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, int* radii, OrcFrame** frame_out);

/* Backward (reference rasterizer_impl.cu:417-526).  Gradient outputs must be zero-initialised
   by the caller, shapes as in rasterize_points.cu:178-186. */
int orc_backward(const OrcFrame* frame, const float* background,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy,
                 const float* pixel_colors, const float* dL_dpix,
                 float* dL_dmean2D /*P*3*/, float* dL_dconic /*P*4*/, float* dL_dopacity /*P*/,
                 float* dL_dcolor /*P*3*/, float* dL_dmean3D /*P*3*/, float* dL_dcov3D /*P*6*/,
                 float* dL_dsh /*P*M*3*/, float* dL_dscale /*P*3*/, float* dL_drot /*P*4*/);

void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present);

void orc_frame_free(OrcFrame* frame);

/* Inspection of intermediate state, for stage-by-stage parity tests.
   name in: depths, clamped(u8), radii(i32), rects2D, means2D, cov3D, cov3D_inv, conic_opacity, rgb,
   tiles_touched(u32), point_offsets(u32), keys_unsorted(u64), values_unsorted(u32), keys(u64),
   point_list(u32), ranges(u32 pairs), final_T, n_contrib(u32).
   Returns element count (in units of the scalar type) or -1. */
int64_t orc_frame_array(const OrcFrame* frame, const char* name, const void** data);
int orc_frame_num_rendered(const OrcFrame* frame);
/* Diagnostic: opacity * exp(-power) of the hierarchical 4x4-culling test, in double, for every entry of `tile` against the
   sub-tile with corner pixel (cx, cy): the distance of each culling decision from the 1/255 threshold. */
int orc_cull_alpha(const OrcFrame* frame, int tile, int cx, int cy, double* out, int cap);

/* Test-only switches.  "ewa_exact_grad"=1: use the mathematically exact Mip-Splatting scaling
   gradient instead of the reference's (see stp_oracle.cpp backward_preprocess).
   "ieee_depth": 1 (default) = evaluate depthAlongRay without fused multiply-adds, as the -ffp-contract=off build of the
   reference itself (oracle/_ref/libstp_ref_ieee.so) and the default product library do; 0 = the fma order of the
   second product library, libstp_raster_fma.so.
   "lazy_pop"=1: in the hierarchical head level and the k-buffer, a candidate that fails its tests causes no
   pop-before-look -- images, final_T and gradients must not change (tests/test_oracle_cpu.py); default 0 = the reference. */
void orc_set_flag(const char* name, int value);

/* Number of OpenMP threads the oracle will use (for bench.py's cpu_baseline.cores). */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
