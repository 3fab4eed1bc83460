/*
 * stp_raster.h -- C ABI of libstp_raster.so, the MI355X (gfx950) sorted Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the hot path of r4dl/StopThePop-Rasterization.  Each entry point
 * replaces one member of the reference's C++ API `CudaRasterizer::Rasterizer`
 * (reference cuda_rasterizer/rasterizer.h:184-258), which is what the reference's pybind layer
 * (rasterize_points.cu:43-253, ext.cpp:15-19) calls.  Signatures are plain C: device pointers, sizes,
 * a POD settings struct, allocator callbacks and a stream handle -- no torch, no C++ types.
 *
 * Conventions (identical to the reference unless stated):
 *   - all array arguments are DEVICE pointers to contiguous fp32/int32 data; an absent optional
 *     input (shs / colors_precomp / scales / rotations / cov3D_precomp) is NULL
 *     (reference tests pointers against nullptr: forward.cu:126,200; rasterizer_impl.cu:367,473,500);
 *   - matrices are 16 floats in the 3DGS row-vector layout (p_hom = [x y z 1] @ M), i.e. element
 *     m[4*col+row] in kernel indexing (auxiliary.h:103-149);
 *   - `stream` is a hipStream_t (NULL = the null stream).  The reference launches on the legacy
 *     default stream; we launch everything on the stream the caller passes;
 *   - functions return >= 0 on success and a negative StpStatus on failure; stp_last_error()
 *     returns a thread-local message for the most recent failure.
 */
#ifndef STP_RASTER_H_INCLUDED
#define STP_RASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STP_ABI_VERSION 7
#define STP_GRAD_RECORD_FLOATS 16 /* floats per Gaussian in grad_records (see stp_backward) */
#define STP_GRAD_RECORD_USED 9    /* of which these carry data; the record stride of stp_backward_phases' compact form (phases bit 2) */

/* Replaces CudaRasterizer::SplattingSettings + SortSettings + SortQueueSizes + CullingSettings
   (rasterizer.h:27-135) and their json parser (rasterizer.h:160-182): the host binding fills this
   from the Python settings dict. */
typedef struct StpSettings {
    int32_t sort_mode;                /* 0 GLOBAL, 1 PER_PIXEL_FULL, 2 PER_PIXEL_KBUFFER, 3 HIERARCHICAL (rasterizer.h:27-33) */
    int32_t sort_order;               /* 0 VIEWSPACE_Z, 1 DISTANCE, 2 PER_TILE_DEPTH_CENTER, 3 PER_TILE_DEPTH_MAXPOS (:35-41) */
    int32_t queue_tile_4x4;           /* parsed, unused (the reference hard-wires 64: hierarchical_render.cuh:286-287) */
    int32_t queue_tile_2x2;           /* hierarchical MID queue: 8, 12 or 20 (rasterizer.h:56) */
    int32_t queue_per_pixel;          /* hierarchical HEAD queue (4, 8, 16; backward also 12) or k-buffer window */
    int32_t rect_bounding;            /* rasterizer.h:79-85 */
    int32_t tight_opacity_bounding;
    int32_t tile_based_culling;
    int32_t hierarchical_4x4_culling;
    int32_t load_balancing;           /* performance hint only; results do not depend on it */
    int32_t proper_ewa_scaling;
    /* Extension (not in the reference): restrict binning + rendering to tile rows [tile_y0, tile_y1)
       for tile-row sharding of one frame over several GPUs.  tile_y1 <= 0 selects all rows. */
    int32_t tile_y0;
    int32_t tile_y1;
    /* Extension (not in the reference): hierarchical mode only.  When non-zero the forward also records, per
       pixel, the order in which it blended its Gaussians (2 bytes per blended pair, 512 B per pixel, inside the
       image buffer); a backward called with the same flag replays that log instead of re-running the resort
       (the re-sorting backward still handles tiles whose log overflowed).  Results are the same sums in a
       different order.  Set it for training forwards; leave it 0 for inference. */
    int32_t record_blend_log;
    /* Replaces DebugVisualizationData::type (stopthepop/rasterizer_debug.h:11-21; rasterizer.h:203): 0 = disabled,
       STP_DEBUG_DEPTH = the depth visualisation the Python flag `render_depth` selects (rasterize_points.cu:104-107):
       the render kernels accumulate depth * alpha * T per pixel, the frame's minimum / maximum normalise it and the
       Turbo colormap turns it into the output image (forward.cu:674-729).  Forward only. */
    int32_t debug_visualization;
} StpSettings;

#define STP_DEBUG_DISABLED 0
#define STP_DEBUG_DEPTH 1

typedef enum StpStatus {
    STP_OK = 0,
    STP_ERR_INVALID_ARGUMENT = -1,
    STP_ERR_NEEDS_SCALE_ROTATION = -2,  /* sorted modes build Sigma^-1 from scales+rotations (forward.cu:208-220) */
    STP_ERR_QUEUE_SIZE = -3,            /* "Not supported head/mid queue size" (forward.cu:455-480, backward.cu:751-760) */
    STP_ERR_SORT_MODE = -4,
    STP_ERR_NO_BACKWARD = -5,           /* "Backward not supported for full per-pixel sort" (backward.cu:733-736) */
    STP_ERR_HIP = -6,                   /* a HIP runtime call or kernel failed; message has the detail */
    STP_ERR_ALLOC = -7,                 /* an allocator callback returned NULL */
    STP_ERR_PREFILTERED = -8            /* a point was culled although prefiltered was set (auxiliary.h:228-232) */
} StpStatus;

/* Replaces the three `std::function<char*(size_t)>` buffer-resize callbacks of Rasterizer::forward
   (rasterizer.h:196-198; rasterize_points.cu:33-41).  Must return a device pointer to at least
   `bytes` bytes, valid until the matching backward has run.
   geometry_alloc and image_alloc are called once per forward.  binning_alloc may be called TWICE: once before the
   num_rendered hand-over with a size guessed from the previous frame of the same kind on this device (P, resolution,
   tile-row window, sort mode; count + 12.5 %), and again with the exact -- larger -- size only when the guess was too
   small.  The SECOND block replaces the first (resize semantics, as the reference's callback has): an arena / bump
   allocator must be prepared to take the first block back, and the buffer it finally hands to stp_backward is the
   one of the LAST call.  A guessed request can exceed the exact size by up to 12.5 % (+ the change of num_rendered from
   frame to frame); STP_BINNING=exact in the environment restores the reference's single exact request. */
typedef void* (*stp_alloc_fn)(void* user, size_t bytes);

/* Replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:195-220, rasterizer_impl.cu:221-413).
   Returns num_rendered (the number of (tile, Gaussian) duplicates) or a negative StpStatus.
   Contains exactly one host synchronisation (the read-back of num_rendered, as in
   rasterizer_impl.cu:317).  `radii` may be NULL (an internal array is used, rasterizer_impl.cu:259-262).
   `out_color` is (3,H,W), written for every pixel of the selected tile rows. */
int stp_forward(stp_alloc_fn geometry_alloc, void* geometry_user,
                stp_alloc_fn binning_alloc, void* binning_user,
                stp_alloc_fn image_alloc, void* image_user,
                int P, int D, int M,
                const float* background, int width, int height,
                const StpSettings* settings,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, int* radii, int debug, void* stream);

/* Extension (not in the reference), for tile-row sharding: the NEXT stp_forward of the calling thread launches its render kernel twice -- tile
   rows [tile_y0, tile_row) first, then [tile_row, tile_y1) -- and records `event` (a hipEvent_t of the caller) on the stream between the two
   launches: when the event has fired, the pixel rows above 16 * tile_row are final, and a rank can put them on the wire while the rest of its
   strip is still being blended.  Same kernels and per-tile work as one launch: pixels, buffers and the backward are unchanged.  A tile_row
   outside (tile_y0, tile_y1), or a call that returns early, records the event behind everything the call enqueued.  event == NULL clears a
   pending request.  The request is consumed by the next stp_forward whatever its outcome. */
void stp_set_forward_split(int tile_row, void* event);

/* Replaces CudaRasterizer::Rasterizer::backward (rasterizer.h:222-257, rasterizer_impl.cu:417-526).
   geom/binning/image buffers are the ones the forward allocated; R is the forward's return value.
   grad_records must be zero-filled by the caller.  The dL_d* outputs need not be (the reference zero-fills them,
   rasterize_points.cu:178-186): every row of every output is written, zeros for Gaussians outside the frustum --
   except dL_dscale / dL_drot when cov3D_precomp is given (then they are left untouched, as in the reference).

   grad_records (P x STP_GRAD_RECORD_FLOATS floats) is the hand-over between the two halves of the backward.
   It takes the place of the reference's dL_dconic scratch tensor (rasterize_points.cu:181): the render half sums
   its nine per-Gaussian terms into ONE 64-byte record per Gaussian,
       [0..2] dL/dcolour rgb   [3..4] dL/dmean2D xy   [5..7] dL/dconic xx, xy, yy   [8] dL/dopacity   [9..15] unused
   (one atomic instruction / one L2 request per flush instead of nine into four arrays: 9x the flush rate on
   MI355X, tools/global_atomic_bench.hip); the per-Gaussian half reads the record and writes dL_dmean2D,
   dL_dopacity and dL_dcolor in the reference's layouts along with the remaining gradients. */
int stp_backward(int P, int D, int M, int R,
                 const float* background, int width, int height,
                 const StpSettings* settings,
                 const float* means3D, const float* shs, const float* opacities, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy,
                 const float* pixel_colors, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer,
                 const float* dL_dpix,
                 float* dL_dmean2D /* P x 3 */, float* grad_records /* P x 16 */, float* dL_dopacity /* P */,
                 float* dL_dcolor /* P x 3 */, float* dL_dmean3D /* P x 3 */, float* dL_dcov3D /* P x 6 */,
                 float* dL_dsh /* P x M x 3 */, float* dL_dscale /* P x 3 */, float* dL_drot /* P x 4 */,
                 int debug, void* stream);

/* Extension (not in the reference): the two halves of the backward separately, for tile-row sharding.
   phases bit 0 = BACKWARD::render (rasterizer_impl.cu:474-495): accumulates the per-Gaussian partial
   sums of THIS rank's tile rows into grad_records (nothing else is written);
   phases bit 1 = BACKWARD::preprocess (rasterizer_impl.cu:501-525): consumes grad_records (after the
   caller has summed them across ranks) and writes every dL_d* output.  phases = 3 == stp_backward.
   phases bit 2 (value 4, with bit 0 and / or bit 1) = COMPACT records: grad_records is P x STP_GRAD_RECORD_USED floats (36 bytes per
   Gaussian, no padding) -- the buffer a tile-row shard all-reduces between the two halves crosses xGMI as it is, without a
   pack / unpack copy on either side.  (The padded 64-byte record is what a single GPU wants: one line, one atomic request per flush.)
   phases bit 3 (value 8, with bit 1) = the per-Gaussian half leaves grad_records ZERO-FILLED again: it reads every record the render
   half can have written (those of the visible Gaussians) and clears it behind the read, so that a caller who keeps the buffer between
   steps never zero-fills it after the first time (the fill is 64 B per Gaussian per step otherwise: 24 us at 1M, 0.14 ms at 6M).
   phases bits 8-15 = K, bits 16-23 = k (with bit 1 only): the per-Gaussian half on chunk k of K equal ranges of Gaussian ids (ranges of
   256-Gaussian blocks).  Gaussians are independent there, so a tile-row shard all-reduces its records in K pieces and runs chunk k as soon as
   piece k has arrived, while piece k + 1 is still on the links (tile_shard.py).  K <= 1: all Gaussians. */
int stp_backward_phases(int phases, int P, int D, int M, int R,
                        const float* background, int width, int height,
                        const StpSettings* settings,
                        const float* means3D, const float* shs, const float* opacities, const float* colors_precomp,
                        const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy,
                        const float* pixel_colors, const int* radii,
                        char* geom_buffer, char* binning_buffer, char* image_buffer,
                        const float* dL_dpix,
                        float* dL_dmean2D, float* grad_records, float* dL_dopacity, float* dL_dcolor,
                        float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                        int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:188-193, rasterizer_impl.cu:161-173).
   `present` is P bytes (bool). */
int stp_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Sizes of the three scratch buffers (the reference's `required<State>()`, rasterizer_impl.h:68-75). */
size_t stp_geometry_buffer_size(int P, const StpSettings* settings);
size_t stp_binning_buffer_size(int R);
size_t stp_image_buffer_size(int width, int height); /* without the optional blend log */
/* Bytes the blend log adds to the image buffer of a forward with StpSettings.record_blend_log = 1 in the hierarchical / k-buffer modes,
   for a frame nothing is known about yet: 192 records of 2 bytes per pixel of the 16x16 tile grid + eight spare rows = 400 B per pixel.
   The depth is ADAPTIVE: every recording forward reports the largest number of blends of any of its pixels, and the next forwards of
   the same kind (P, resolution, tile-row window, mode) on the device size their log by it (+12.5 %, multiple of 16, 32..512 records) --
   304 B per pixel once a frame like BASELINE's C2 (at most 114 blends per pixel) has been seen, 464 B for C5 (195).  A pixel that blends
   more than its frame's depth flags its tile, whose backward then re-sorts (correct, slow), and the frames after it get the deeper log.
   STP_LOG_DEPTH=n in the environment fixes the depth.  _rows: of a forward restricted to the tile rows [tile_y0, tile_y1). */
size_t stp_blend_log_bytes(int width, int height);
size_t stp_blend_log_bytes_rows(int width, int height, int tile_y0, int tile_y1);
/* The depth (records per pixel) of the blend log in an image buffer a forward filled (0: it recorded none).  Every forward leaves a header in
   its image and binning buffers (their first 256 bytes: magic, value, ~value), so the buffers are self-contained like the reference's: the
   library keeps a pointer -> value cache for the buffers of this process and reads the header back (one blocking 16-byte copy) for a pointer it
   does not know -- a clone, a copy, a buffer of another process.  Negative (STP_ERR_INVALID_ARGUMENT) for a buffer without a valid header. */
int stp_blend_log_depth(const void* image_buffer);
/* Drops the library's cached layout of a binning / image buffer at this address.  An allocator that recycles scratch buffers calls it when
   it releases or frees one, so that a later tenant of the address -- a clone or a restored copy of ANOTHER forward's buffer, which no forward
   of this process carved there -- is resolved from the header it carries instead of the previous tenant's entry.  (A forward that carves the
   address again overwrites the entry by itself.)  Harmless for unknown pointers. */
void stp_forget_buffer(const void* buffer);
/* Bytes of a blend log of `depth` records per pixel (depth <= 0: of the DEEPEST log a forward may carve, 512 records) for the tile rows
   [tile_y0, tile_y1) (0, 0 = the whole frame): what a memory policy should budget for a frame whose depth it does not know yet. */
size_t stp_blend_log_bytes_depth(int width, int height, int tile_y0, int tile_y1, int depth);

/* Introspection of the (otherwise opaque) scratch buffers, for parity tests and debugging.
   Fills byte offset and element count of a named sub-array; returns 0 or STP_ERR_INVALID_ARGUMENT.
   geometry names: depths clamped radii rects2D means2D cov3D cov3D_inv conic_opacity rgb tiles_touched point_offsets
   binning names : point_list point_list_unsorted keys keys_unsorted
   image names   : final_T n_contrib ranges */
int stp_geometry_layout(int P, const StpSettings* settings, const char* name, size_t* offset, size_t* count);
int stp_binning_layout(int R, const char* name, size_t* offset, size_t* count);
/* The entry count a binning buffer was carved with by the forward that last used it: num_rendered, or -- for a run-ahead forward, which
   carves and launches before num_rendered is known -- the capacity it guessed (>= num_rendered; the first num_rendered elements of
   every sub-array are the valid ones).  Pass the result to stp_binning_layout.  From the library's cache, else from the buffer's own header
   (see stp_blend_log_depth); negative (STP_ERR_INVALID_ARGUMENT) for a buffer that carries none or was carved for fewer than R entries.
   stp_backward does this lookup itself -- and refuses such a buffer instead of carving it on a guess: callers keep passing (buffer,
   num_rendered) as in the reference. */
int stp_binning_layout_count(const void* binning_buffer, int R);
/* Forgets the per-device size guesses (tile-list entries of the previous frames of each kind) that the run-ahead forward and the early
   binning request are sized by: the next forward of every kind takes the reference's path again (hand-over in the middle of the frame,
   exact request).  For tests and benchmarks that want a defined starting state; never needed for correctness. */
void stp_reset_size_guesses(void);
/* Run-ahead forward.  mode 0 = never, 1 = always, 2 = auto (the default; STP_RUN_AHEAD=0 / 1 in the environment sets 0 / 1 at load time).
   With it, a forward that has a size guess (i.e. every forward but the first of its kind) enqueues ALL its kernels on the guessed capacity
   before it reads num_rendered back -- no host wait in the middle of the frame; a frame that does not fit its guess is redone with the
   exact size before stp_forward returns.  Results are identical either way (keys, lists, image, gradients).  Measured on MI355X it
   costs 0.5 % at 1M Gaussians / 1080p (the padded entries pass through the device-wide sort) and gains 7 % at 1k Gaussians / 256x256
   (the round trip is a tenth of that frame): auto = run ahead when the guess is below 2^18 tile-list entries. */
void stp_set_run_ahead(int mode);
int stp_get_run_ahead(void);
int stp_image_layout(int width, int height, const char* name, size_t* offset, size_t* count);
/* ... of the image buffer of a forward restricted to the tile rows [tile_y0, tile_y1) (StpSettings::tile_y0 / tile_y1): it holds the
   window's pixel rows and tiles only, element 0 of every sub-array being the window's first pixel / tile (0, 0 = the whole frame).
   The count reported for "blend_log" is that of the default depth (see stp_blend_log_bytes). */
int stp_image_layout_rows(int width, int height, int tile_y0, int tile_y1, const char* name, size_t* offset, size_t* count);
/* ... with the blend log at the depth the buffer was really carved with (stp_blend_log_depth(buffer); 0 = the buffer holds no log, the
   name "blend_log" is then unknown): the count reported for "blend_log" matches the buffer. */
int stp_image_layout_depth(int width, int height, int tile_y0, int tile_y1, int log_depth, const char* name, size_t* offset, size_t* count);

/* Stage timer, the counterpart of the reference's `Timer` (rasterizer_impl.h:77-147; stages
   "Preprocess","Duplicate","Sort","Render", rasterizer_impl.cu:248) plus "BwdRender","BwdPreprocess".
   While enabled, every forward/backward records hipEvents around its stages on the call's stream (no extra
   host synchronisation); stp_timing_read waits for the recorded events and returns the MEAN milliseconds per
   stage over the calls since stp_timing_enable(1) (6 floats, unmeasured stages are -1).  One timer PER DEVICE behind a
   mutex: stp_timing_read / stp_timing_text report the calling thread's CURRENT device (hipSetDevice first); a backward is
   attributed to the latest forward of its device, so time one caller per device.  stp_timing_read returns STP_ERR_HIP
   when an event could not be created or recorded (the timings are then incomplete). */
void stp_timing_enable(int enabled);
int stp_timing_read(float* ms6);
/* Per-call stage times instead of their means: fills ms6[6 * k + stage] for the last n = min(calls since stp_timing_enable(1), 1024,
   capacity) forward(+backward) calls of the current device in chronological order (-1 = stage not measured in that call) and returns n.
   What makes a slow step attributable: bench.py puts the worst step's six stage times beside the median step's. */
int stp_timing_history(float* ms6, int capacity);
/* The same for the HOST: milliseconds the launching thread spent between recording a stage's two events (normally microseconds: launches are
   asynchronous).  A stage whose GPU interval is long while its kernels are not was waiting for a launch: the host figure tells a late host --
   a descheduled thread, a blocking driver call -- from a slow kernel. */
int stp_timing_history_host(float* ms6, int capacity);
/* The text the reference hands to the SIBR viewer (DebugVisualizationData::timings_text, rasterizer_impl.cu:391-399):
   "Timings: \n - Preprocess: <ms>ms\n - Duplicate: ...\n - Sort: ...\n - Render: ...\n - Total: <sum>ms\n" over the forward
   stages measured since stp_timing_enable(1); measured backward stages follow as two more lines.  Writes at most
   `size` bytes including the terminating NUL and returns the length the full text needs (as snprintf does). */
size_t stp_timing_text(char* buf, size_t size);

/* Measurement helper (bench.py `hbm_measured`; not part of the reference's interface): one launch of a plain float4 streaming kernel over
   `bytes` (a multiple of 16) on `stream` -- kind 0: read `src` (dst may be NULL), 1: write `dst`, 2: copy src -> dst, + 4: with non-temporal
   loads / stores -- with `blocks` workgroups of 256 threads, eight 16-byte accesses in flight per thread.  The caller times it with events on the same stream.  What the box's
   HBM delivers to the streaming stages of the path, next to the 8 TB/s spec peak. */
int stp_hbm_probe(int kind, void* dst, const void* src, size_t bytes, int blocks, void* stream);

const char* stp_last_error(void);
int stp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
