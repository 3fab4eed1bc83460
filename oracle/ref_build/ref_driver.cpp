/*
 * oracle/ref_build/ref_driver.cpp -- C ABI over THE REFERENCE'S OWN rasterizer, compiled for gfx950.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/stp_oracle.h).  This file holds no rasterizer logic: it moves
 * host arrays to the GPU, calls CudaRasterizer::Rasterizer::{forward, backward, markVisible}
 * (ref: cuda_rasterizer/rasterizer.h:184-258, rasterizer_impl.cu:161-526) exactly as the reference's
 * torch binding does (ref: rasterize_points.cu:43-252), and copies results plus the reference's
 * intermediate state back.  The state is located with the reference's own carving functions
 * (GeometryState / ImageState / BinningState ::fromChunk, ref: rasterizer_impl.cu:175-217).
 *
 * Built by oracle/ref_build/build_ref.sh into oracle/_ref/ (git-ignored), together with the
 * reference's three translation units as hipify-perl translates them.  The exported functions mirror
 * the CPU oracle's (oracle/stp_oracle.h: orc_* -> ref_*) so that oracle/reference.py can drive either.
 */
#include "rasterizer_impl.h"   // the hipify-perl translation in the build's temporary directory
#include "config.h"            // BLOCK_X / BLOCK_Y

#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

std::string g_error;

struct DevBuf {
    char* p = nullptr;
    size_t cap = 0;
    char* grow(size_t n) {
        if (n > cap) {
            if (p) (void)hipFree(p);
            size_t want = n + n / 8 + 256;
            if (hipMalloc(&p, want) != hipSuccess) throw std::runtime_error("hipMalloc failed in the reference driver");
            cap = want;
        }
        return p;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <class T> T* to_dev(const T* host, size_t n, std::vector<void*>& owned) {
    if (!host || n == 0) return nullptr;
    void* d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed in the reference driver");
    owned.push_back(d);
    if (hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy H2D failed");
    return static_cast<T*>(d);
}
template <class T> T* dev_zeros(size_t n, std::vector<void*>& owned) {
    void* d = nullptr;
    if (hipMalloc(&d, (n ? n : 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed in the reference driver");
    owned.push_back(d);
    (void)hipMemset(d, 0, (n ? n : 1) * sizeof(T));
    return static_cast<T*>(d);
}
void check_device(const char* where) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string(where) + ": " + hipGetErrorString(e));
}

} // namespace

/* Same layout as OrcSettings (oracle/stp_oracle.h). */
struct RefSettings {
    int32_t sort_mode, sort_order, queue_tile_4x4, queue_tile_2x2, queue_per_pixel;
    int32_t rect_bounding, tight_opacity_bounding, tile_based_culling, hierarchical_4x4_culling;
    int32_t load_balancing, proper_ewa_scaling, tile_y0, tile_y1, debug_visualization;
};

struct RefFrame {
    int P = 0, D = 0, M = 0, W = 0, H = 0, R = 0;
    CudaRasterizer::SplattingSettings settings{};
    DevBuf geom, binning, img;
    std::vector<void*> owned;
    /* device inputs */
    float *bg = nullptr, *means3D = nullptr, *shs = nullptr, *colors = nullptr, *opac = nullptr, *scales = nullptr,
          *rots = nullptr, *cov3D = nullptr, *view = nullptr, *proj = nullptr, *inv = nullptr, *cam = nullptr;
    float mod = 1.f, tfx = 0.f, tfy = 0.f;
    float* out_color = nullptr;
    int* radii = nullptr;
    std::map<std::string, std::vector<char>> cache;
    /* gradient outputs + dL_dpix of the timing loop */
    float *dpix = nullptr, *g_mean2D = nullptr, *g_conic = nullptr, *g_opac = nullptr, *g_color = nullptr, *g_mean3D = nullptr,
          *g_cov3D = nullptr, *g_sh = nullptr, *g_scale = nullptr, *g_rot = nullptr;
    ~RefFrame() { for (void* p : owned) (void)hipFree(p); }
};

static CudaRasterizer::SplattingSettings to_ref(const RefSettings& s) {
    CudaRasterizer::SplattingSettings r{};
    r.sort_settings.sort_mode = (CudaRasterizer::SortMode)s.sort_mode;
    r.sort_settings.sort_order = (CudaRasterizer::GlobalSortOrder)s.sort_order;
    r.sort_settings.queue_sizes.tile_4x4 = s.queue_tile_4x4;
    r.sort_settings.queue_sizes.tile_2x2 = s.queue_tile_2x2;
    r.sort_settings.queue_sizes.per_pixel = s.queue_per_pixel;
    r.culling_settings.rect_bounding = s.rect_bounding != 0;
    r.culling_settings.tight_opacity_bounding = s.tight_opacity_bounding != 0;
    r.culling_settings.tile_based_culling = s.tile_based_culling != 0;
    r.culling_settings.hierarchical_4x4_culling = s.hierarchical_4x4_culling != 0;
    r.load_balancing = s.load_balancing != 0;
    r.proper_ewa_scaling = s.proper_ewa_scaling != 0;
    return r;
}

static int run_forward(RefFrame* f, int prefiltered, bool depth_viz) {
    DebugVisualizationData dbg;
    if (depth_viz) dbg.type = DebugVisualization::Depth;   /* ref: rasterize_points.cu:104-107 */
    auto gf = [f](size_t n) { return f->geom.grow(n); };
    auto bf = [f](size_t n) { return f->binning.grow(n); };
    auto imf = [f](size_t n) { return f->img.grow(n); };
    return CudaRasterizer::Rasterizer::forward(gf, bf, imf, f->P, f->D, f->M, f->bg, f->W, f->H, f->settings, dbg,
                                               f->means3D, f->shs, f->colors, f->opac, f->scales, f->mod, f->rots,
                                               f->cov3D, f->view, f->proj, f->inv, f->cam, f->tfx, f->tfy,
                                               prefiltered != 0, f->out_color, f->radii, false);
}

static void run_backward(RefFrame* f, const float* pixel_colors, const float* dpix) {
    CudaRasterizer::Rasterizer::backward(f->P, f->D, f->M, f->R, f->bg, f->W, f->H, f->settings.sort_settings,
                                         f->settings.culling_settings, f->settings.proper_ewa_scaling, f->means3D, f->shs,
                                         f->opac, f->colors, f->scales, f->mod, f->rots, f->cov3D, f->view, f->proj, f->inv,
                                         f->cam, f->tfx, f->tfy, pixel_colors, f->radii, f->geom.p, f->binning.p, f->img.p,
                                         dpix, f->g_mean2D, f->g_conic, f->g_opac, f->g_color, f->g_mean3D, f->g_cov3D,
                                         f->g_sh, f->g_scale, f->g_rot, false);
}

static void alloc_grads(RefFrame* f) {
    if (f->g_mean2D) return;
    const size_t P = f->P, M = f->M;
    f->g_mean2D = dev_zeros<float>(P * 3, f->owned);
    f->g_conic = dev_zeros<float>(P * 4, f->owned);
    f->g_opac = dev_zeros<float>(P, f->owned);
    f->g_color = dev_zeros<float>(P * 3, f->owned);
    f->g_mean3D = dev_zeros<float>(P * 3, f->owned);
    f->g_cov3D = dev_zeros<float>(P * 6, f->owned);
    f->g_sh = dev_zeros<float>(P * M * 3, f->owned);
    f->g_scale = dev_zeros<float>(P * 3, f->owned);
    f->g_rot = dev_zeros<float>(P * 4, f->owned);
}
static void zero_grads(RefFrame* f) {   /* the binding's torch::zeros, ref: rasterize_points.cu:178-186 */
    const size_t P = f->P, M = f->M;
    (void)hipMemsetAsync(f->g_mean2D, 0, P * 3 * 4);
    (void)hipMemsetAsync(f->g_conic, 0, P * 4 * 4);
    (void)hipMemsetAsync(f->g_opac, 0, P * 4);
    (void)hipMemsetAsync(f->g_color, 0, P * 3 * 4);
    (void)hipMemsetAsync(f->g_mean3D, 0, P * 3 * 4);
    (void)hipMemsetAsync(f->g_cov3D, 0, P * 6 * 4);
    if (M) (void)hipMemsetAsync(f->g_sh, 0, P * M * 3 * 4);
    (void)hipMemsetAsync(f->g_scale, 0, P * 3 * 4);
    (void)hipMemsetAsync(f->g_rot, 0, P * 4 * 4);
}

extern "C" {

const char* ref_last_error(void) { return g_error.c_str(); }

/* "hipify-perl + hipcc <flags>" description of this build, for reports. */
const char* ref_build_info(void) {
#ifdef STP_REF_BUILD_INFO
    return STP_REF_BUILD_INFO;
#else
    return "unknown";
#endif
}

int ref_forward(int P, int D, int M, const float* background, int W, int H, const RefSettings* s,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii, RefFrame** frame_out) {
    RefFrame* f = nullptr;
    try {
        if (s->tile_y1 > 0) throw std::runtime_error("tile-row windows are our extension; the reference has none");
        f = new RefFrame;
        f->P = P; f->D = D; f->M = M; f->W = W; f->H = H;
        f->settings = to_ref(*s);
        f->mod = scale_modifier; f->tfx = tan_fovx; f->tfy = tan_fovy;
        f->bg = to_dev(background, 3, f->owned);
        f->means3D = to_dev(means3D, (size_t)P * 3, f->owned);
        f->shs = to_dev(shs, (size_t)P * M * 3, f->owned);
        f->colors = to_dev(colors_precomp, (size_t)P * 3, f->owned);
        f->opac = to_dev(opacities, (size_t)P, f->owned);
        f->scales = to_dev(scales, (size_t)P * 3, f->owned);
        f->rots = to_dev(rotations, (size_t)P * 4, f->owned);
        f->cov3D = to_dev(cov3D_precomp, (size_t)P * 6, f->owned);
        f->view = to_dev(viewmatrix, 16, f->owned);
        f->proj = to_dev(projmatrix, 16, f->owned);
        f->inv = to_dev(inv_viewprojmatrix, 16, f->owned);
        f->cam = to_dev(cam_pos, 3, f->owned);
        f->out_color = dev_zeros<float>((size_t)3 * W * H, f->owned);   /* torch::full(0), ref: rasterize_points.cu:80 */
        f->radii = dev_zeros<int>((size_t)P, f->owned);
        int R = 0;
        if (P != 0) R = run_forward(f, prefiltered, s->debug_visualization == 1);
        check_device("reference forward");
        f->R = R;
        (void)hipMemcpy(out_color, f->out_color, (size_t)3 * W * H * 4, hipMemcpyDeviceToHost);
        if (radii && P) (void)hipMemcpy(radii, f->radii, (size_t)P * 4, hipMemcpyDeviceToHost);
        *frame_out = f;
        return R;
    } catch (const std::exception& e) {
        g_error = e.what();
        delete f;
        *frame_out = nullptr;
        return -1;
    }
}

int ref_backward(RefFrame* f, const float* pixel_colors, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                 float* dL_drot) {
    try {
        const size_t P = f->P, M = f->M, N = (size_t)f->W * f->H;
        if (P == 0) return 0;
        std::vector<void*> tmp;
        struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } guard{tmp};
        float* pc = to_dev(pixel_colors, 3 * N, tmp);
        float* dp = to_dev(dL_dpix, 3 * N, tmp);
        alloc_grads(f);
        zero_grads(f);
        run_backward(f, pc, dp);
        check_device("reference backward");
        auto back = [](float* h, const float* d, size_t n) { if (h && n) (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost); };
        back(dL_dmean2D, f->g_mean2D, P * 3); back(dL_dconic, f->g_conic, P * 4); back(dL_dopacity, f->g_opac, P);
        back(dL_dcolor, f->g_color, P * 3); back(dL_dmean3D, f->g_mean3D, P * 3); back(dL_dcov3D, f->g_cov3D, P * 6);
        back(dL_dsh, f->g_sh, P * M * 3); back(dL_dscale, f->g_scale, P * 3); back(dL_drot, f->g_rot, P * 4);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

/* Times `steps` forward(+backward) passes of the reference on the frame's resident inputs (after `warmup`
   untimed ones), hipEvents on the null stream the reference launches on.  Writes ms per step for
   forward and backward; returns 0.  dL_dpix = host (3,H,W) or NULL for forward only. */
int ref_time_steps(RefFrame* f, const float* dL_dpix, int warmup, int steps, float* fwd_ms, float* bwd_ms) {
    try {
        const size_t N = (size_t)f->W * f->H;
        if (dL_dpix && !f->dpix) f->dpix = to_dev(dL_dpix, 3 * N, f->owned);
        if (dL_dpix) alloc_grads(f);
        hipEvent_t e0, e1, e2;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
        double tf = 0, tb = 0;
        for (int it = 0; it < warmup + steps; ++it) {
            (void)hipEventRecord(e0, 0);
            f->R = run_forward(f, 0, false);
            (void)hipEventRecord(e1, 0);
            if (dL_dpix) { zero_grads(f); run_backward(f, f->out_color, f->dpix); }
            (void)hipEventRecord(e2, 0);
            (void)hipEventSynchronize(e2);
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2);
            if (it >= warmup) { tf += a; tb += b; }
        }
        check_device("reference timing loop");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        *fwd_ms = (float)(tf / steps); *bwd_ms = (float)(tb / steps);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

void ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    std::vector<void*> tmp;
    float* m = to_dev(means3D, (size_t)P * 3, tmp);
    float* v = to_dev(viewmatrix, 16, tmp);
    float* p = to_dev(projmatrix, 16, tmp);
    bool* out = reinterpret_cast<bool*>(dev_zeros<uint8_t>((size_t)P, tmp));
    if (P) CudaRasterizer::Rasterizer::markVisible(P, m, v, p, out);
    (void)hipDeviceSynchronize();
    if (P) (void)hipMemcpy(present, out, (size_t)P, hipMemcpyDeviceToHost);
    for (void* q : tmp) (void)hipFree(q);
}

void ref_frame_free(RefFrame* f) { delete f; }
int ref_frame_num_rendered(const RefFrame* f) { return f->R; }

/* Same names and scalar types as orc_frame_array (oracle/stp_oracle.h). */
int64_t ref_frame_array(RefFrame* f, const char* name, const void** data) {
    using namespace CudaRasterizer;
    const std::string key(name);
    auto it = f->cache.find(key);
    if (it == f->cache.end()) {
        const size_t P = f->P, R = f->R, N = (size_t)f->W * f->H;
        const bool inv = f->settings.sort_settings.requiresDepthAlongRay();
        char* gp = f->geom.p; char* bp = f->binning.p; char* ip = f->img.p;
        if (!gp || !ip) return -1;
        GeometryState g = GeometryState::fromChunk(gp, P, inv);
        ImageState im = ImageState::fromChunk(ip, N);
        BinningState b{};
        if (bp && R > 0) b = BinningState::fromChunk(bp, R);
        const int tiles = ((f->W + BLOCK_X - 1) / BLOCK_X) * ((f->H + BLOCK_Y - 1) / BLOCK_Y);
        const void* src = nullptr; size_t bytes = 0;
        if (key == "depths") { src = g.depths; bytes = P * 4; }
        else if (key == "clamped") { src = g.clamped; bytes = P * 3; }
        else if (key == "radii") { src = f->radii; bytes = P * 4; }
        else if (key == "rects2D") { src = g.rects2D; bytes = P * 8; }
        else if (key == "means2D") { src = g.means2D; bytes = P * 8; }
        else if (key == "cov3D") { src = g.cov3D; bytes = P * 24; }
        else if (key == "cov3D_inv") { src = g.cov3D_inv; bytes = inv ? P * 48 : 0; }
        else if (key == "conic_opacity") { src = g.conic_opacity; bytes = P * 16; }
        else if (key == "rgb") { src = g.rgb; bytes = P * 12; }
        else if (key == "tiles_touched") { src = g.tiles_touched; bytes = P * 4; }
        else if (key == "point_offsets") { src = g.point_offsets; bytes = P * 4; }
        else if (key == "keys_unsorted") { src = b.point_list_keys_unsorted; bytes = R * 8; }
        else if (key == "values_unsorted") { src = b.point_list_unsorted; bytes = R * 4; }
        else if (key == "keys") { src = b.point_list_keys; bytes = R * 8; }
        else if (key == "point_list") { src = b.point_list; bytes = R * 4; }
        else if (key == "ranges") { src = im.ranges; bytes = (size_t)tiles * 8; }
        else if (key == "final_T") { src = im.accum_alpha; bytes = N * 4; }
        else if (key == "n_contrib") { src = im.n_contrib; bytes = N * 4; }
        else return -1;
        std::vector<char> host(bytes);
        if (bytes && src) (void)hipMemcpy(host.data(), src, bytes, hipMemcpyDeviceToHost);
        it = f->cache.emplace(key, std::move(host)).first;
    }
    static const std::map<std::string, int> width = {
        {"depths", 4}, {"clamped", 1}, {"radii", 4}, {"rects2D", 4}, {"means2D", 4}, {"cov3D", 4}, {"cov3D_inv", 4},
        {"conic_opacity", 4}, {"rgb", 4}, {"tiles_touched", 4}, {"point_offsets", 4}, {"keys_unsorted", 8},
        {"values_unsorted", 4}, {"keys", 8}, {"point_list", 4}, {"ranges", 4}, {"final_T", 4}, {"n_contrib", 4}};
    *data = it->second.data();
    return (int64_t)(it->second.size() / width.at(key));
}

} // extern "C"
