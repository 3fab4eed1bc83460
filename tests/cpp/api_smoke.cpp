// api_smoke.cpp -- drives libstp_raster.so through the C++ face (include/stp_rasterizer.hpp) the way a C++ user of the
// reference's `CudaRasterizer::Rasterizer` would (SIBR viewer, reference rasterizer.h:184-258): hipMalloc-backed
// std::function allocators, static forward / backward / markVisible.  Test infrastructure: reads one scene written by
// tests/test_cpp_api.py, writes every output back; the test compares them with the Python binding's results.
//
//   api_smoke <scene.bin> <out.bin>
//
// scene.bin: int32[18] {P, D, M, W, H, sort_mode, sort_order, tile_2x2, per_pixel, rect, tight, tbc, h44, lb, ewa,
//            render_depth, do_backward, record_log}, float[3] {tanfovx, tanfovy, scale_modifier}, then float arrays
//            bg[3] means3D[3P] shs[3MP] opacities[P] scales[3P] rotations[4P] view[16] proj[16] inv_viewproj[16] campos[3] dL_dpix[3WH]
// out.bin:   int32 num_rendered, float color[3WH], int32 radii[P], uint8 present[P],
//            (do_backward) float dL_dmean2D[3P] dL_dopacity[P] dL_dmean3D[3P] dL_dsh[3MP] dL_dscale[3P] dL_drot[4P],
//            int32 n, char timings_text[n]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "stp_rasterizer.hpp"

namespace {

void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) { std::fprintf(stderr, "api_smoke: %s: %s\n", what, hipGetErrorString(e)); std::exit(2); }
}

struct DeviceBuffer { // grows like the torch tensor behind the reference's resizeFunctional (rasterize_points.cu:33-41)
    char* ptr = nullptr;
    size_t cap = 0;
    char* resize(size_t n)
    {
        if (n > cap) {
            if (ptr) hip_check(hipFree(ptr), "hipFree");
            hip_check(hipMalloc(reinterpret_cast<void**>(&ptr), n), "hipMalloc");
            cap = n;
        }
        return ptr;
    }
    ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
};

template <class T> T* to_device(const std::vector<T>& h)
{
    T* d = nullptr;
    hip_check(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(h.size(), 1) * sizeof(T)), "hipMalloc");
    if (!h.empty()) hip_check(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D");
    return d;
}
template <class T> T* device_zeros(size_t n)
{
    T* d = nullptr;
    hip_check(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc");
    hip_check(hipMemset(d, 0, std::max<size_t>(n, 1) * sizeof(T)), "hipMemset");
    return d;
}
template <class T> void write_back(std::FILE* f, const T* d, size_t n)
{
    std::vector<T> h(n);
    if (n) hip_check(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost), "D2H");
    std::fwrite(h.data(), sizeof(T), n, f);
}
std::vector<float> read_floats(std::FILE* f, size_t n)
{
    std::vector<float> v(n);
    if (std::fread(v.data(), sizeof(float), n, f) != n) { std::fprintf(stderr, "api_smoke: short scene file\n"); std::exit(2); }
    return v;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: api_smoke <scene.bin> <out.bin>\n"); return 2; }
    std::FILE* in = std::fopen(argv[1], "rb");
    if (!in) { std::perror(argv[1]); return 2; }
    int hdr[18];
    if (std::fread(hdr, sizeof(int), 18, in) != 18) return 2;
    const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4];
    const bool render_depth = hdr[15] != 0, do_backward = hdr[16] != 0, record_log = hdr[17] != 0;
    const std::vector<float> fl = read_floats(in, 3);
    const float tanfovx = fl[0], tanfovy = fl[1], scale_modifier = fl[2];
    const size_t N = (size_t)W * H;
    float* bg = to_device(read_floats(in, 3));
    float* means3D = to_device(read_floats(in, 3 * (size_t)P));
    float* shs = to_device(read_floats(in, 3 * (size_t)M * P));
    float* opac = to_device(read_floats(in, P));
    float* scales = to_device(read_floats(in, 3 * (size_t)P));
    float* rots = to_device(read_floats(in, 4 * (size_t)P));
    float* view = to_device(read_floats(in, 16));
    float* proj = to_device(read_floats(in, 16));
    float* invvp = to_device(read_floats(in, 16));
    float* campos = to_device(read_floats(in, 3));
    float* dL_dpix = to_device(read_floats(in, 3 * N));
    std::fclose(in);

    using namespace StpRasterizer;
    SplattingSettings st;
    st.sort_settings.sort_mode = (SortMode)hdr[5];
    st.sort_settings.sort_order = (GlobalSortOrder)hdr[6];
    st.sort_settings.queue_sizes.tile_2x2 = hdr[7];
    st.sort_settings.queue_sizes.per_pixel = hdr[8];
    st.culling_settings.rect_bounding = hdr[9];
    st.culling_settings.tight_opacity_bounding = hdr[10];
    st.culling_settings.tile_based_culling = hdr[11];
    st.culling_settings.hierarchical_4x4_culling = hdr[12];
    st.load_balancing = hdr[13];
    st.proper_ewa_scaling = hdr[14];

    DeviceBuffer geom, binning, img;
    float* out_color = device_zeros<float>(3 * N);
    int* radii = device_zeros<int>(P);
    bool* present = device_zeros<bool>(P);
    DebugVisualizationData dbg;
    dbg.type = render_depth ? DebugVisualization::Depth : DebugVisualization::Disabled;
    dbg.timing_enabled = true;

    std::FILE* out = std::fopen(argv[2], "wb");
    if (!out) { std::perror(argv[2]); return 2; }
    try {
        Rasterizer::markVisible(P, means3D, view, proj, present);
        const int rendered = Rasterizer::forward(
            [&](size_t n) { return geom.resize(n); }, [&](size_t n) { return binning.resize(n); }, [&](size_t n) { return img.resize(n); },
            P, D, M, bg, W, H, st, dbg, means3D, shs, nullptr, opac, scales, scale_modifier, rots, nullptr, view, proj, invvp, campos,
            tanfovx, tanfovy, false, out_color, radii, false, nullptr, record_log);
        hip_check(hipDeviceSynchronize(), "forward");
        std::fwrite(&rendered, sizeof(int), 1, out);
        write_back(out, out_color, 3 * N);
        write_back(out, radii, P);
        write_back(out, reinterpret_cast<unsigned char*>(present), P);
        if (do_backward) {
            float* dmean2D = device_zeros<float>(3 * (size_t)P);
            float* records = device_zeros<float>((size_t)STP_GRAD_RECORD_FLOATS * P);
            float* dopac = device_zeros<float>(P);
            float* dcolor = device_zeros<float>(3 * (size_t)P);
            float* dmean3D = device_zeros<float>(3 * (size_t)P);
            float* dcov3D = device_zeros<float>(6 * (size_t)P);
            float* dsh = device_zeros<float>(3 * (size_t)M * P);
            float* dscale = device_zeros<float>(3 * (size_t)P);
            float* drot = device_zeros<float>(4 * (size_t)P);
            Rasterizer::backward(P, D, M, rendered, bg, W, H, st.sort_settings, st.culling_settings, st.proper_ewa_scaling, means3D, shs,
                                 opac, nullptr, scales, scale_modifier, rots, nullptr, view, proj, invvp, campos, tanfovx, tanfovy,
                                 out_color, radii, geom.ptr, binning.ptr, img.ptr, dL_dpix, dmean2D, records, dopac, dcolor, dmean3D,
                                 dcov3D, dsh, dscale, drot, false, nullptr, record_log);
            hip_check(hipDeviceSynchronize(), "backward");
            write_back(out, dmean2D, 3 * (size_t)P);
            write_back(out, dopac, P);
            write_back(out, dmean3D, 3 * (size_t)P);
            write_back(out, dsh, 3 * (size_t)M * P);
            write_back(out, dscale, 3 * (size_t)P);
            write_back(out, drot, 4 * (size_t)P);
        }
        const int n = (int)dbg.timings_text.size();
        std::fwrite(&n, sizeof(int), 1, out);
        std::fwrite(dbg.timings_text.data(), 1, n, out);
    } catch (const std::exception& e) {
        std::fclose(out);
        std::fprintf(stderr, "api_smoke: exception: %s\n", e.what());
        return 3;
    }
    std::fclose(out);
    return 0;
}
