// stp_torch_binding.cpp -- native host binding of libstp_raster.so for PyTorch-ROCm: the module
// `diff_gaussian_rasterization._stp_host` that `_C.py` delegates its three hot functions to.
//
// Replaces the reference's pybind layer (ext.cpp:15-19, rasterize_points.cu:33-253): same three functions, same argument
// and tuple orders (rasterize_points.h:26-82), tensors allocated through ATen, the current HIP stream taken from c10 --
// and NOTHING ELSE: no kernel lives here, every stage runs in libstp_raster.so behind its C ABI (include/stp_raster.h).
// It is a plain C++ translation unit built by g++ as a torch CppExtension (no .hip / .cu source, hence no hipify pass).
//
// Why it exists: the ctypes binding of rounds 1-2 spent 0.43 ms of host time per C1 step and 0.1 ms per C2 step on three
// Python allocator callbacks and ~35 argument conversions per call ("Host path").  Here the buffer-resize
// callbacks are C functions, and the scratch pool for the two large buffers (tile lists with their list-ordered entry
// records, image state with the blend log) lives in this file.
//
// The scratch pool.  Cycling a GB-sized block through torch's caching allocator every step invites splitting: smaller
// requests carve pieces off the free block, the next forward finds no hole of that size and the allocator falls back to
// hipMalloc (tens of ms per step -- observed on MI355X).  Buffers of >= 256 MiB are therefore kept on a small free list per
// device: handed out by the forward, handed back by the backward (release_scratch), capacities in steps of 64 MiB so that a
// buffer whose size follows num_rendered finds its predecessor.  Reuse across streams is ordered by an event recorded on
// the releasing stream; a generation counter per address lets the autograd function detect a backward that comes after
// its buffers were recycled (check_scratch).
#include <torch/extension.h>

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../../include/stp_raster.h"

namespace {

constexpr int64_t BIG_BYTES = 256ll << 20, BIG_STEP = 64ll << 20;

struct Pooled { torch::Tensor t; hipEvent_t ev; }; // ev: recorded on the releasing stream (owned by g_events, never destroyed: creating an
                                                   // event can cost a driver call when the runtime's signal pool grows -- not per step)
std::mutex g_mutex;
std::map<int, std::vector<Pooled>> g_free;          // device -> free buffers (oldest first)
std::unordered_map<uintptr_t, int64_t> g_generation; // data_ptr -> how often the buffer at this address was handed out
std::unordered_map<uintptr_t, hipEvent_t> g_events;  // data_ptr -> the event that orders reuse of the buffer at this address
struct KeptRecords { torch::Tensor t; hipStream_t stream; };
std::map<int, KeptRecords> g_records;               // device -> the gradient-record buffer of the last whole backward (all zeros at rest)
int g_keep = 4;                                       // free buffers kept per device
int64_t g_max_bytes = -1;                             // optional cap on the pooled bytes per device

// The library is bound at RUN time (load_library, called by _C.py with the path it resolved: the in-tree libstp_raster.so,
// or STP_RASTER_LIB / _C.use_library for A/B runs and the test-only IEEE-depth build), not at link time.
struct Api {
    void* handle = nullptr;
    decltype(&stp_forward) forward = nullptr;
    decltype(&stp_backward_phases) backward_phases = nullptr;
    decltype(&stp_mark_visible) mark_visible = nullptr;
    decltype(&stp_last_error) last_error = nullptr;
    decltype(&stp_abi_version) abi_version = nullptr;
    decltype(&stp_forget_buffer) forget_buffer = nullptr;
    decltype(&stp_set_forward_split) set_forward_split = nullptr;
} g_api;

int load_library(const std::string& path)
{
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) throw std::runtime_error(std::string("cannot load ") + path + ": " + dlerror());
    Api a;
    a.handle = h;
    a.forward = reinterpret_cast<decltype(a.forward)>(dlsym(h, "stp_forward"));
    a.backward_phases = reinterpret_cast<decltype(a.backward_phases)>(dlsym(h, "stp_backward_phases"));
    a.mark_visible = reinterpret_cast<decltype(a.mark_visible)>(dlsym(h, "stp_mark_visible"));
    a.last_error = reinterpret_cast<decltype(a.last_error)>(dlsym(h, "stp_last_error"));
    a.abi_version = reinterpret_cast<decltype(a.abi_version)>(dlsym(h, "stp_abi_version"));
    a.forget_buffer = reinterpret_cast<decltype(a.forget_buffer)>(dlsym(h, "stp_forget_buffer"));
    a.set_forward_split = reinterpret_cast<decltype(a.set_forward_split)>(dlsym(h, "stp_set_forward_split"));
    if (!a.set_forward_split || !a.forget_buffer || !a.forward || !a.backward_phases || !a.mark_visible || !a.last_error || !a.abi_version)
        throw std::runtime_error(path + " does not export the C ABI of include/stp_raster.h");
    if (a.abi_version() != STP_ABI_VERSION) throw std::runtime_error(path + ": ABI version mismatch");
    g_api = a; // (a previously loaded library stays mapped: buffers of its forwards may still be in flight)
    return a.abi_version();
}

void need_library()
{
    if (!g_api.forward) throw std::runtime_error("libstp_raster.so is not loaded (diff_gaussian_rasterization._C._load() does it); there is no CPU fallback");
}

[[noreturn]] void raise_last(int rc)
{
    const char* msg = g_api.last_error();
    throw std::runtime_error((msg && *msg) ? std::string(msg) : "libstp_raster error " + std::to_string(rc));
}

// A pooled buffer is used on whichever stream is current when it is handed out, not only on the stream torch::empty allocated it on: the caching
// allocator must know, or it could re-issue the block -- once the pool evicts it -- while kernels of another stream still read it.  (A no-op for
// the allocation stream itself, i.e. for single-stream training.)
void note_stream_use(const torch::Tensor& buf)
{
    c10::hip::HIPCachingAllocator::recordStream(buf.storage().data_ptr(), c10::hip::getCurrentHIPStream(buf.get_device()));
}

void put_back(const torch::Tensor& buf) // caller holds g_mutex
{
    auto& fl = g_free[buf.get_device()];
    for (auto& p : fl)
        if (p.t.data_ptr() == buf.data_ptr()) return;
    note_stream_use(buf);
    g_api.forget_buffer(buf.data_ptr()); // (its next tenant is described by what a forward carves there, or by the header a copy brings along)
    hipEvent_t& ev = g_events[(uintptr_t)buf.data_ptr()];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    bool recorded = ev && hipEventRecord(ev, c10::hip::getCurrentHIPStream(buf.get_device()).stream()) == hipSuccess;
    if (!recorded) (void)hipStreamSynchronize(c10::hip::getCurrentHIPStream(buf.get_device()).stream()); // (no event: order reuse the slow way)
    fl.push_back({buf, recorded ? ev : nullptr});
    auto total = [&]() { int64_t s = 0; for (auto& p : fl) s += p.t.numel(); return s; };
    while ((int)fl.size() > g_keep || (g_max_bytes >= 0 && fl.size() > 1 && total() > g_max_bytes)) fl.erase(fl.begin());
}

// The reference's resizeFunctional (rasterize_points.cu:33-41): grows a byte tensor on request.  The library may call the
// binning one twice per forward (a size guess before the num_rendered hand-over, the exact size afterwards if the guess
// was short: include/stp_raster.h): a request the current buffer already covers returns the same pointer.
// Which allocation a scratch buffer IS: the library caches (address, num_rendered) -> layout for the buffers its forwards carved, and
// an address alone does not say whose memory it is now -- a buffer is freed, a clone of ANOTHER forward's buffer lands on its address with the same
// num_rendered (static views repeat it exactly), and the stale entry would describe it.  The binding knows more than the address: it remembers the
// StorageImpl of every binning / image buffer its forwards returned (weak reference); a backward that is handed a tensor with another storage at a
// known address, or one whose storage has died since, tells the library to forget the address first, and the library reads the header the
// buffer carries (include/stp_raster.h: stp_forget_buffer).
std::unordered_map<uintptr_t, c10::weak_intrusive_ptr<c10::StorageImpl>> g_storage_of; // behind g_mutex
void remember_storage(const torch::Tensor& t)
{
    if (!t.defined() || t.numel() == 0) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_storage_of.size() > 4096) // forwards whose buffers nobody came back for
        for (auto it = g_storage_of.begin(); it != g_storage_of.end();) it = it->second.expired() ? g_storage_of.erase(it) : std::next(it);
    g_storage_of.insert_or_assign((uintptr_t)t.data_ptr(), t.storage().getWeakStorageImpl());
}
void forget_unless_same_storage(const torch::Tensor& t)
{
    if (!t.defined() || t.numel() == 0) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    const auto it = g_storage_of.find((uintptr_t)t.data_ptr());
    if (it != g_storage_of.end() && !it->second.expired() && it->second._unsafe_get_target() == t.storage().unsafeGetStorageImpl()) return;
    if (it != g_storage_of.end()) g_storage_of.erase(it);
    g_api.forget_buffer(t.data_ptr());
}

// Placement experiment (profiles/EXPERIMENTS.md, round 6: the k-buffer ring kernel's speed follows WHERE the frame's scratch buffers lie):
// STP_SCRATCH_OFFSET_GEOM / _BINNING / _IMAGE = bytes (multiple of 256, below 64 MiB) the buffer of that kind starts behind the start of its
// allocation.  Read at every request; unset or 0 = the allocation's own start.
constexpr int64_t OFFSET_SLACK = 64ll << 20;
int64_t scratch_offset(int kind)
{
    static const char* const names[3] = {"STP_SCRATCH_OFFSET_GEOM", "STP_SCRATCH_OFFSET_BINNING", "STP_SCRATCH_OFFSET_IMAGE"};
    const char* e = kind >= 0 && kind < 3 ? std::getenv(names[kind]) : nullptr;
    if (!e) return 0;
    const int64_t v = std::strtoll(e, nullptr, 0) & ~255ll;
    return v < 0 ? 0 : (v >= OFFSET_SLACK ? OFFSET_SLACK - 256 : v);
}
torch::Tensor empty_at_offset(int64_t n, const torch::TensorOptions& opt, int64_t off)
{
    return off ? torch::empty({n + OFFSET_SLACK}, opt).narrow(0, off, n) : torch::empty({n}, opt);
}

struct Resizer {
    torch::Tensor t;
    bool pooled = false, from_pool = false;
    int kind = -1;
    static void* call(void* user, size_t nbytes_)
    {
        auto* self = static_cast<Resizer*>(user);
        try {
            const int64_t nbytes = (int64_t)nbytes_;
            if (nbytes > 0 && nbytes <= self->t.numel()) return self->t.data_ptr();
            const int64_t off = scratch_offset(self->kind);
            if (self->pooled && nbytes >= BIG_BYTES) {
                std::lock_guard<std::mutex> lock(g_mutex);
                if (self->from_pool) put_back(self->t); // the guess was too small: the buffer goes back, a larger one comes
                const int64_t cap = (nbytes + BIG_STEP - 1) / BIG_STEP * BIG_STEP;
                auto& fl = g_free[self->t.get_device()];
                int hit = -1;
                for (int i = 0; i < (int)fl.size(); i++) {
                    const int64_t n = fl[i].t.numel();
                    if (n >= nbytes && n <= cap + cap / 4 && fl[i].t.storage_offset() == off && (hit < 0 || n < fl[hit].t.numel())) hit = i;
                }
                if (hit >= 0) {
                    Pooled p = fl[hit];
                    fl.erase(fl.begin() + hit);
                    if (p.ev) // the releasing stream's kernels may still be reading it: order this stream behind them
                        (void)hipStreamWaitEvent(c10::hip::getCurrentHIPStream(p.t.get_device()).stream(), p.ev, 0);
                    self->t = p.t;
                    note_stream_use(self->t);
                } else self->t = empty_at_offset(cap, self->t.options(), off);
                self->from_pool = true;
                g_generation[(uintptr_t)self->t.data_ptr()]++;
                return self->t.data_ptr();
            }
            if (off && nbytes) self->t = empty_at_offset(nbytes, self->t.options(), off);
            else self->t.resize_({nbytes});
            return nbytes ? self->t.data_ptr() : nullptr;
        } catch (...) { // surfaces as STP_ERR_ALLOC on the C side
            return nullptr;
        }
    }
};

const float* fptr(const torch::Tensor& t) // empty tensor -> NULL (reference convention: `torch.Tensor([])` marks an absent optional input)
{
    return t.defined() && t.numel() != 0 ? t.data_ptr<float>() : nullptr;
}

torch::Tensor prep(const torch::Tensor& t, const torch::Device& dev)
{
    if (!t.defined() || t.numel() == 0) return t;
    TORCH_CHECK(t.device() == dev, "expected all tensors on ", dev, ", got one on ", t.device());
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, "expected float32 tensor, got ", t.scalar_type());
    return t.contiguous();
}

// dict -> POD.  All keys are mandatory, as in the reference's json parser (rasterizer.h:160-182, every field read with
// .at()); a missing key raises KeyError where the reference raised json::out_of_range.
StpSettings settings_from_dict(const py::dict& d, bool record_log)
{
    StpSettings s{};
    const py::dict ss = d["sort_settings"].cast<py::dict>(), cs = d["culling_settings"].cast<py::dict>(), q = ss["queue_sizes"].cast<py::dict>();
    s.sort_mode = ss["sort_mode"].cast<int>();
    s.sort_order = ss["sort_order"].cast<int>();
    s.queue_tile_4x4 = q["tile_4x4"].cast<int>();
    s.queue_tile_2x2 = q["tile_2x2"].cast<int>();
    s.queue_per_pixel = q["per_pixel"].cast<int>();
    s.rect_bounding = cs["rect_bounding"].cast<bool>();
    s.tight_opacity_bounding = cs["tight_opacity_bounding"].cast<bool>();
    s.tile_based_culling = cs["tile_based_culling"].cast<bool>();
    s.hierarchical_4x4_culling = cs["hierarchical_4x4_culling"].cast<bool>();
    s.load_balancing = d["load_balancing"].cast<bool>();
    s.proper_ewa_scaling = d["proper_ewa_scaling"].cast<bool>();
    if (d.contains("_tile_rows") && !d["_tile_rows"].is_none()) {
        const py::tuple tr = py::tuple(d["_tile_rows"]);
        s.tile_y0 = tr[0].cast<int>();
        s.tile_y1 = tr[1].cast<int>();
    }
    s.record_blend_log = record_log ? 1 : 0;
    return s;
}

// == RasterizeGaussiansCUDA (reference rasterize_points.cu:43-138).
// Returns (num_rendered, out_color (3,H,W), radii (P,) int32, geomBuffer, binningBuffer, imgBuffer).
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                    const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const torch::Tensor& inv_viewprojmatrix, const float tan_fovx,
                    const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                    const torch::Tensor& campos, const bool prefiltered, const py::dict& settings, const bool render_depth, const bool debug,
                    const bool record_log)
{
    need_library();
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build) needs tensors on a GPU device; there is no CPU path in the product");
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    const bool windowed = settings.contains("_tile_rows") && !settings["_tile_rows"].is_none();
    const auto fopt = means3D.options().dtype(torch::kFloat32), iopt = means3D.options().dtype(torch::kInt32), bopt = means3D.options().dtype(torch::kByte);
    // the render kernels write every pixel of the tile rows they cover and preprocess writes every Gaussian's radius:
    // zero-filled outputs are only needed when nothing runs (P == 0) or when a tile-row window leaves rows untouched
    const bool zero = P == 0 || windowed;
    torch::Tensor out_color = zero ? torch::zeros({3, H, W}, fopt) : torch::empty({3, H, W}, fopt);
    torch::Tensor radii = zero ? torch::zeros({P}, iopt) : torch::empty({P}, iopt);
    Resizer geom{torch::empty({0}, bopt), false, false, 0}, binning{torch::empty({0}, bopt), true, false, 1}, img{torch::empty({0}, bopt), true, false, 2};
    int rendered = 0;
    if (P != 0) {
        const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
        StpSettings s = settings_from_dict(settings, record_log);
        if (render_depth) { // DebugVisualization::Depth (reference rasterize_points.cu:104-107); no log: it has no backward
            s.debug_visualization = STP_DEBUG_DEPTH;
            s.record_blend_log = 0;
        }
        const torch::Tensor bg_ = prep(background, dev), m3_ = prep(means3D, dev), sh_ = prep(sh, dev), col_ = prep(colors, dev), op_ = prep(opacity, dev),
                            sc_ = prep(scales, dev), ro_ = prep(rotations, dev), c3_ = prep(cov3D_precomp, dev), vm_ = prep(viewmatrix, dev),
                            pm_ = prep(projmatrix, dev), inv_ = prep(inv_viewprojmatrix, dev), cam_ = prep(campos, dev);
        const c10::hip::HIPGuard guard(dev.index());
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        int rc;
        {
            py::gil_scoped_release nogil; // (the call blocks once, on the num_rendered hand-over)
            rc = g_api.forward(&Resizer::call, &geom, &Resizer::call, &binning, &Resizer::call, &img, P, degree, M, fptr(bg_), W, H, &s, fptr(m3_),
                             fptr(sh_), fptr(col_), fptr(op_), fptr(sc_), scale_modifier, fptr(ro_), fptr(c3_), fptr(vm_), fptr(pm_), fptr(inv_),
                             fptr(cam_), tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(), radii.data_ptr<int>(), debug ? 1 : 0,
                             (void*)stream);
        }
        if (rc < 0) raise_last(rc);
        rendered = rc;
        remember_storage(binning.t);
        remember_storage(img.t);
    }
    return std::make_tuple(rendered, out_color, radii, geom.t, binning.t, img.t);
}

// == RasterizeGaussiansBackwardCUDA (reference rasterize_points.cu:140-232).
// phases / partial: extension for tile-row sharding (include/stp_raster.h, stp_backward_phases): phases = 1 runs only the
// render half and returns the (P,16) gradient records; phases = 2 takes the records (after the caller's all-reduce) and
// runs the per-Gaussian half.
std::vector<torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& opacities,
                             const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                             const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                             const torch::Tensor& inv_viewprojmatrix, const float tan_fovx, const float tan_fovy, const torch::Tensor& pixel_colors,
                             const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                             const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                             const py::dict& settings, const bool debug, const bool record_log, const int phases, const c10::optional<torch::Tensor>& partial,
                             const c10::optional<std::vector<torch::Tensor>>& outputs)
{
    need_library();
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build) needs tensors on a GPU device; there is no CPU path in the product");
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const int M = sh.numel() != 0 ? (int)sh.size(1) : 0;
    const auto fopt = means3D.options().dtype(torch::kFloat32);
    const int rec_floats = (phases & 4) ? STP_GRAD_RECORD_USED : STP_GRAD_RECORD_FLOATS; // (bit 2: compact records, the tile-row shard's wire format)
    // STP_KEEP_RECORDS=1 (MEASURED, round 4, off): a whole backward keeps its record buffer between steps and the per-Gaussian half clears what
    // it reads (phases bit 3), so that the 64 B per Gaussian are zero-filled once, not every step.  One buffer per device, reused only by the
    // stream that used it last and only at the same size.  The clearing stores cost the per-Gaussian kernel more than torch's fill saves:
    // C2-full step 2.376 -> 2.384 ms (BwdPreprocess 0.116 -> 0.126), C5 6.25 -> 6.39 ms (0.72-0.80 -> 0.90), C3 unchanged
    // (profiles/r04_keep_records_ab.txt) -- a 7 TB/s fill kernel is hard to beat with scattered 48-byte stores.
    static const bool keep_enabled = [] { const char* e = std::getenv("STP_KEEP_RECORDS"); return e && e[0] == '1'; }();
    const bool keep_records = keep_enabled && phases == 3 && !partial.has_value() && P != 0;
    torch::Tensor records;
    hipStream_t rec_stream = nullptr;
    if (keep_records) {
        rec_stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_records.find(dev.index());
        if (it != g_records.end() && it->second.stream == rec_stream && it->second.t.size(0) == P && it->second.t.size(1) == rec_floats) records = it->second.t;
        if (it != g_records.end()) g_records.erase(it); // (in use, or superseded)
    }
    if (!records.defined()) records = partial.has_value() ? *partial : torch::zeros({P, rec_floats}, fopt);
    TORCH_CHECK(records.dim() == 2 && records.size(0) == P && records.size(1) == rec_floats && records.scalar_type() == torch::kFloat32 &&
                    records.is_contiguous(), "partial must be a contiguous float32 (P,", rec_floats, ") tensor");
    torch::Tensor dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations;
    if ((phases & 2) && outputs.has_value()) {
        // a chunked per-Gaussian half (phases bits 8-23): the later chunks write into the tensors the first one allocated
        TORCH_CHECK(outputs->size() == 8, "outputs: the eight gradient tensors of an earlier chunk");
        dL_dmeans2D = (*outputs)[0]; dL_dcolors = (*outputs)[1]; dL_dopacity = (*outputs)[2]; dL_dmeans3D = (*outputs)[3];
        dL_dcov3D = (*outputs)[4]; dL_dsh = (*outputs)[5]; dL_dscales = (*outputs)[6]; dL_drotations = (*outputs)[7];
    } else if (phases & 2) {
        // the per-Gaussian half writes every row of its outputs (zeros for invisible Gaussians): no zero-fill needed,
        // except for the scale/rotation gradients when a precomputed covariance is used (then they are not touched)
        const bool have_scales = scales.numel() != 0;
        dL_dmeans2D = torch::empty({P, 3}, fopt); dL_dcolors = torch::empty({P, 3}, fopt); dL_dopacity = torch::empty({P, 1}, fopt);
        dL_dmeans3D = torch::empty({P, 3}, fopt); dL_dcov3D = torch::empty({P, 6}, fopt); dL_dsh = torch::empty({P, M, 3}, fopt);
        dL_dscales = have_scales ? torch::empty({P, 3}, fopt) : torch::zeros({P, 3}, fopt);
        dL_drotations = have_scales ? torch::empty({P, 4}, fopt) : torch::zeros({P, 4}, fopt);
    }
    const StpSettings s = settings_from_dict(settings, record_log);
    if (P != 0) {
        const torch::Tensor bg_ = prep(background, dev), m3_ = prep(means3D, dev), sh_ = prep(sh, dev), col_ = prep(colors, dev), op_ = prep(opacities, dev),
                            sc_ = prep(scales, dev), ro_ = prep(rotations, dev), c3_ = prep(cov3D_precomp, dev), vm_ = prep(viewmatrix, dev),
                            pm_ = prep(projmatrix, dev), inv_ = prep(inv_viewprojmatrix, dev), cam_ = prep(campos, dev), pix_ = prep(pixel_colors, dev),
                            dl_ = prep(dL_dout_color, dev);
        const torch::Tensor radii_ = radii.contiguous();
        auto optf = [](const torch::Tensor& t) -> float* { return t.defined() && t.numel() != 0 ? t.data_ptr<float>() : nullptr; };
        auto optb = [](const torch::Tensor& t) -> char* { return t.defined() && t.numel() != 0 ? reinterpret_cast<char*>(t.data_ptr()) : nullptr; };
        const c10::hip::HIPGuard guard(dev.index());
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        forget_unless_same_storage(binningBuffer); // (a clone, a copy or a recycled address: the library then reads the buffer's own header)
        forget_unless_same_storage(imageBuffer);
        const int rc = g_api.backward_phases(keep_records ? (phases | 8) : phases, P, degree, M, R, fptr(bg_), W, H, &s, fptr(m3_), fptr(sh_), fptr(op_), fptr(col_), fptr(sc_),
                                           scale_modifier, fptr(ro_), fptr(c3_), fptr(vm_), fptr(pm_), fptr(inv_), fptr(cam_), tan_fovx, tan_fovy,
                                           fptr(pix_), radii_.numel() ? radii_.data_ptr<int>() : nullptr, optb(geomBuffer), optb(binningBuffer),
                                           optb(imageBuffer), fptr(dl_), optf(dL_dmeans2D), records.data_ptr<float>(), optf(dL_dopacity), optf(dL_dcolors),
                                           optf(dL_dmeans3D), optf(dL_dcov3D), optf(dL_dsh), optf(dL_dscales), optf(dL_drotations), debug ? 1 : 0,
                                           (void*)stream);
        if (rc < 0) raise_last(rc);
        if (keep_records) { // all zeros again once the kernels just enqueued have run
            std::lock_guard<std::mutex> lock(g_mutex);
            g_records[dev.index()] = KeptRecords{records, rec_stream};
        }
    }
    if ((phases & 3) == 1) return {records};
    return {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations};
}

// == markVisible (reference rasterize_points.cu:234-253)
torch::Tensor mark_visible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix)
{
    need_library();
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build) needs tensors on a GPU device; there is no CPU path in the product");
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor m3_ = prep(means3D, dev), vm_ = prep(viewmatrix, dev), pm_ = prep(projmatrix, dev);
        const c10::hip::HIPGuard guard(dev.index());
        const int rc = g_api.mark_visible(P, fptr(m3_), fptr(vm_), fptr(pm_), reinterpret_cast<uint8_t*>(present.data_ptr()),
                                        (void*)c10::hip::getCurrentHIPStream(dev.index()).stream());
        if (rc < 0) raise_last(rc);
    }
    return present;
}

// ---- scratch pool surface -------------------------------------------------------------------------------------------
int64_t scratch_generation(const torch::Tensor& buf) // token the autograd function keeps with a pooled buffer (0 for ordinary ones)
{
    if (buf.numel() < BIG_BYTES) return 0;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_generation.find((uintptr_t)buf.data_ptr());
    return it == g_generation.end() ? 0 : it->second;
}

void check_scratch(const torch::Tensor& buf, int64_t generation)
{
    if (generation && scratch_generation(buf) != generation)
        throw std::runtime_error("a scratch buffer of this forward (tile lists / blend log) was recycled by a later forward pass; run the forward "
                                 "again before this backward");
}

void release_scratch(const torch::Tensor& buf) // hand a pooled buffer back after the backward that consumed it
{
    if (buf.numel() < BIG_BYTES || !buf.is_cuda()) return;
    const c10::hip::HIPGuard guard(buf.get_device());
    std::lock_guard<std::mutex> lock(g_mutex);
    put_back(buf);
}

int64_t clear_scratch_pool(int device) // device < 0: all devices; returns the bytes released
{
    std::lock_guard<std::mutex> lock(g_mutex);
    int64_t freed = 0;
    for (auto it = g_free.begin(); it != g_free.end();) {
        if (device >= 0 && it->first != device) { ++it; continue; }
        for (auto& p : it->second) freed += p.t.numel();
        it = g_free.erase(it);
    }
    for (auto it = g_records.begin(); it != g_records.end();) { // the kept gradient-record buffer goes too
        if (device >= 0 && it->first != device) { ++it; continue; }
        freed += it->second.t.numel() * (int64_t)sizeof(float);
        it = g_records.erase(it);
    }
    return freed;
}

void set_scratch_pool_limit(int max_buffers, int64_t max_bytes)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    g_keep = max_buffers;
    g_max_bytes = max_bytes;
}

std::vector<int64_t> pooled_sizes(int device)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    std::vector<int64_t> out;
    auto it = g_free.find(device);
    if (it != g_free.end()) for (auto& p : it->second) out.push_back(p.t.numel());
    return out;
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "native host binding of libstp_raster.so (no kernels here: every stage runs behind the C ABI of include/stp_raster.h)";
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("mark_visible", &mark_visible);
    m.def("scratch_generation", &scratch_generation);
    m.def("check_scratch", &check_scratch);
    m.def("release_scratch", &release_scratch);
    m.def("forget_unless_same_storage", &forget_unless_same_storage);
    m.def("set_forward_split", [](int tile_row, uintptr_t event) { need_library(); g_api.set_forward_split(tile_row, reinterpret_cast<void*>(event)); },
          "the next rasterize_gaussians of this thread renders tile rows below / from tile_row in two launches with `event` (a hipEvent_t handle) recorded between them");
    m.def("clear_scratch_pool", &clear_scratch_pool);
    m.def("set_scratch_pool_limit", &set_scratch_pool_limit);
    m.def("pooled_sizes", &pooled_sizes);
    m.def("load_library", &load_library);
    m.attr("BIG_BYTES") = BIG_BYTES;
}
