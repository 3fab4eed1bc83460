// stp_api.hip -- the C ABI of libstp_raster.so (declared in include/stp_raster.h) and the host
// orchestration of one frame.  Replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (reference cuda_rasterizer/rasterizer_impl.cu:221-413, 417-526, 161-173) and the state carving of
// rasterizer_impl.cu:175-217.
//
// Stage order of a forward is the reference's: preprocess -> inclusive scan -> (one host read-back of
// num_rendered) -> duplicate -> sort -> tile ranges -> render, where "sort" is by default a radix sort on the tile bits
// followed by the per-tile (depth, id) sort fused with the entry gather (stp_tilesort.hip; STP_SORT selects the
// alternatives, see stp_forward).  Everything is enqueued on the caller's stream; the only host synchronisation is
// the read-back.  The calls are re-entrant: the scratch buffers belong to the caller, stp_last_error is per thread, the
// per-device helpers (mailbox ring, side stream, binning-size guesses) are created once behind acquire/release flags,
// and the optional stage timer is one instance PER DEVICE behind a mutex (a backward is attributed to the latest
// forward of its device: meant for one timed caller per device -- bench.py, the viewer's timings text).
// Environment switches (STP_SORT, STP_BINNING, STP_SIDE_STREAM, STP_KBUFFER) select code paths and are read ONCE, at the
// first forward of the process.
#include "stp_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace stp {

static thread_local std::string g_last_error;
static bool g_timing = false;

// Counterpart of the reference's Timer (rasterizer_impl.h:77-147): hipEvents around the stages of every call,
// recorded on the call's stream.  A ring of event sets lets a whole timed region run without any extra host
// synchronisation; spans are harvested lazily and averaged (mean over the calls since stp_timing_enable(1)).
struct StageTimer {
    static constexpr int SETS = 64, EV = 8; // events 0..4: forward stage boundaries, 5..7: backward
    static constexpr int HIST = 1024;       // per-call stage times kept since the last reset (stp_timing_history)
    struct Set { hipEvent_t ev[EV]; bool have[EV]; bool used; long seq; std::chrono::steady_clock::time_point host[EV]; };
    Set sets[SETS] = {};
    bool created = false;
    int cur = 0;
    double sum[6] = {};
    long cnt[6] = {};
    long failures = 0; // hipEventCreate / Record failures since the last reset (surfaced by stp_timing_read)
    long calls = 0;    // forwards begun since the last reset
    float hist[HIST][6]; // stage times of call (seq mod HIST), -1 = not measured
    float hist_host[HIST][6]; // ... and the HOST time between recording the stage's two events (the launching thread's own time in that part of the call)
    void ensure()
    {
        if (created) return;
        for (auto& s : sets) { for (auto& e : s.ev) if (hipEventCreate(&e) != hipSuccess) failures++; for (auto& h : s.have) h = false; s.used = false; }
        created = true;
    }
    void harvest(Set& s)
    {
        if (!s.used) return;
        static const int from[6] = {0, 1, 2, 3, 5, 6}, to[6] = {1, 2, 3, 4, 6, 7};
        for (int i = 0; i < 6; i++) {
            if (!(s.have[from[i]] && s.have[to[i]])) continue;
            if (hipEventSynchronize(s.ev[to[i]]) != hipSuccess) continue;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, s.ev[from[i]], s.ev[to[i]]) == hipSuccess) {
                sum[i] += ms; cnt[i]++; hist[s.seq % HIST][i] = ms;
                hist_host[s.seq % HIST][i] = std::chrono::duration<float, std::milli>(s.host[to[i]] - s.host[from[i]]).count();
            }
        }
        for (auto& h : s.have) h = false;
        s.used = false;
    }
    void begin_forward()
    {
        if (!g_timing) return;
        ensure();
        cur = (cur + 1) % SETS;
        harvest(sets[cur]); // only blocks if the ring wrapped around unharvested work
        sets[cur].used = true;
        sets[cur].seq = calls++;
        for (auto& v : hist[sets[cur].seq % HIST]) v = -1.0f;
        for (auto& v : hist_host[sets[cur].seq % HIST]) v = -1.0f;
    }
    void begin_backward()
    {
        if (!g_timing) return;
        ensure();
        sets[cur].used = true;
        for (int i = 5; i < EV; i++) sets[cur].have[i] = false;
    }
    void mark(int i, hipStream_t st)
    {
        if (!g_timing) return;
        ensure();
        if (hipEventRecord(sets[cur].ev[i], st) != hipSuccess) { failures++; return; }
        sets[cur].host[i] = std::chrono::steady_clock::now();
        sets[cur].have[i] = true;
    }
    void reset()
    {
        if (created) for (auto& s : sets) { for (auto& h : s.have) h = false; s.used = false; }
        for (auto& v : sum) v = 0.0;
        for (auto& c : cnt) c = 0;
        failures = 0;
        calls = 0;
    }
};
// One timer per device (its events live on that device; a backward is attributed to the latest forward OF ITS DEVICE), all
// behind one mutex: timing is a debugging aid, the lock is uncontended in the single-threaded use the reference knows.
static StageTimer g_timers[32];
static std::mutex g_timer_mutex;
static StageTimer& current_timer()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) d = 0;
    return g_timers[d];
}
struct TimerFacade { // keeps the call sites short: g_timer.mark(...) locks and forwards to the current device's timer
    void begin_forward() { if (!g_timing) return; std::lock_guard<std::mutex> l(g_timer_mutex); current_timer().begin_forward(); }
    void begin_backward() { if (!g_timing) return; std::lock_guard<std::mutex> l(g_timer_mutex); current_timer().begin_backward(); }
    void mark(int i, hipStream_t st) { if (!g_timing) return; std::lock_guard<std::mutex> l(g_timer_mutex); current_timer().mark(i, st); }
};
static TimerFacade g_timer;

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
static int fail_hip(hipError_t e, const char* what)
{
    return fail(STP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

GeometryState carve_geometry(char* base, size_t P, bool with_inv, size_t* total, NamedOffset* names, int* n_names)
{
    Carver c(base);
    GeometryState g{};
    size_t off;
    int n = 0;
    auto note = [&](const char* nm, size_t o, size_t cnt) { if (names) names[n] = {nm, o, cnt}; n++; };
    g.status = c.take<uint32_t>(64, &off);
    g.depths = c.take<float>(P, &off); note("depths", off, P);
    g.clamped = c.take<uint8_t>(3 * P, &off); note("clamped", off, 3 * P);
    g.internal_radii = c.take<int32_t>(P, &off); note("radii", off, P);
    g.rects2D = c.take<float2>(P, &off); note("rects2D", off, 2 * P);
    g.means2D = c.take<float2>(P, &off); note("means2D", off, 2 * P);
    g.cov3D = c.take<float>(6 * P, &off); note("cov3D", off, 6 * P);
    if (with_inv) { g.cov3D_inv = c.take<float4>(3 * P, &off); note("cov3D_inv", off, 12 * P); }
    if (with_inv) { g.gpack = c.take<float4>(4 * P, &off); note("gpack", off, 16 * P); }
    g.conic_opacity = c.take<float4>(P, &off); note("conic_opacity", off, 4 * P);
    g.rgb = c.take<float>(3 * P, &off); note("rgb", off, 3 * P);
    g.tiles_touched = c.take<uint32_t>(P, &off); note("tiles_touched", off, P);
    g.point_offsets = c.take<uint32_t>(P, &off); note("point_offsets", off, P);
    g.block_sums = c.take<uint32_t>((P + 255) / 256, &off);
    g.block_prefix = c.take<uint32_t>((P + 255) / 256, &off);
    g.scan_temp_bytes = scan_temp_bytes(P);
    g.scan_temp = c.take<char>(g.scan_temp_bytes);
    if (total) *total = c.total();
    if (n_names) *n_names = n;
    return g;
}

// The image-side state covers the frame's TILE-ROW WINDOW only (StpSettings::tile_y0 / tile_y1; the whole frame by default): a rank of a
// tile-row shard holds 1 / N of the per-pixel arrays and of the blend log (4.3 GB per frame at 4K), not the whole frame's.  The kernels keep
// indexing by frame coordinates (pixel id W * y + x, tile id gx * ty + tx): the sub-array pointers handed to them are shifted back by the
// window's first pixel row / tile, so that index -> address is unchanged inside the window and nothing outside it is ever touched (every
// loop over tiles runs over [gx * ty0, gx * ty1), every kernel's grid over the window's tiles).
ImageState carve_image(char* base, int W, int H, int ty0, int ty1, int log_depth, size_t* total, NamedOffset* names, int* n_names)
{
    Carver c(base);
    ImageState s{};
    size_t off;
    int n = 0;
    auto note = [&](const char* nm, size_t o, size_t cnt) { if (names) names[n] = {nm, o, cnt}; n++; };
    const int gx = (W + TILE - 1) / TILE;
    const int py0 = ty0 * TILE < H ? ty0 * TILE : H, py1 = ty1 * TILE < H ? ty1 * TILE : H;
    const size_t N = (size_t)W * (size_t)(py1 > py0 ? py1 - py0 : 0), T = (size_t)gx * (size_t)(ty1 > ty0 ? ty1 - ty0 : 0);
    s.header = c.take<uint32_t>(64, &off); note("header", off, 4); // first 256 bytes of the buffer, whatever the frame and the log's depth
    s.final_T = c.take<float>(N, &off); note("final_T", off, N);
    s.n_contrib = c.take<uint32_t>(N, &off); note("n_contrib", off, N);
    s.ranges = c.take<uint2>(T, &off); note("ranges", off, 2 * T);
    s.dbg_minmax = c.take<uint32_t>(2, &off); note("dbg_minmax", off, 2);
    s.tile_counts = c.take<uint32_t>(T, &off); note("tile_counts", off, T);
    s.tile_cursor = c.take<uint32_t>(T, &off); note("tile_cursor", off, T);
    s.bin_total = c.take<uint32_t>(2, &off); note("bin_total", off, 2);
    // tile_flags is ALWAYS there: a forward that records no log marks every tile "no valid log" (0xFFFFFFFF), so a backward
    // that is (wrongly) told a log exists -- e.g. after a render_depth forward -- replays nothing and re-sorts every tile
    // instead of reading a log that was never allocated.
    s.tile_flags = c.take<uint32_t>(T, &off); note("tile_flags", off, T);
    const size_t recs_per_tile = 4 * (size_t)blend_log_rows(log_depth) * 64; // 4 waves x (depth + spare) records x 64 lanes, 2 B each
    s.log_depth = log_depth;
    if (log_depth > 0) { // blend log of the recording forward: [tile][wave][record][lane]
        const size_t recs = T * recs_per_tile;
        s.blend_log = c.take<uint32_t>(recs / 2, &off); note("blend_log", off, recs);
    }
    if (total) *total = c.total();
    if (n_names) *n_names = n;
    if (base) { // frame-coordinate indexing (see above)
        const size_t pix0 = (size_t)W * (size_t)py0, tile0 = (size_t)gx * (size_t)ty0;
        s.final_T -= pix0; s.n_contrib -= pix0;
        s.ranges -= tile0; s.tile_counts -= tile0; s.tile_cursor -= tile0; s.tile_flags -= tile0;
        if (s.blend_log) s.blend_log -= tile0 * (recs_per_tile / 2);
    }
    return s;
}

BinningState carve_binning(char* base, size_t R, size_t* total, NamedOffset* names, int* n_names)
{
    Carver c(base);
    BinningState b{};
    size_t off;
    int n = 0;
    auto note = [&](const char* nm, size_t o, size_t cnt) { if (names) names[n] = {nm, o, cnt}; n++; };
    b.header = c.take<uint32_t>(64, &off); note("header", off, 4);
    b.point_list = c.take<uint32_t>(R, &off); note("point_list", off, R);
    b.point_list_unsorted = c.take<uint32_t>(R, &off); note("point_list_unsorted", off, R);
    b.keys = c.take<uint64_t>(R, &off); note("keys", off, R);
    b.keys_unsorted = c.take<uint64_t>(R, &off); note("keys_unsorted", off, R);
    b.sort_temp_bytes = sort_temp_bytes(R);
    b.sort_temp = c.take<char>(b.sort_temp_bytes);
    b.entA = c.take<float4>(R, &off); note("entA", off, 4 * R);
    b.entB = c.take<float4>(R, &off); note("entB", off, 4 * R);
    b.entC = c.take<float4>(R, &off); note("entC", off, 4 * R);
    b.entD = c.take<float4>(R, &off); note("entD", off, 4 * R);
    b.entF = c.take<float4>(R, &off); note("entF", off, 4 * R);
    if (total) *total = c.total();
    if (n_names) *n_names = n;
    return b;
}

static int check_settings(const StpSettings& s, bool backward)
{
    if (s.sort_mode < MODE_GLOBAL || s.sort_mode > MODE_HIER) return fail(STP_ERR_SORT_MODE, "invalid sort mode");
    if (s.sort_order < ORDER_Z || s.sort_order > ORDER_PTD_MAX) return fail(STP_ERR_SORT_MODE, "invalid sort order");
    if (backward && s.sort_mode == MODE_FULL) return fail(STP_ERR_NO_BACKWARD, "Backward not supported for full per-pixel sort");
    if (s.sort_mode == MODE_HIER) {
        const int h = s.queue_per_pixel, m = s.queue_tile_2x2;
        if (!(m == 8 || m == 12 || m == 20)) return fail(STP_ERR_QUEUE_SIZE, "Not supported mid queue size");
        const bool head_ok = backward ? (h == 4 || h == 8 || h == 12 || h == 16) : (h == 4 || h == 8 || h == 16);
        if (!head_ok) return fail(STP_ERR_QUEUE_SIZE, "Not supported head queue size");
    }
    return 0;
}

static void fill_frame(FrameParams& f, int P, int D, int M, const float* background, int width, int height, const StpSettings& s,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, int prefiltered)
{
    f.P = P; f.D = D; f.M = M; f.W = width; f.H = height;
    f.gx = (width + TILE - 1) / TILE; f.gy = (height + TILE - 1) / TILE;
    f.ty0 = 0; f.ty1 = f.gy;
    if (s.tile_y1 > 0) {
        f.ty0 = s.tile_y0 < 0 ? 0 : (s.tile_y0 > f.gy ? f.gy : s.tile_y0);
        f.ty1 = s.tile_y1 > f.gy ? f.gy : s.tile_y1;
        if (f.ty1 < f.ty0) f.ty1 = f.ty0;
    }
    f.focal_y = (float)height / (2.0f * tan_fovy); // reference rasterizer_impl.cu:251-252
    f.focal_x = (float)width / (2.0f * tan_fovx);
    f.tan_fovx = tan_fovx; f.tan_fovy = tan_fovy; f.scale_modifier = scale_modifier; f.s = s;
    f.background = background; f.means3D = means3D; f.shs = shs; f.colors_precomp = colors_precomp; f.opacities = opacities;
    f.scales = scales; f.rotations = rotations; f.cov3D_precomp = cov3D_precomp; f.viewmatrix = viewmatrix; f.projmatrix = projmatrix;
    f.inv_viewprojmatrix = inv_viewprojmatrix; f.cam_pos = cam_pos; f.prefiltered = prefiltered;
    f.wild_cov = 1; // (until the forward has read the status word: the kernels with the domain check)
    f.log_depth = 0; f.log_need = nullptr; f.log_tag = 0;
}

} // namespace stp

using namespace stp;

#define STP_TRY(expr, what)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return fail_hip(_e, what);      \
    } while (0)
#define STP_DEBUG_SYNC(what)                                                     \
    do {                                                                         \
        if (debug) {                                                             \
            hipError_t _e = hipStreamSynchronize(st);                            \
            if (_e != hipSuccess) return fail_hip(_e, what);                     \
        }                                                                        \
    } while (0)

// ---- num_rendered mailbox: host-mapped pinned words + an event, a small ring per device (concurrent forwards on one
// ---- device -- several streams or threads -- each get their own slot)
namespace {
// (`done`: recorded on the side stream behind this forward's SH -> RGB kernel -- one per slot, so that concurrent forwards
// on one device do not re-record each other's event)
struct Mailbox { volatile uint32_t* host = nullptr; uint32_t* dev = nullptr; hipEvent_t ev = nullptr; hipEvent_t done = nullptr; int device = 0; uint32_t ticket = 0; };
constexpr int MAX_DEVICES = 32, MAILBOX_RING = 8;
struct MailboxRing { Mailbox slot[MAILBOX_RING]; std::atomic<unsigned> next{0}; std::atomic<bool> ready{false};
                     uint32_t* log_need = nullptr; /* device words, one per guess slot: report_log_need (stp_blend.h) */ };
MailboxRing g_mailboxes[MAX_DEVICES];
std::mutex g_mailbox_mutex;
// Binning-size guesses: tile-list entries of the previous forward OF THE SAME KIND on each device.  "Kind" = (P, width,
// height, tile-row window, sort mode): a small frame that follows a 4K frame (an eval render between training steps, a
// second rasterizer module) does not inherit the big frame's count.  Direct-mapped, 16 kinds per device; a collision only
// costs the second allocator call.
constexpr int GUESS_SLOTS = 16;
struct SizeGuess { std::atomic<uint64_t> key{0}; std::atomic<uint32_t> R{0}; std::atomic<uint32_t> log_need{0}; }; // log_need: blends per pixel the kind's recording forwards needed
SizeGuess g_guess[MAX_DEVICES][GUESS_SLOTS];
uint64_t guess_key(const FrameParams& f)
{
    uint64_t k = 0x9E3779B97F4A7C15ull;
    for (uint64_t v : {(uint64_t)f.P, (uint64_t)f.W, (uint64_t)f.H, (uint64_t)f.ty0, (uint64_t)f.ty1, (uint64_t)f.s.sort_mode,
                       (uint64_t)(f.s.tile_based_culling * 8 + f.s.rect_bounding * 4 + f.s.tight_opacity_bounding * 2 + (f.s.sort_order == ORDER_PTD_MAX))})
        k = (k ^ v) * 0xBF58476D1CE4E5B9ull, k ^= k >> 29;
    return k | 1ull;
}

// A second stream per device for the SH -> RGB kernel: nothing before the entry gather needs the colours, so the kernel (a
// pure HBM stream, 70 us at C2) runs BESIDE the host hand-over, duplicate and the tile-bit sort (atomics, small launches and
// 1.6 TB/s radix passes) instead of in front of them.  It starts behind the mailbox event and is joined back into the
// caller's stream before the first reader of the colours -- and on every early return, so the caller's buffers are never
// touched by work the caller's stream does not know about.  STP_SIDE_STREAM=0: everything on the caller's stream.
struct SideStream { hipStream_t stream = nullptr; std::atomic<bool> ready{false}; };
SideStream g_side[MAX_DEVICES];
SideStream* side_stream(int device)
{
    static const char* const env = std::getenv("STP_SIDE_STREAM");
    static const bool off = env && std::strcmp(env, "0") == 0;
    if (off || device < 0 || device >= MAX_DEVICES) return nullptr;
    SideStream& s = g_side[device];
    if (!s.ready.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_mailbox_mutex);
        if (!s.ready.load(std::memory_order_relaxed)) {
            if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            s.ready.store(true, std::memory_order_release);
        }
    }
    return &s;
}
struct SideJoin { // joins the side stream's work into `st` when it goes out of scope, unless done earlier
    hipEvent_t done; hipStream_t st; bool pending;
    hipError_t join() { if (!pending) return hipSuccess; pending = false; return hipStreamWaitEvent(st, done, 0); }
    ~SideJoin() { (void)join(); }
};

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}

// Which entry count a binning buffer was CARVED with, and which depth an image buffer's blend log.  A run-ahead forward (stp_forward) carves
// and launches on a capacity before num_rendered is known; the sub-arrays of the buffer then sit at the offsets of that capacity, not of the
// count stp_forward returns; and the blend log's depth is chosen per frame.  The backward and the introspection helpers are handed (pointer,
// num_rendered) only, as in the reference -- whose buffers are self-contained blobs.  Ours are too: every forward writes a HEADER into the
// buffer itself (device side, no extra launch: duplicate_kernel / frame_init_kernel) --
//     binning: first 256 bytes  {STP_HEADER_MAGIC_BINNING, capacity, ~capacity, 0}
//     image:   first 256 bytes  {STP_HEADER_MAGIC_IMAGE, depth of the blend log (0: none), ~depth, 0}
// -- and the HOST keeps a cache pointer -> value so that the backward of the same process needs no read-back (one entry per buffer address,
// overwritten whenever a forward carves that address again; least-recently-used entries are dropped in batches).  An entry also remembers the
// num_rendered of its forward, and the backward -- which is handed num_rendered -- takes it only if that matches: an address the allocator has
// re-issued for somebody else's buffer (a clone of another forward's buffers) does not get the previous tenant's layout.  A pointer the cache
// does not know -- a buffer that was cloned, copied, moved, or whose entry was dropped -- is looked up in the buffer's own header (one blocking
// 16-byte copy: the rare path); a buffer without a valid header is REFUSED (STP_ERR_INVALID_ARGUMENT) instead of being carved on a guess.
struct LayoutCache {
    struct Entry { uint32_t value; int64_t R; uint64_t tick; };
    std::unordered_map<const void*, Entry> map;
    uint64_t tick = 0;
    static constexpr size_t CAP = 8192;
    void put(const void* p, uint32_t v, int64_t R)
    {
        if (map.size() >= CAP && map.find(p) == map.end()) { // drop the least recently used quarter (forwards whose buffers nobody came back for)
            std::vector<uint64_t> t; t.reserve(map.size());
            for (const auto& kv : map) t.push_back(kv.second.tick);
            std::nth_element(t.begin(), t.begin() + t.size() / 4, t.end());
            const uint64_t cut = t[t.size() / 4];
            for (auto it = map.begin(); it != map.end();) it = it->second.tick <= cut ? map.erase(it) : std::next(it);
        }
        map[p] = Entry{v, R, ++tick};
    }
    bool get(const void* p, int64_t R, uint32_t* v) // R < 0: whatever forward carved the address last (introspection right behind a forward)
    {
        const auto it = map.find(p);
        if (it == map.end() || (R >= 0 && it->second.R != R)) return false;
        it->second.tick = ++tick;
        *v = it->second.value;
        return true;
    }
};
std::mutex g_layout_mutex;
LayoutCache g_layout, g_log_depth;
void remember_layout(const void* binning, uint32_t count, int64_t R) { std::lock_guard<std::mutex> l(g_layout_mutex); g_layout.put(binning, count, R); }
void remember_log_depth(const void* image, uint32_t depth, int64_t R) { std::lock_guard<std::mutex> l(g_layout_mutex); g_log_depth.put(image, depth, R); }
// the header a forward left in the buffer: 0 and *value on success, else a negative STP_ERR_* (message set)
// (the copy is ordered on the CALLER's stream -- a clone made on a non-blocking stream is not visible to the null stream's copy -- and waited for;
//  introspection calls have no stream: they wait for the device first)
int read_buffer_header(const uint32_t* dev_header, uint32_t magic, const char* what, uint32_t* value, hipStream_t st, bool have_stream)
{
    uint32_t h[4] = {0, 0, 0, 0};
    if (!have_stream) (void)hipDeviceSynchronize();
    if (hipMemcpyAsync(h, dev_header, sizeof(h), hipMemcpyDeviceToHost, have_stream ? st : nullptr) != hipSuccess ||
        hipStreamSynchronize(have_stream ? st : nullptr) != hipSuccess) { (void)hipGetLastError(); return fail(STP_ERR_HIP, std::string("could not read the header of the ") + what + " buffer"); }
    if (h[0] != magic || h[2] != ~h[1])
        return fail(STP_ERR_INVALID_ARGUMENT, std::string("the ") + what + " buffer does not carry a header of this library: it was not written by stp_forward (or has been overwritten)");
    *value = h[1];
    return 0;
}
// entries the binning buffer was carved for: cache, else the buffer's own header
int layout_of(const char* binning, uint32_t R, uint32_t* cap, hipStream_t st = nullptr, bool have_stream = false)
{
    {
        std::lock_guard<std::mutex> l(g_layout_mutex);
        if (g_layout.get(binning, (int64_t)R, cap) && *cap >= R) return 0;
    }
    if (int rc = read_buffer_header(reinterpret_cast<const uint32_t*>(binning), STP_HEADER_MAGIC_BINNING, "binning", cap, st, have_stream)) return rc;
    if (*cap < R) return fail(STP_ERR_INVALID_ARGUMENT, "the binning buffer was carved for fewer entries than num_rendered");
    remember_layout(binning, *cap, (int64_t)R);
    return 0;
}

// run-ahead forward: 0 = never, 1 = whenever a size guess exists, 2 (default) = for SMALL frames only (guess below RUN_AHEAD_AUTO_MAX entries)
constexpr uint32_t RUN_AHEAD_AUTO_MAX = 1u << 18;
std::atomic<int> g_run_ahead{[] { const char* e = std::getenv("STP_RUN_AHEAD"); return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : 2; }()};

// depth the image buffer's blend log was carved with: cache, else the buffer's own header (whose offset does not depend on the depth)
int log_depth_of(const char* image, int64_t R, uint32_t* depth, hipStream_t st = nullptr, bool have_stream = false)
{
    {
        std::lock_guard<std::mutex> l(g_layout_mutex);
        if (g_log_depth.get(image, R, depth)) return 0;
    }
    if (int rc = read_buffer_header(reinterpret_cast<const uint32_t*>(image), STP_HEADER_MAGIC_IMAGE, "image", depth, st, have_stream)) return rc;
    if (*depth != 0u && (int)*depth != blend_log_clamp_depth((int)*depth)) return fail(STP_ERR_INVALID_ARGUMENT, "the image buffer's header holds an impossible blend-log depth");
    remember_log_depth(image, *depth, R);
    return 0;
}
// Depth of this frame's blend log: the largest blend count per pixel that the recording forwards of this kind reported (slowly forgotten:
// read_mailbox), + 12.5 % + 4, rounded up to 16 records; a frame nothing is known about gets the default.  STP_LOG_DEPTH=n fixes it.
int log_depth_for(const SizeGuess& slot, uint64_t key)
{
    static const int fixed = [] { const char* e = std::getenv("STP_LOG_DEPTH"); return e ? std::atoi(e) : 0; }();
    if (fixed > 0) return blend_log_clamp_depth((fixed + 1) & ~1);
    const uint32_t need = slot.key.load(std::memory_order_acquire) == key ? slot.log_need.load(std::memory_order_relaxed) : 0u;
    if (need == 0) return blend_log_default_depth();
    return blend_log_clamp_depth((int)((need + need / 8 + 4 + 15) & ~15u));
}

int acquire_mailbox(Mailbox* out)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= MAX_DEVICES) return fail(STP_ERR_HIP, "hipGetDevice failed");
    MailboxRing& ring = g_mailboxes[device];
    if (!ring.ready.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_mailbox_mutex);
        if (!ring.ready.load(std::memory_order_relaxed)) {
            for (int i = 0; i < MAILBOX_RING; i++) {
                void* h = nullptr; void* d = nullptr;
                if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess ||
                    hipEventCreateWithFlags(&ring.slot[i].ev, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ring.slot[i].done, hipEventDisableTiming) != hipSuccess)
                    return fail(STP_ERR_HIP, "cannot create the num_rendered mailbox");
                std::memset(h, 0, 64); // (a recycled pinned page may hold an old ticket: the first tickets are the small integers 1..8)
                ring.slot[i].host = static_cast<volatile uint32_t*>(h);
                ring.slot[i].dev = static_cast<uint32_t*>(d);
                ring.slot[i].device = device;
            }
            if (hipMalloc(reinterpret_cast<void**>(&ring.log_need), sizeof(uint32_t) * 64) != hipSuccess || hipMemset(ring.log_need, 0, sizeof(uint32_t) * 64) != hipSuccess)
                return fail(STP_ERR_HIP, "cannot create the blend-log depth words");
            ring.ready.store(true, std::memory_order_release);
        }
    }
    const unsigned seq = ring.next.fetch_add(1u);
    *out = ring.slot[seq % MAILBOX_RING];
    out->ticket = seq + 1u == 0u ? 1u : seq + 1u; // what the kernel writes LAST into the slot's third word: unique per use of the slot
    return 0;
}
} // namespace

extern "C" {

int stp_abi_version(void) { return STP_ABI_VERSION; }
const char* stp_last_error(void) { return g_last_error.c_str(); }

size_t stp_geometry_buffer_size(int P, const StpSettings* settings)
{
    size_t total = 0;
    carve_geometry(nullptr, (size_t)P, settings ? requires_depth_along_ray(*settings) : true, &total);
    return total;
}
size_t stp_binning_buffer_size(int R)
{
    size_t total = 0;
    carve_binning(nullptr, (size_t)(R > 0 ? R : 0), &total);
    return total;
}
size_t stp_image_buffer_size(int width, int height)
{
    size_t total = 0;
    carve_image(nullptr, width, height, 0, (height + TILE - 1) / TILE, 0, &total);
    return total;
}

static void clamp_rows(int height, int& y0, int& y1) // the window fill_frame derives from StpSettings::tile_y0 / tile_y1
{
    const int gy = (height + TILE - 1) / TILE;
    if (y1 <= 0) { y0 = 0; y1 = gy; return; }
    y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0);
    y1 = y1 > gy ? gy : y1;
    if (y1 < y0) y1 = y0;
}

size_t stp_blend_log_bytes_rows(int width, int height, int tile_y0, int tile_y1)
{
    clamp_rows(height, tile_y0, tile_y1);
    size_t plain = 0, with_log = 0;
    carve_image(nullptr, width, height, tile_y0, tile_y1, 0, &plain);
    carve_image(nullptr, width, height, tile_y0, tile_y1, blend_log_default_depth(), &with_log); // (a frame nothing is known about: see stp_raster.h)
    return with_log - plain;
}

size_t stp_blend_log_bytes(int width, int height) { return stp_blend_log_bytes_rows(width, height, 0, 0); }
size_t stp_blend_log_bytes_depth(int width, int height, int tile_y0, int tile_y1, int depth) // depth <= 0: the deepest log a forward may carve
{
    clamp_rows(height, tile_y0, tile_y1);
    size_t plain = 0, with_log = 0;
    carve_image(nullptr, width, height, tile_y0, tile_y1, 0, &plain);
    carve_image(nullptr, width, height, tile_y0, tile_y1, depth > 0 ? blend_log_clamp_depth(depth) : blend_log_clamp_depth(1 << 30), &with_log);
    return with_log - plain;
}

static int find_name(const NamedOffset* names, int n, const char* name, size_t* offset, size_t* count)
{
    for (int i = 0; i < n; i++)
        if (std::strcmp(names[i].name, name) == 0) {
            if (offset) *offset = names[i].offset;
            if (count) *count = names[i].count;
            return 0;
        }
    return fail(STP_ERR_INVALID_ARGUMENT, std::string("unknown sub-array name: ") + name);
}
int stp_geometry_layout(int P, const StpSettings* settings, const char* name, size_t* offset, size_t* count)
{
    NamedOffset names[24]; int n = 0;
    carve_geometry(nullptr, (size_t)P, settings ? requires_depth_along_ray(*settings) : true, nullptr, names, &n);
    return find_name(names, n, name, offset, count);
}
int stp_binning_layout(int R, const char* name, size_t* offset, size_t* count)
{
    NamedOffset names[16]; int n = 0;
    carve_binning(nullptr, (size_t)(R > 0 ? R : 0), nullptr, names, &n);
    return find_name(names, n, name, offset, count);
}
void stp_set_run_ahead(int mode) { g_run_ahead.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed); }
int stp_get_run_ahead(void) { return g_run_ahead.load(std::memory_order_relaxed); }
void stp_reset_size_guesses(void)
{
    for (auto& dev : g_guess)
        for (auto& slot : dev) { slot.key.store(0, std::memory_order_release); slot.R.store(0u, std::memory_order_relaxed); slot.log_need.store(0u, std::memory_order_relaxed); }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;
    for (int d = 0; d < MAX_DEVICES; d++) // ... and what recording forwards have reported but no forward has collected yet
        if (g_mailboxes[d].ready.load(std::memory_order_acquire) && g_mailboxes[d].log_need && hipSetDevice(d) == hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipMemset(g_mailboxes[d].log_need, 0, sizeof(uint32_t) * 64);
        }
    (void)hipSetDevice(cur);
}
void stp_forget_buffer(const void* buffer)
{
    if (!buffer) return;
    std::lock_guard<std::mutex> l(g_layout_mutex);
    g_layout.map.erase(buffer);
    g_log_depth.map.erase(buffer);
}
int stp_blend_log_depth(const void* image_buffer)
{
    if (!image_buffer) return fail(STP_ERR_INVALID_ARGUMENT, "null image buffer");
    uint32_t d = 0;
    if (int rc = log_depth_of((const char*)image_buffer, -1, &d)) return rc;
    return (int)d;
}
int stp_binning_layout_count(const void* binning_buffer, int R)
{
    if (!binning_buffer) return fail(STP_ERR_INVALID_ARGUMENT, "null binning buffer");
    uint32_t cap = 0;
    if (int rc = layout_of((const char*)binning_buffer, (uint32_t)(R > 0 ? R : 0), &cap)) return rc;
    return (int)cap;
}
int stp_image_layout_rows(int width, int height, int tile_y0, int tile_y1, const char* name, size_t* offset, size_t* count)
{
    NamedOffset names[16]; int n = 0;
    clamp_rows(height, tile_y0, tile_y1);
    carve_image(nullptr, width, height, tile_y0, tile_y1, blend_log_default_depth(), nullptr, names, &n);
    return find_name(names, n, name, offset, count);
}
int stp_image_layout(int width, int height, const char* name, size_t* offset, size_t* count)
{
    return stp_image_layout_rows(width, height, 0, 0, name, offset, count);
}
int stp_image_layout_depth(int width, int height, int tile_y0, int tile_y1, int log_depth, const char* name, size_t* offset, size_t* count)
{
    NamedOffset names[16]; int n = 0;
    clamp_rows(height, tile_y0, tile_y1);
    carve_image(nullptr, width, height, tile_y0, tile_y1, log_depth > 0 ? blend_log_clamp_depth(log_depth) : 0, nullptr, names, &n);
    return find_name(names, n, name, offset, count);
}

void stp_timing_enable(int enabled)
{
    std::lock_guard<std::mutex> l(g_timer_mutex);
    g_timing = enabled != 0;
    if (g_timing) {
        for (auto& t : g_timers) t.reset();
        current_timer().ensure(); // the calling thread's device: its 512 events exist before the first timed call (creating them inside it
                                  // put 2-3 ms of driver calls into the first step of a timed region)
    }
}

int stp_timing_read(float* ms6) // the calling thread's current device
{
    if (!ms6) return fail(STP_ERR_INVALID_ARGUMENT, "null output");
    for (int i = 0; i < 6; i++) ms6[i] = -1.0f;
    std::lock_guard<std::mutex> l(g_timer_mutex);
    StageTimer& t = current_timer();
    if (!t.created) return 0;
    for (auto& s : t.sets) t.harvest(s);
    // 0 Preprocess (+scan+read-back), 1 Duplicate, 2 Sort (+ranges), 3 Render, 4 BwdRender, 5 BwdPreprocess
    for (int i = 0; i < 6; i++)
        if (t.cnt[i] > 0) ms6[i] = (float)(t.sum[i] / (double)t.cnt[i]);
    if (t.failures > 0) return fail(STP_ERR_HIP, "stage timer: " + std::to_string(t.failures) + " hipEvent create/record call(s) failed; timings are incomplete");
    return 0;
}

static int timing_history(float* ms6, int capacity, bool host);
int stp_timing_history(float* ms6, int capacity) { return timing_history(ms6, capacity, false); } // the calling thread's current device
int stp_timing_history_host(float* ms6, int capacity) { return timing_history(ms6, capacity, true); }
static int timing_history(float* ms6, int capacity, bool host)
{
    if (!ms6 || capacity < 0) return fail(STP_ERR_INVALID_ARGUMENT, "null output");
    std::lock_guard<std::mutex> l(g_timer_mutex);
    StageTimer& t = current_timer();
    if (!t.created) return 0;
    for (auto& s : t.sets) t.harvest(s);
    const long n = std::min<long>(std::min<long>(t.calls, StageTimer::HIST), capacity);
    for (long k = 0; k < n; k++) // chronological: the last n calls
        std::memcpy(ms6 + 6 * k, (host ? t.hist_host : t.hist)[(t.calls - n + k) % StageTimer::HIST], 6 * sizeof(float));
    return (int)n;
}

size_t stp_timing_text(char* buf, size_t size)
{
    float ms[6];
    (void)stp_timing_read(ms);
    static const char* names[6] = {"Preprocess", "Duplicate", "Sort", "Render", "BwdRender", "BwdPreprocess"};
    std::string text = "Timings: \n";
    float total = 0.0f;
    char line[96];
    for (int i = 0; i < 4; i++) {
        const float v = ms[i] >= 0.0f ? ms[i] : 0.0f;
        std::snprintf(line, sizeof(line), " - %s: %gms\n", names[i], v);
        text += line;
        total += v;
    }
    std::snprintf(line, sizeof(line), " - Total: %gms\n", total); // reference Timer::total (rasterizer_impl.h:79,131-134)
    text += line;
    for (int i = 4; i < 6; i++)
        if (ms[i] >= 0.0f) {
            std::snprintf(line, sizeof(line), " - %s: %gms\n", names[i], ms[i]);
            text += line;
        }
    if (buf && size > 0) {
        const size_t n = text.size() < size - 1 ? text.size() : size - 1;
        std::memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return text.size();
}

// stp_set_forward_split: the request of the calling thread for its NEXT stp_forward (consumed there, whatever the call's outcome)
struct ForwardSplit { int row = 0; hipEvent_t event = nullptr; bool armed = false; };
thread_local ForwardSplit t_forward_split;
void stp_set_forward_split(int tile_row, void* event)
{
    t_forward_split.row = tile_row;
    t_forward_split.event = (hipEvent_t)event;
    t_forward_split.armed = event != nullptr;
}

int stp_forward(stp_alloc_fn geometry_alloc, void* geometry_user, stp_alloc_fn binning_alloc, void* binning_user,
                stp_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
                const StpSettings* settings, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* inv_viewprojmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const ForwardSplit split = t_forward_split; // (one forward per request)
    t_forward_split = ForwardSplit{};
    struct SplitGuard { // whatever happens to the call: the caller's event is recorded on its stream behind everything this call enqueued
        const ForwardSplit& s; hipStream_t st; bool done = false;
        ~SplitGuard() { if (s.armed && !done) (void)hipEventRecord(s.event, st); }
    } split_guard{split, st};
    if (!settings || !geometry_alloc || !binning_alloc || !image_alloc) return fail(STP_ERR_INVALID_ARGUMENT, "null settings or allocator");
    if (P < 0 || width <= 0 || height <= 0) return fail(STP_ERR_INVALID_ARGUMENT, "bad sizes");
    if (P == 0) return 0; // reference rasterize_points.cu:93 -- nothing launched, caller's zero image stands
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !inv_viewprojmatrix || !cam_pos || !out_color)
        return fail(STP_ERR_INVALID_ARGUMENT, "null required input");
    if (int rc = check_settings(*settings, false)) return rc;
    if (!colors_precomp && !shs) return fail(STP_ERR_INVALID_ARGUMENT, "neither SHs nor precomputed colours given");
    if (!cov3D_precomp && !(scales && rotations)) return fail(STP_ERR_INVALID_ARGUMENT, "neither scale/rotation nor precomputed covariance given");
    const bool with_inv = requires_depth_along_ray(*settings);
    if (with_inv && !(scales && rotations)) return fail(STP_ERR_NEEDS_SCALE_ROTATION, "sorted modes need scales and rotations");

    FrameParams f;
    fill_frame(f, P, D, M, background, width, height, *settings, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
               rotations, cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered);

    size_t geom_bytes = 0;
    carve_geometry(nullptr, (size_t)P, with_inv, &geom_bytes);
    char* geom_ptr = (char*)geometry_alloc(geometry_user, geom_bytes);
    if (!geom_ptr) return fail(STP_ERR_ALLOC, "geometry allocator returned NULL");
    GeometryState g = carve_geometry(geom_ptr, (size_t)P, with_inv, nullptr);
    if (!radii) radii = g.internal_radii;

    Mailbox mb; // (taken here already: the device's guess slots size the blend log)
    if (int rc = acquire_mailbox(&mb)) return rc;
    const uint64_t gkey = guess_key(f);
    const unsigned gidx = (unsigned)((gkey >> 1) % GUESS_SLOTS);
    SizeGuess& gslot = g_guess[mb.device][gidx];
    uint32_t* const log_need_word = g_mailboxes[mb.device].log_need + gidx;
    size_t img_bytes = 0;
    const bool with_log = uses_blend_log(*settings);
    const int log_depth = with_log ? log_depth_for(gslot, gkey) : 0;
    f.log_depth = log_depth;
    f.log_need = with_log ? log_need_word : nullptr;
    const uint32_t log_tag = (uint32_t)((gkey >> 40) & 0xFFFFu) | 1u; // (never 0: an empty word carries no tag)
    f.log_tag = log_tag;
    carve_image(nullptr, width, height, f.ty0, f.ty1, log_depth, &img_bytes); // (the tile-row window's share: see carve_image)
    char* img_ptr = (char*)image_alloc(image_user, img_bytes);
    if (!img_ptr) return fail(STP_ERR_ALLOC, "image allocator returned NULL");
    ImageState img = carve_image(img_ptr, width, height, f.ty0, f.ty1, log_depth, nullptr);
    // (the buffer's own header is written by frame_init_kernel; the host-side cache entry follows when num_rendered is known)

    // How the (tile, depth) order is established:
    //   default           device-wide radix sort on the tile bits only (two passes), then the tile's own workgroup sorts its
    //                     segment by (depth, Gaussian id) in LDS
    //   STP_SORT=radix    the reference's single device-wide radix sort on (tile, depth)
    //   STP_SORT=counters no device-wide sort: preprocess counts every tile's entries, duplicate writes each entry straight
    //                     into its tile's segment through an atomic cursor, then the same per-tile sort.  Measured SLOWER on
    //                     MI355X (the 2 x R device-scope atomics cost more than the two radix passes they replace: C2
    //                     preprocess + duplicate + sort 0.54 ms against 0.50 ms); kept selectable and tested.
    // (STP_SORT is read ONCE, at the first forward of the process: it selects code paths, not per-call behaviour)
    static const char* const sort_env = std::getenv("STP_SORT");
    static const bool tile_local_sort = !(sort_env && std::strcmp(sort_env, "radix") == 0);
    static const bool atomic_bin = sort_env && std::strcmp(sort_env, "counters") == 0;

    // STP_SCAN=rocprim: the device-wide scan of round 1-2 (rocPRIM inclusive_scan + a one-thread mailbox kernel) instead of the two-level scan
    // folded into preprocess_kernel / duplicate_kernel (read once, like the other path switches)
    static const char* const scan_env = std::getenv("STP_SCAN");
    static const bool two_level_scan = !(scan_env && std::strcmp(scan_env, "rocprim") == 0);
    if (!two_level_scan) { g.block_sums = nullptr; g.block_prefix = nullptr; }

    g_timer.begin_forward();
    g_timer.mark(0, st);
    STP_TRY(launch_frame_init(g, img, f.gx * f.ty0, f.gx * (f.ty1 - f.ty0), with_log, atomic_bin, st), "frame init launch");
    STP_TRY(launch_preprocess(f, g, radii, atomic_bin ? img.tile_counts : nullptr, st), "preprocess launch");
    STP_DEBUG_SYNC("preprocess");
    if (!two_level_scan) STP_TRY(launch_scan(f, g, st), "inclusive scan");
    STP_DEBUG_SYNC("scan");
    if (atomic_bin) STP_TRY(launch_tile_scan(f, img, st), "tile scan"); // counters -> ranges + cursors (no host value needed)

    // The one mandatory host hand-over: num_rendered sizes the binning buffers (reference :317, a blocking 4-byte copy into
    // pageable memory).  Here a one-thread kernel drops the two words into host-mapped pinned memory and an event marks
    // the spot; the SH -> RGB kernel -- which nothing before the render stage depends on -- is enqueued BEHIND it, so
    // the GPU keeps working while the host wakes up, sizes the buffer and launches duplicate / sort.
    if (two_level_scan) STP_TRY(launch_block_prefix_mailbox(f, g, mb.dev, mb.ticket, log_need_word, st), "workgroup prefixes + mailbox launch");
    else STP_TRY(launch_mailbox(g.point_offsets + (P - 1), g.status + 1, mb.dev, mb.ticket, log_need_word, st), "mailbox launch");
    STP_TRY(hipEventRecord(mb.ev, st), "record mailbox event");
    SideStream* const side = side_stream(mb.device);
    SideJoin colours{mb.done, st, false};
    // Where the colour kernel starts on the side stream.  Rounds 2-3: behind the mailbox, i.e. in the host's hand-over bubble and then beside
    // duplicate_kernel -- two bandwidth-bound kernels that slow each other down (duplicate 61 us alone, 98 us beside it).  Since the host watches the
    // mailbox word the bubble is a few microseconds, and the kernel now starts behind duplicate_kernel, beside the tile-bit sort, whose radix
    // passes run at 1.6 TB/s and leave it room (late round 3, one box, alternating: duplicate 0.098 -> 0.058 ms, sort stage 0.289 -> 0.337, the
    // step -6 .. -10 us at C2-full, -40 .. -70 us at C5, C3 / C4 / C2-min unchanged).  STP_COLOUR_LATE=0 restores the earlier start.
    static const char* const late_env = std::getenv("STP_COLOUR_LATE");
    static const bool colour_late = !(late_env && late_env[0] == '0');
    auto colour_on_side = [&]() -> int {
        STP_TRY(hipStreamWaitEvent(side->stream, mb.ev, 0), "side stream wait");
        STP_TRY(launch_sh_color(f, g, radii, side->stream), "SH colour launch");
        // from here on the side stream works on the caller's buffers: every return path joins it (SideJoin); should the
        // event that the join waits for fail to record, the side stream is drained on the spot instead
        if (hipError_t e = hipEventRecord(mb.done, side->stream); e != hipSuccess) {
            (void)hipStreamSynchronize(side->stream);
            return fail_hip(e, "record colour event");
        }
        colours.pending = true;
        return 0;
    };
    if (side) {
        if (!colour_late) { if (int rc = colour_on_side()) return rc; }
    } else STP_TRY(launch_sh_color(f, g, radii, st), "SH colour launch");
    // The binning buffer is requested BEFORE num_rendered is known, sized by the counts of the previous frames of the same kind on this
    // device (+12.5 %): in the steady state of training or serving no allocator callback runs between the kernels.  The exact-size
    // request of the reference follows only when the guess was too small (STP_BINNING=exact: always) -- so binning_alloc may be called
    // TWICE per forward, the second time with the larger size (include/stp_raster.h).
    static const char* const bin_env = std::getenv("STP_BINNING");
    static const bool speculative = !(bin_env && std::strcmp(bin_env, "exact") == 0);
    // RUN-AHEAD (round 4; by default for small frames only: STP_RUN_AHEAD=0 / 1 in the environment or stp_set_run_ahead(0 / 1 / 2) say never / always / auto).  The reference -- and the
    // default path here -- stop the host after the scan until num_rendered has come back, and only then enqueue duplicate / sort / render:
    // a stall of the launching thread there is GPU idle time.  With a size guess the whole forward is enqueued at once ON THE GUESSED
    // CAPACITY: the sub-arrays are carved for `cap` entries, duplicate_kernel guards its writes and pads [num_rendered, cap) with entries
    // that sort behind every tile, the sort / range passes run over `cap`, the render kernels are the ones for a tame Sigma^-1 -- and the
    // host reads the mailbox AFTER the last launch, when the word has long arrived.  Only a frame that does not fit (or whose status word
    // asks for the checked reciprocal) is redone from duplicate_kernel on with the exact size, before the call returns: results never
    // depend on the guess (tests/test_gpu_parity.py::test_run_ahead_overflow_is_redone; every GpuRun of the tests renders its frame both ways).
    // MEASURED (one box, alternating, profiles/r04_run_ahead_ab.txt): the padding costs the device-wide passes what it weighs -- C2-full sort
    // stage 0.325 -> 0.343 ms, step 2.410 -> 2.422 ms; C5 +0.04 ms; C4 +0.03 ms -- and nothing comes back: the hand-over bubble was already
    // hidden (mailbox word + colour kernel behind it), `ms_per_step - sum(stages)` stays at 0.04 ms, and C1 is bound by the ~25 launches of a
    // step, not by the round trip.  Hence off for large frames by default; what it does buy there is a frame whose GPU time no longer depends on
    // the launching thread being scheduled in the middle of it.
    // Small frames are the exception (mode 2, the default: guesses below 2^18 entries): there the padding weighs nothing and the round trip is
    // a tenth of the frame -- C1 0.179 -> 0.167 ms per step (profiles/r04_host_profile_c1.txt).
    const int run_ahead_mode = g_run_ahead.load(std::memory_order_relaxed);
    const bool run_ahead = run_ahead_mode != 0;
    size_t bin_have = 0;
    char* bin_ptr = nullptr;
    const uint32_t guess = (speculative && gslot.key.load(std::memory_order_acquire) == gkey) ? gslot.R.load(std::memory_order_relaxed) : 0u;
    const bool ahead = run_ahead && guess > 0 && (run_ahead_mode == 1 || guess < RUN_AHEAD_AUTO_MAX) && two_level_scan && !atomic_bin && !debug;
    const uint32_t cap = guess + guess / 8 + (ahead ? 1024u : 0u);
    if (guess > 0) {
        carve_binning(nullptr, (size_t)cap, &bin_have);
        bin_ptr = (char*)binning_alloc(binning_user, bin_have);
        if (!bin_ptr) return fail(STP_ERR_ALLOC, "binning allocator returned NULL");
    }
    // The host watches the slot itself: the kernel's last store (the ticket) is visible a microsecond after it was made, the event behind the
    // kernel is signalled by a barrier packet some microseconds later, and hipEventSynchronize's wake-up adds its own.  The event is still
    // polled now and then: it completes if the kernel has, and it is how a device fault surfaces (STP_MAILBOX=event: wait on the event only).
    static const char* const mbx_env = std::getenv("STP_MAILBOX");
    static const bool mbx_spin = !(mbx_env && std::strcmp(mbx_env, "event") == 0);
    auto wait_mailbox = [&]() -> int {
        if (mbx_spin) {
            for (unsigned it = 1;; it++) {
                if (mb.host[2] == mb.ticket) break;
                if ((it & 255u) == 0u) {
                    const hipError_t q = hipEventQuery(mb.ev);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) return fail_hip(q, "query (num_rendered)");
                    if (it > (1u << 22)) { STP_TRY(hipEventSynchronize(mb.ev), "synchronize (num_rendered)"); break; } // (seconds of spinning: stop burning a core)
                }
                cpu_relax();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else STP_TRY(hipEventSynchronize(mb.ev), "synchronize (num_rendered)");
        return 0;
    };
    // everything behind the hand-over: duplicate -> (colour kernel on the side stream) -> sort -> ranges -> per-tile sort + gather -> render.
    // L = entries the device-wide passes run over: num_rendered, or the capacity of a run-ahead launch (dup_cap = the same value then)
    bool colour_started = !(side && colour_late);
    auto binning_and_render = [&](const GeometryState& gd, const BinningState& b, int L, uint32_t dup_cap) -> int {
        uint32_t* zero_ptr = nullptr; size_t zero_words = 0; // (the tile-bit sort's histograms / look-back states / block counters: cleared here, once)
        if (!atomic_bin && tile_local_sort) sort_zero_region(b, (size_t)L, (uint32_t)(f.gx * f.gy), &zero_ptr, &zero_words);
        STP_TRY(launch_duplicate(f, gd, radii, b, atomic_bin ? img.tile_cursor : nullptr, dup_cap, (uint32_t)L, zero_ptr, zero_words, st), "duplicate launch");
        STP_DEBUG_SYNC("duplicate");
        g_timer.mark(2, st);
        if (!colour_started) {
            STP_TRY(hipEventRecord(mb.ev, st), "record event behind duplicate");
            if (int rc = colour_on_side()) return rc;
            colour_started = true;
        }
        if (atomic_bin) {
            STP_TRY(launch_bin_pad(b, img, L, st), "pad entries");
        } else {
            STP_TRY(launch_sort(f, b, L, tile_local_sort, zero_words != 0, st), "radix sort");
            STP_DEBUG_SYNC("sort");
            STP_TRY(launch_ranges(f, b, img, L, st), "tile ranges");
            STP_DEBUG_SYNC("ranges");
        }
        // The tile order (one workgroup: 7 us at 1080p, 31 us at 4K) needs the ranges and is needed by the render kernel only: on the side stream it
        // runs beside the entry gather.  The mailbox's two events serve a second time: `ev` marks "ranges done" for the side stream, `done` -- re-recorded
        // behind the order kernel AFTER the caller's stream has been told to wait for its first recording, the colour kernel's -- is joined in front of
        // the render launch.  MEASURED (one box, alternating, sort stage ms): 4K 0.489 -> 0.473; 1080p 0.324 -> 0.329 (C2L, C5 likewise: the two event
        // operations and the company of the gather cost more than seven microseconds hidden) -- so only frames of 16 384 tiles and more take the side stream.
        const bool order_wanted = !atomic_bin && tile_order_used(f);
#ifdef STP_ORDER_MAIN   // (A/B builds: the order kernel on the caller's stream, in front of the gather)
        const bool order_on_side = false;
#else
        const bool order_on_side = order_wanted && side != nullptr && gather_order_mode() == 0 && f.gx * (f.ty1 - f.ty0) >= 16384;
#endif
        if (order_on_side) {
            STP_TRY(hipEventRecord(mb.ev, st), "record event behind the ranges");
            STP_TRY(hipStreamWaitEvent(side->stream, mb.ev, 0), "side stream wait (ranges)");
            STP_TRY(launch_tile_order(f, img, side->stream), "tile order");
        } else if (order_wanted) STP_TRY(launch_tile_order(f, img, st), "tile order");
        STP_TRY(colours.join(), "join colour stream"); // (the entry gather -- or, in GLOBAL mode, the render kernel -- reads the colours)
        SideJoin ordering{mb.done, st, false};
        if (order_on_side) {
            if (hipError_t e = hipEventRecord(mb.done, side->stream); e != hipSuccess) {
                (void)hipStreamSynchronize(side->stream);
                return fail_hip(e, "record tile-order event");
            }
            ordering.pending = true;
        }
        if (tile_local_sort) STP_TRY(launch_tile_sort_gather(f, g, b, img, L, atomic_bin, st), "tile sort + entry gather");
        else STP_TRY(launch_gather_entries(f, g, b, L, st), "entry gather");
        STP_DEBUG_SYNC("entry gather");
        STP_TRY(ordering.join(), "join tile order");
        g_timer.mark(3, st);
        std::string err;
        hipError_t e;
        if (split.armed && split.row > f.ty0 && split.row < f.ty1 && f.s.debug_visualization == 0) {
            // two launches, tile rows [ty0, row) and [row, ty1), the caller's event between them: a tile-row shard sends the first half of its
            // strip while the second half renders (include/stp_raster.h: stp_set_forward_split).  Same kernels, same per-tile work, same pixels.
            FrameParams f1 = f, f2 = f;
            f1.ty1 = split.row; f2.ty0 = split.row;
            f1.split_launch = f2.split_launch = 1;
            e = launch_render_forward(f1, g, b, img, out_color, st, &err);
            if (e == hipSuccess) { e = hipEventRecord(split.event, st); split_guard.done = e == hipSuccess; }
            if (e == hipSuccess) e = launch_render_forward(f2, g, b, img, out_color, st, &err);
        } else e = launch_render_forward(f, g, b, img, out_color, st, &err);
        if (e != hipSuccess) {
            if (!err.empty()) return fail(STP_ERR_QUEUE_SIZE, err);
            return fail_hip(e, "render launch");
        }
        STP_DEBUG_SYNC("render");
        STP_TRY(launch_render_debug_finish(f, img, out_color, st), "debug visualisation");
        g_timer.mark(4, st);
        return 0;
    };
    auto read_mailbox = [&](int* R_out, bool* wild_out) -> int {
        if (int rc = wait_mailbox()) return rc;
        const uint32_t host_status[2] = {mb.host[0], mb.host[1]};
        const uint32_t word = mb.host[3]; // (tag << 16 | blends per pixel): the report of the last recording forward(s) that used this slot's word
        if (const uint32_t reported = (word >> 16) == log_tag ? (word & 0xFFFFu) : 0u) { // of THIS kind: never less than 31/32 of what was known
            const uint32_t known = gslot.key.load(std::memory_order_acquire) == gkey ? gslot.log_need.load(std::memory_order_relaxed) : 0u;
            const uint32_t keep = known - known / 32;
            gslot.log_need.store(reported > keep ? reported : keep, std::memory_order_relaxed);
        } else if (gslot.key.load(std::memory_order_acquire) != gkey) gslot.log_need.store(0u, std::memory_order_relaxed); // (the slot changes hands)
        if (host_status[1] & 1u) return fail(STP_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
        *wild_out = (host_status[1] & 2u) != 0;
        *R_out = (int)host_status[0];
        // next frame's guess: this frame's count, but never less than 31/32 of the last guess -- with a moving camera the count jumps from
        // frame to frame, and a guess that follows every dip overflows at the next peak (a redone frame costs far more than padding)
        const uint32_t prev = gslot.key.load(std::memory_order_acquire) == gkey ? gslot.R.load(std::memory_order_relaxed) : 0u;
        const uint32_t keep = run_ahead ? prev - prev / 32 : 0u;
        gslot.R.store((uint32_t)*R_out > keep ? (uint32_t)*R_out : keep, std::memory_order_relaxed);
        gslot.key.store(gkey, std::memory_order_release);
        return 0;
    };

    int R = 0;
    bool wild = false;
    GeometryState g_dup = g; // what duplicate_kernel sees (a redone frame finds the finished scan in point_offsets: no second level)
    if (ahead) {
        f.wild_cov = 0; // (every sane frame; the status word says otherwise afterwards)
        g_timer.mark(1, st);
        const BinningState b = carve_binning(bin_ptr, (size_t)cap, nullptr);
        if (int rc = binning_and_render(g_dup, b, (int)cap, cap)) return rc;
        if (int rc = read_mailbox(&R, &wild)) return rc;
        if ((uint32_t)R <= cap && !wild) {
            remember_layout(bin_ptr, cap, R);
            remember_log_depth(img_ptr, (uint32_t)log_depth, R);
            return R;
        }
        // the frame did not fit its guess (or needs the checked reciprocal): once more from duplicate_kernel on, exact this time
        g_dup.block_prefix = nullptr; g_dup.block_sums = nullptr;
        STP_TRY(launch_frame_init(g, img, f.gx * f.ty0, f.gx * (f.ty1 - f.ty0), with_log, atomic_bin, st), "frame init launch"); // ranges and tile flags of the discarded pass
    } else {
        if (int rc = read_mailbox(&R, &wild)) return rc;
        STP_DEBUG_SYNC("SH colour");
        g_timer.mark(1, st);
    }
    f.wild_cov = wild ? 1 : 0;
    size_t bin_bytes = 0;
    carve_binning(nullptr, (size_t)R, &bin_bytes);
    if (bin_bytes > bin_have) {
        bin_ptr = (char*)binning_alloc(binning_user, bin_bytes);
        if (!bin_ptr) return fail(STP_ERR_ALLOC, "binning allocator returned NULL");
    }
    const BinningState b = carve_binning(bin_ptr, (size_t)R, nullptr);
    if (int rc = binning_and_render(g_dup, b, R, 0xFFFFFFFFu)) return rc;
    if (bin_ptr) remember_layout(bin_ptr, (uint32_t)R, R);
    remember_log_depth(img_ptr, (uint32_t)log_depth, R);
    return R;
}

int stp_backward_phases(int phases, int P, int D, int M, int R, const float* background, int width, int height, const StpSettings* settings,
                 const float* means3D, const float* shs, const float* opacities, const float* colors_precomp, const float* scales,
                 float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* inv_viewprojmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                 const float* pixel_colors, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                 const float* dL_dpix, float* dL_dmean2D, float* grad_records, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!settings) return fail(STP_ERR_INVALID_ARGUMENT, "null settings");
    if (P == 0) return 0; // reference rasterize_points.cu:191
    if (int rc = check_settings(*settings, true)) return rc;
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail(STP_ERR_INVALID_ARGUMENT, "null scratch buffer");
    if (!grad_records) return fail(STP_ERR_INVALID_ARGUMENT, "null gradient record buffer");
    if ((phases & 1) && (!dL_dpix || !pixel_colors)) return fail(STP_ERR_INVALID_ARGUMENT, "null image gradient");
    if ((phases & 2) && (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot))
        return fail(STP_ERR_INVALID_ARGUMENT, "null gradient buffer");

    FrameParams f;
    fill_frame(f, P, D, M, background, width, height, *settings, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
               rotations, cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, cam_pos, tan_fovx, tan_fovy, 0);
    const bool with_inv = requires_depth_along_ray(*settings);
    GeometryState g = carve_geometry(geom_buffer, (size_t)P, with_inv, nullptr);
    // what the buffers were carved with travels with them (cache of this process, else the buffers' own headers): a buffer that carries none is refused
    uint32_t bin_cap = 0, log_depth = 0;
    if (R > 0) { if (int rc = layout_of(binning_buffer, (uint32_t)R, &bin_cap, (hipStream_t)stream, true)) return rc; } // (a run-ahead forward carved it for its capacity)
    if (uses_blend_log(*settings)) { if (int rc = log_depth_of(image_buffer, (int64_t)R, &log_depth, (hipStream_t)stream, true)) return rc; }
    BinningState b = carve_binning(binning_buffer, (size_t)bin_cap, nullptr);
    ImageState img = carve_image(image_buffer, width, height, f.ty0, f.ty1, (int)log_depth, nullptr);
    if (!radii) radii = g.internal_radii;

    BackwardParams bw;
    bw.pixel_colors = pixel_colors; bw.dL_dpix = dL_dpix; bw.dL_dmean2D = dL_dmean2D; bw.grad_rec = grad_records;
    bw.grad_stride = (phases & 4) ? STP_GRAD_RECORD_USED : STP_GRAD_RECORD_FLOATS;
    bw.clear_rec = (phases & 8) ? 1 : 0;
    bw.chunks = (phases >> 8) & 0xFF; bw.chunk = (phases >> 16) & 0xFF; // per-Gaussian half by id range (see stp_raster.h)
    if (bw.chunks > 1 && (bw.chunk >= bw.chunks || (phases & 1))) return fail(STP_ERR_INVALID_ARGUMENT, "chunked per-Gaussian half: chunk index out of range, or combined with the render half");
    bw.dL_dopacity = dL_dopacity; bw.dL_dcolor = dL_dcolor; bw.dL_dmean3D = dL_dmean3D; bw.dL_dcov3D = dL_dcov3D; bw.dL_dsh = dL_dsh;
    bw.dL_dscale = dL_dscale; bw.dL_drot = dL_drot;

    if (phases & 1) {
        g_timer.begin_backward();
        g_timer.mark(5, st);
        std::string err;
        hipError_t e = launch_render_backward(f, g, b, img, bw, st, &err);
        if (e != hipSuccess) {
            if (!err.empty()) return fail(settings->sort_mode == MODE_FULL ? STP_ERR_NO_BACKWARD : STP_ERR_QUEUE_SIZE, err);
            return fail_hip(e, "backward render launch");
        }
        STP_DEBUG_SYNC("backward render");
        g_timer.mark(6, st);
    }
    if (phases & 2) {
        if (!(phases & 1)) g_timer.mark(6, st);
        STP_TRY(launch_preprocess_backward(f, g, radii, bw, st), "backward preprocess launch");
        STP_DEBUG_SYNC("backward preprocess");
        g_timer.mark(7, st);
    }
    return 0;
}

int stp_backward(int P, int D, int M, int R, const float* background, int width, int height, const StpSettings* settings,
                 const float* means3D, const float* shs, const float* opacities, const float* colors_precomp, const float* scales,
                 float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* inv_viewprojmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                 const float* pixel_colors, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                 const float* dL_dpix, float* dL_dmean2D, float* grad_records, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    return stp_backward_phases(3, P, D, M, R, background, width, height, settings, means3D, shs, opacities, colors_precomp, scales,
                               scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, cam_pos, tan_fovx,
                               tan_fovy, pixel_colors, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, grad_records,
                               dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream);
}

int stp_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream)
{
    (void)projmatrix; // reference checkFrustum only applies the view-space near test (rasterizer_impl.cu:113-128)
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(STP_ERR_INVALID_ARGUMENT, "null input");
    STP_TRY(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream), "mark_visible launch");
    return 0;
}

} // extern "C"
