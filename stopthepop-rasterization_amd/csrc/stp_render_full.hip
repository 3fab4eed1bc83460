// stp_render_full.hip -- PPX_FULL: (near-)full per-pixel depth sort, forward only.
//
// Replaces renderSortedFullCUDA<3,false> (reference stopthepop/resorted_render.cuh:474-675): the
// quality-evaluation mode.  Semantics: one pixel at a time, the workgroup keeps a window of 1024
// candidates keyed by depth along THAT pixel's ray (all list entries are candidates -- no alpha
// pre-test and, unlike the other sorted modes, negative depths are not rejected); each round the
// 256 nearest are blended in order and 256 new list entries join.  It is slow by design (the
// reference's own comment), used as ground truth, and has no backward (backward.cu:733-736).
//
// Our machinery: the window lives in LDS as 1024 unique 64-bit keys (order-preserving depth bits
// << 32 | list position), re-sorted per round by an in-LDS bitonic network run by 256 threads;
// the 256 per-entry alpha evaluations of a round run in parallel, only the transmittance
// recurrence is serial.  Equal depths order by list position.
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {

namespace {

__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ float ordered_to_float(uint32_t o)
{
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ void __launch_bounds__(256) render_full_fwd_kernel(const RenderArgs a)
{
    __shared__ uint64_t s_key[1024];
    __shared__ float s_alpha[256];  // < 0: skipped entry, otherwise alpha
    __shared__ float s_col[3][256];
    __shared__ int s_stop;          // index of the first filler entry in the front 256, or 256

    const int rows = a.ty1 - a.ty0;
    const int t = (int)blockIdx.x; // no XCD remap needed: one tile runs for a very long time
    (void)rows;
    const int tile_x = t % a.gx, tile_y = a.ty0 + t / a.gx;
    const uint2 range = a.ranges[tile_y * a.gx + tile_x];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + 255) / 256;
    const int tid = (int)threadIdx.x;
    const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
    const size_t N = (size_t)a.W * a.H;
    constexpr uint64_t FILLER = ~0ull;

    for (int lx = 0; lx < TILE; lx++)
        for (int ly = 0; ly < TILE; ly++) {
            const int px = tile_x * TILE + lx, py = tile_y * TILE + ly;
            if (!(px < a.W && py < a.H)) continue; // uniform for the workgroup
            const float3 dir = view_ray(a.inv_vp, cam, (float)px, (float)py, a.W, a.H);

            auto make_key = [&](int p) -> uint64_t {
                if (p >= total) return FILLER;
                const int id = (int)a.point_list[range.x + p];
                const float d = depth_along_ray(f4_xyz(a.cov3D_inv[3 * (size_t)id]), f4_xyz(a.cov3D_inv[3 * (size_t)id + 1]),
                                                f4_xyz(a.cov3D_inv[3 * (size_t)id + 2]), dir);
                return ((uint64_t)float_to_ordered(d) << 32) | (uint32_t)p;
            };
            __syncthreads();
            for (int i = 0; i < 3; i++) s_key[256 * (i + 1) + tid] = make_key(i * 256 + tid); // slots 256..1023
            float T = 1.0f, C[3] = {0, 0, 0}, depth_acc = 0.0f;
            uint32_t contributor = 0, last_contributor = 0;
            bool done = false;
            int todo = total;
            for (int r = 0; r < rounds; r++, todo -= 256) {
                s_key[tid] = make_key((r + 3) * 256 + tid); // front slots were consumed last round
                __syncthreads();
                // bitonic sort of 1024 unique keys, ascending
                for (int k = 2; k <= 1024; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int c = tid; c < 512; c += 256) {
                            const int lo = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                            const int hi = lo | j;
                            const bool up = (lo & k) == 0;
                            const uint64_t x = s_key[lo], y = s_key[hi];
                            if ((x > y) == up) { s_key[lo] = y; s_key[hi] = x; }
                        }
                        __syncthreads();
                    }
                // evaluate the 256 nearest in parallel
                {
                    const uint64_t key = s_key[tid];
                    float alpha = -1.0f;
                    if (key != FILLER) {
                        const int id = (int)a.point_list[range.x + (uint32_t)key];
                        const float2 xy = a.means2D[id];
                        const float4 co = a.conic_opacity[id];
                        const float dx = xy.x - (float)px, dy = xy.y - (float)py;
                        const float power = opacity_factor(dx, dy, co);
                        if (!(power < 0.0f)) {
                            const float al = fminf(0.99f, co.w * exp_blend(-power));
                            if (!(al < ALPHA_THRESHOLD)) alpha = al;
                        }
                        s_col[0][tid] = a.features[3 * (size_t)id];
                        s_col[1][tid] = a.features[3 * (size_t)id + 1];
                        s_col[2][tid] = a.features[3 * (size_t)id + 2];
                    }
                    s_alpha[tid] = alpha;
                    if (tid == 0) s_stop = 256;
                    __syncthreads();
                    if (key == FILLER && (tid == 0 || s_key[tid - 1] != FILLER)) s_stop = tid;
                    __syncthreads();
                }
                if (tid == 0 && !done) { // the serial transmittance recurrence
                    const int n = min(min(256, todo), s_stop);
                    for (int idx = 0; !done && idx < n; idx++) {
                        contributor++;
                        const float alpha = s_alpha[idx];
                        if (alpha < 0.0f) continue;
                        const float test_T = T * (1 - alpha);
                        if (test_T < T_THRESHOLD) { done = true; continue; }
                        for (int ch = 0; ch < 3; ch++) C[ch] += s_col[ch][idx] * alpha * T;
                        if (a.debug_depth) depth_acc += ordered_to_float((uint32_t)(s_key[idx] >> 32)) * alpha * T; // reference resorted_render.cuh:647
                        T = test_T;
                        last_contributor = contributor;
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                const size_t pid = (size_t)a.W * py + px;
                a.final_T[pid] = T;
                a.n_contrib[pid] = last_contributor;
                if (a.debug_depth) { a.out_color[pid] = depth_acc; a.out_color[N + pid] = T; }
                else for (int ch = 0; ch < 3; ch++) a.out_color[ch * N + pid] = C[ch] + T * a.bg[ch];
            }
        }
}

} // namespace

struct RenderArgs;
hipError_t launch_full_fwd(const FrameParams& f, const RenderArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(render_full_fwd_kernel, dim3(f.gx * (f.ty1 - f.ty0)), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace stp
