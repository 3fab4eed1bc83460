// What a hipGraph could save the small-frame forward ("hipGraph the host-independent part ... or show"): host time to enqueue a chain of
// N small kernels (a C1 forward enqueues 14 behind its hand-over, a backward 4) -- as N launches, as ONE graph launch with unchanged arguments, and as a
// graph launch after N hipGraphExecKernelNodeSetParams (what a frame needs: the scratch pool, torch's output tensors and the adaptive sizes move the
// pointer arguments from call to call).  Host microseconds per chain, median of 2000; the GPU side is the same chain either way.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { float* p; int n; float s; const float* q; int pad[24]; }; // (a kernarg block of the size the library's kernels take: ~128 bytes)
__global__ void tiny(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0 && a.n == 12345) a.p[0] += a.s; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    float* d; hipMalloc(&d, 4096);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int N : {4, 14, 25}) {
        Args a{d, 1, 1.0f, d, {}};
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::vector<double> t_launch, t_graph, t_update;
        for (int it = 0; it < 2000; it++) {
            const double t0 = now();
            for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, a);
            t_launch.push_back(now() - t0);
            if (it % 64 == 63) hipStreamSynchronize(st);
        }
        hipStreamSynchronize(st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, a);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
        std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
        for (int it = 0; it < 2000; it++) {
            const double t0 = now();
            hipGraphLaunch(ge, st);
            t_graph.push_back(now() - t0);
            if (it % 64 == 63) hipStreamSynchronize(st);
        }
        hipStreamSynchronize(st);
        void* kargs[1] = {&a};
        hipKernelNodeParams kp{}; kp.func = (void*)tiny; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = kargs; kp.extra = nullptr;
        for (int it = 0; it < 2000; it++) {
            a.n = it & 7;
            const double t0 = now();
            for (size_t k = 0; k < nn; k++) hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
            hipGraphLaunch(ge, st);
            t_update.push_back(now() - t0);
            if (it % 64 == 63) hipStreamSynchronize(st);
        }
        hipStreamSynchronize(st);
        printf("N = %2d kernels: %6.1f us as launches, %6.1f us as one graph launch, %6.1f us as %zu node updates + graph launch\n", N, med(t_launch), med(t_graph), med(t_update), nn);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
