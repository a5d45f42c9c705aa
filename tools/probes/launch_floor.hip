// Diagnostic (not part of the library): spacing of back-to-back kernels in a stream and in a hipGraph, for empty
// kernels of several grid sizes, with and without a tiny global store.  hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty(int* p) {
    if (p != nullptr && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void k_store(int* p) {
    if (threadIdx.x == 0) p[blockIdx.x] = blockIdx.x;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    int* buf;
    CK(hipMalloc(&buf, 1 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int n = 160;
    for (int store = 0; store < 2; ++store)
        for (int grid : {1, 256, 768, 1536, 4096}) {
            for (int threads : {256}) {
                auto launch = [&](hipStream_t s) {
                    if (store) hipLaunchKernelGGL(k_store, dim3(grid), dim3(threads), 0, s, buf);
                    else hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, s, buf);
                };
                for (int i = 0; i < 50; ++i) launch(st);
                CK(hipStreamSynchronize(st));
                auto t0 = std::chrono::steady_clock::now();
                for (int rep = 0; rep < 10; ++rep)
                    for (int i = 0; i < n; ++i) launch(st);
                CK(hipStreamSynchronize(st));
                double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * n);
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                for (int i = 0; i < n; ++i) launch(st);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                t0 = std::chrono::steady_clock::now();
                for (int rep = 0; rep < 20; ++rep) CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (20.0 * n);
                printf("%s grid %5d x %3d threads: stream %6.2f us/launch, graph %6.2f us/launch\n", store ? "store" : "empty", grid, threads, us_stream, us_graph);
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
        }
    return 0;
}
