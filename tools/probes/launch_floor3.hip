// Diagnostic: does the per-launch floor in a hipGraph depend on the kernel's footprint (VGPRs, LDS, kernarg size)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Big { int v[120]; };

__global__ void __launch_bounds__(256) k_small(int* p) {
    if (p != nullptr && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void __launch_bounds__(256) k_vgpr(int* p) {  // ~120 live VGPRs, never executed past the guard
    int r[112];
    if (blockIdx.x == 0x7fffffff) {
#pragma unroll
        for (int i = 0; i < 112; ++i) r[i] = p[i * 64 + threadIdx.x];
        int s = 0;
#pragma unroll
        for (int i = 0; i < 112; ++i) s += r[i] * (i + 1);
        p[threadIdx.x] = s;
    }
}
__global__ void __launch_bounds__(256) k_lds(int* p) {
    extern __shared__ int sm[];
    if (blockIdx.x == 0x7fffffff) { sm[threadIdx.x] = 1; __syncthreads(); p[threadIdx.x] = sm[255 - threadIdx.x]; }
}
__global__ void __launch_bounds__(256) k_arg(int* p, Big b) {
    if (blockIdx.x == 0x7fffffff) p[threadIdx.x] = b.v[threadIdx.x % 120];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename F>
static int run(const char* name, hipStream_t st, F launch) {
    const int n = 160;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 20; ++rep) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (20.0 * n);
    printf("%-28s %6.2f us/launch\n", name, us);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    int* buf;
    CK(hipMalloc(&buf, 1 << 22));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Big b{};
    run("small, 768 x 256", st, [&] { hipLaunchKernelGGL(k_small, dim3(768), dim3(256), 0, st, buf); });
    run("120 VGPRs, 768 x 256", st, [&] { hipLaunchKernelGGL(k_vgpr, dim3(768), dim3(256), 0, st, buf); });
    run("8 KB dyn LDS, 768 x 256", st, [&] { hipLaunchKernelGGL(k_lds, dim3(768), dim3(256), 8192, st, buf); });
    run("480 B kernarg, 768 x 256", st, [&] { hipLaunchKernelGGL(k_arg, dim3(768), dim3(256), 0, st, buf, b); });
    run("small, 32 x 1024", st, [&] { hipLaunchKernelGGL(k_small, dim3(32), dim3(1024), 0, st, buf); });
    run("small, 2752 x 256", st, [&] { hipLaunchKernelGGL(k_small, dim3(2752), dim3(256), 0, st, buf); });
    return 0;
}
