// Diagnostic: per-launch time in a hipGraph of a kernel that only READS `mb` megabytes (768 x 256 threads, 16 B per
// load, several loads in flight), normal vs non-temporal loads; shows the fixed cost next to the streaming time.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, long n16, unsigned* sink) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a, b, c, d;
        if (NT) {
            a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride);
            c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride);
        } else {
            a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
        }
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345u && blockIdx.x == 0x7fffffff) *sink = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const long total = 3L << 30;  // rotate through 3 GB so that nothing is cache resident
    char* buf;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int n = 100;
    for (int nt = 0; nt < 2; ++nt)
        for (double mb : {0.0, 1.0, 8.0, 25.0, 48.0}) {
            const long bytes = (long)(mb * 1e6) / 16 * 16;
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            long off = 0;
            for (int i = 0; i < n; ++i) {
                if (off + bytes > total) off = 0;
                if (nt) hipLaunchKernelGGL(k_read<true>, dim3(768), dim3(256), 0, st, (const u32x4*)(buf + off), bytes / 16, sink);
                else hipLaunchKernelGGL(k_read<false>, dim3(768), dim3(256), 0, st, (const u32x4*)(buf + off), bytes / 16, sink);
                off += (bytes + 4095) / 4096 * 4096;
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 10; ++rep) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * n);
            printf("%s read %5.1f MB: %6.2f us/launch  (%.0f GB/s; minus 1.6 us floor: %.0f GB/s)\n", nt ? "nt    " : "normal", mb, us,
                   bytes / us / 1e3, us > 1.7 ? bytes / (us - 1.6) / 1e3 : 0.0);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    return 0;
}
