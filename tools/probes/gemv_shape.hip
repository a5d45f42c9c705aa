// Diagnostic: which structural feature of the int4 GEMV costs time relative to a pure streaming read?
// All variants read the same 26.7 MB "matrix" (12288 rows x 2064 B + 64 B of metadata per row slab) with the GEMV's
// access pattern (wave = one 1-KiB slab of a row per load, 8 rows per wave, 4 waves per workgroup, 768 workgroups).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROW16 = 129;  // 16-byte units per row (2 slabs x 64 + 1 metadata unit)
constexpr int N = 12288;

// MODE bit 0: rolling window of 3 (else all 8 at once); bit 1: metadata loads; bit 2: x loads first + use;
// bit 3: LDS reduce + barrier + epilogue store; bit 4: non-temporal loads
template <int MODE>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ W, const u32x4* __restrict__ x, unsigned short* out) {
    __shared__ float red[4][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slab = wave >> 1, j = wave & 1;
    const int r0 = blockIdx.x * 16 + j * 8;
    unsigned acc = 0;
    u32x4 xv = {0, 0, 0, 0};
    if (MODE & 4) xv = x[slab * 64 + lane];
    u32x4 w[8];
    unsigned mt[8];
    auto issue = [&](int u) {
        const u32x4* rec = W + (long)(r0 + u) * ROW16;
        if (MODE & 16) w[u] = __builtin_nontemporal_load(rec + slab * 64 + lane); else w[u] = rec[slab * 64 + lane];
        if (MODE & 2) mt[u] = reinterpret_cast<const unsigned*>(rec + 128)[slab * 2 + (lane >> 5)]; else mt[u] = 0;
    };
    constexpr int PRIME = (MODE & 1) ? 3 : 8;
#pragma unroll
    for (int u = 0; u < PRIME; ++u) issue(u);
    asm volatile("" ::: "memory");
    if (MODE & 4) acc ^= xv.x ^ xv.y ^ xv.z ^ xv.w;
    float part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u + PRIME < 8) issue(u + PRIME);
        asm volatile("" ::: "memory");
        unsigned v = w[u].x ^ w[u].y ^ w[u].z ^ w[u].w ^ mt[u] ^ acc;
        part[u] = __uint_as_float(v & 0x3fffffffu);
        asm volatile("" : "+v"(part[u])::"memory");
    }
    if (MODE & 8) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += part[u];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane < 8) red[wave][lane] = s + part[lane & 7];
        __syncthreads();
        if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = (unsigned short)(red[threadIdx.x >> 3][threadIdx.x & 7] + red[2 + (threadIdx.x >> 3)][threadIdx.x & 7]);
    } else {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += part[u];
        if (s == 1.2345f) out[threadIdx.x] = 1;
    }
}

// SL = 2 shape: a wave owns BOTH slabs of its 8 rows (two 1-KiB loads per row), 4 waves = 32 rows per workgroup,
// no cross-wave reduction: wave sum -> epilogue store by the wave itself, no LDS, no barrier
template <int PRIMEV>
__global__ void __launch_bounds__(256) k2(const u32x4* __restrict__ W, const u32x4* __restrict__ x, unsigned short* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 32 + wave * 8;
    unsigned acc = 0;
    const u32x4 xa = x[lane], xb = x[64 + lane];
    u32x4 w[8][2];
    unsigned mt[8][2];
    auto issue = [&](int u) {
        const u32x4* rec = W + (long)(r0 + u) * ROW16;
        w[u][0] = __builtin_nontemporal_load(rec + lane);
        w[u][1] = __builtin_nontemporal_load(rec + 64 + lane);
        mt[u][0] = reinterpret_cast<const unsigned*>(rec + 128)[lane >> 5];
        mt[u][1] = reinterpret_cast<const unsigned*>(rec + 128)[2 + (lane >> 5)];
    };
#pragma unroll
    for (int u = 0; u < PRIMEV; ++u) issue(u);
    asm volatile("" ::: "memory");
    acc ^= xa.x ^ xa.y ^ xb.z ^ xb.w;
    float part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u + PRIMEV < 8) issue(u + PRIMEV);
        asm volatile("" ::: "memory");
        unsigned v = w[u][0].x ^ w[u][0].y ^ w[u][0].z ^ w[u][0].w ^ w[u][1].x ^ w[u][1].y ^ w[u][1].z ^ w[u][1].w ^ mt[u][0] ^ mt[u][1] ^ acc;
        part[u] = __uint_as_float(v & 0x3fffffffu);
        asm volatile("" : "+v"(part[u])::"memory");
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += part[u];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((lane & 7) == 0) out[r0 + (lane >> 3)] = (unsigned short)(s + part[lane >> 3 & 7]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
int run(const char* name, hipStream_t st, char* buf, long total, u32x4* x, unsigned short* out) {
    const long bytes = (long)N * ROW16 * 16;
    const int n = 100;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    long off = 0;
    for (int i = 0; i < n; ++i) {
        if (off + bytes > total) off = 0;
        hipLaunchKernelGGL(k<MODE>, dim3(N / 16), dim3(256), 0, st, (const u32x4*)(buf + off), (const u32x4*)x, out);
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
    printf("%-58s %6.2f us/launch (%.0f GB/s)\n", name, us, bytes / us / 1e3);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

template <int PRIMEV>
int run2(const char* name, hipStream_t st, char* buf, long total, u32x4* x, unsigned short* out) {
    const long bytes = (long)N * ROW16 * 16;
    const int n = 100;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    long off = 0;
    for (int i = 0; i < n; ++i) {
        if (off + bytes > total) off = 0;
        hipLaunchKernelGGL(k2<PRIMEV>, dim3(N / 32), dim3(256), 0, st, (const u32x4*)(buf + off), (const u32x4*)x, out);
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
    printf("%-58s %6.2f us/launch (%.0f GB/s)\n", name, us, bytes / us / 1e3);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    const long total = 3L << 30;
    char* buf;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    u32x4* x;
    CK(hipMalloc(&x, 1 << 16));
    unsigned short* out;
    CK(hipMalloc(&out, 1 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    run<0>("burst of 8 rows", st, buf, total, x, out);
    run<1>("rolling window 3", st, buf, total, x, out);
    run<16>("burst, non-temporal", st, buf, total, x, out);
    run<17>("rolling, non-temporal", st, buf, total, x, out);
    run<19>("rolling, nt, + metadata loads", st, buf, total, x, out);
    run<23>("rolling, nt, meta, + x load first", st, buf, total, x, out);
    run<31>("rolling, nt, meta, x, + LDS reduce/barrier/epilogue", st, buf, total, x, out);
    run<30>("burst, nt, meta, x, LDS reduce/barrier/epilogue", st, buf, total, x, out);
    run2<2>("wave owns both slabs, no barrier, window 2", st, buf, total, x, out);
    run2<3>("wave owns both slabs, no barrier, window 3", st, buf, total, x, out);
    run2<4>("wave owns both slabs, no barrier, window 4", st, buf, total, x, out);
    return 0;
}
