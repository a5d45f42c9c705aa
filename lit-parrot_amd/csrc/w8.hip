// LLM.int8 Linear (quantize/bnb.py:18-60).  The arithmetic lives in the third-party bitsandbytes wheel
// (>= 0.40.0, not vendored in the reference); this file restates the published algorithm as configured
// there (has_fp16_weights=False, threshold=6.0):
//   weights   : CB = rint(127 * fp16(W) / absmax_row) int8, SCB = absmax_row            (double_quant, bnb.py:55)
//   per token : A = fp16(x); columns with |A| >= threshold are outliers: they are zero in the int8 copy and
//               excluded from the row absmax; CA = rint(127 * A / absmax_row), SCA = absmax_row
//   product   : C32 = CA . CB^T (int32);  out16 = fp16(C32 * (1/127^2) * SCA * SCB + bias)      (mm_dequant)
//   outliers  : out16 = fp16(out16 + fp16(sum_k A[k] * fp16(CB[o,k] * SCB[o] / 127)))          (mixed decomposition)
//   result    : cast back to the input dtype (bf16), then the layer's epilogue.
// Rows of a multi-row call are treated independently (== M separate single-token calls).
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include "parrot_common.h"

namespace parrot {

__device__ __forceinline__ float rhalf(float v) { return __half2float(__float2half(v)); }
constexpr float kMmDequant = 6.200012e-05f;  // 1 / (127 * 127)

__device__ __forceinline__ float block_max_256(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// the checkpoint value as the reference quantises it: `weight.contiguous().half()` (quantize/bnb.py:54) whatever the dtype
__device__ __forceinline__ float w8_as_half(bf16_t v) { return rhalf(bf2f(v)); }
__device__ __forceinline__ float w8_as_half(__half v) { return __half2float(v); }
__device__ __forceinline__ float w8_as_half(float v) { return rhalf(v); }

// one workgroup (256 threads) per weight row
template <typename T>
__global__ void __launch_bounds__(256)
w8_quantize_rows_kernel(const T* __restrict__ W, int K, int8_t* __restrict__ CB, float* __restrict__ SCB) {
    __shared__ float sh[4];
    const T* w = W + (int64_t)blockIdx.x * K;
    float mx = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(w8_as_half(w[k])));
    mx = block_max_256(mx, sh);
    const float inv = mx > 0.f ? __fdiv_rn(127.0f, mx) : 0.f;  // correctly rounded, like the host oracle
    for (int k = threadIdx.x; k < K; k += 256)
        CB[(int64_t)blockIdx.x * K + k] = (int8_t)rintf(__fmul_rn(w8_as_half(w[k]), inv));
    if (threadIdx.x == 0) SCB[blockIdx.x] = mx;
}

// One workgroup (1024 threads) per token row, one pass: every thread keeps up to kPrepV 16-byte chunks of the row in
// registers (K <= 1024 * 8 * kPrepV), so the row is read once for: optional fused norm -> fp16 cast -> outlier split ->
// row absmax -> int8.  The outlier columns are also written as a compact, deterministic index list (thread-major order)
// so that the GEMV's mixed-precision part touches only those columns.
constexpr int kPrepThreads = 1024;
constexpr int kPrepV = 4;

__device__ __forceinline__ float block_max_1024(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0];
#pragma unroll
    for (int i = 1; i < kPrepThreads / 64; ++i) t = fmaxf(t, sh[i]);
    return t;
}

__global__ void __launch_bounds__(kPrepThreads)
w8_prep_act_kernel(const bf16_t* __restrict__ x, int ldx, int K, float threshold, int8_t* __restrict__ xq,
                   float* __restrict__ xout, float* __restrict__ sca, int32_t* __restrict__ nout,
                   int32_t* __restrict__ oidx, NormArgs na) {
    __shared__ float sh[16];
    __shared__ float stat[16];
    __shared__ int scan[kPrepThreads / 64];
    const int m = blockIdx.x;
    const int chunks = K >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx);
    float a[kPrepV][8];
    // ---- load (+ fused norm) -> fp16-rounded values in registers
    uint32_t raw[kPrepV][4];
#pragma unroll
    for (int i = 0; i < kPrepV; ++i) {
        const int c = threadIdx.x + i * kPrepThreads;
        const uint4 v = c < chunks ? xr[c] : make_uint4(0, 0, 0, 0);
        raw[i][0] = v.x; raw[i][1] = v.y; raw[i][2] = v.z; raw[i][3] = v.w;
    }
    float mean = 0.f, r = 1.f;
    if (na.kind != 0) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < kPrepV; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) s1 += norm_stat1(raw[i][q], na.kind);
        s1 = block_sum_waves(s1, stat, kPrepThreads / 64);
        if (na.kind == 2) {
            mean = s1 / (float)K;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < kPrepV; ++i)
                if (threadIdx.x + i * kPrepThreads < chunks)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s2 += norm_stat2(raw[i][q], mean);
            r = norm_scale(na, block_sum_waves(s2, stat, kPrepThreads / 64));
        } else {
            r = norm_scale(na, s1);
        }
    }
    float mx = 0.f;
    int local = 0;
#pragma unroll
    for (int i = 0; i < kPrepV; ++i) {
        const int c = threadIdx.x + i * kPrepThreads;
        const bool ok = c < chunks;
        uint32_t nrm[4] = {raw[i][0], raw[i][1], raw[i][2], raw[i][3]};
        if (na.kind != 0 && ok) {
            const uint4 wv = reinterpret_cast<const uint4*>(na.weight)[c];
            uint4 bv = make_uint4(0, 0, 0, 0);
            if (na.kind == 2 && na.bias != nullptr) bv = reinterpret_cast<const uint4*>(na.bias)[c];
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) nrm[q] = norm_apply(raw[i][q], ww[q], bb[q], na.kind, mean, r);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[i][2 * q] = rhalf(bflo(nrm[q]));
            a[i][2 * q + 1] = rhalf(bfhi(nrm[q]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool outlier = ok && threshold > 0.f && fabsf(a[i][e]) >= threshold;
            if (ok && !outlier) mx = fmaxf(mx, fabsf(a[i][e]));
            local += outlier ? 1 : 0;
        }
    }
    mx = block_max_1024(mx, sh);
    const float inv = mx > 0.f ? __fdiv_rn(127.0f, mx) : 0.f;  // correctly rounded, like the host oracle
    // ---- exclusive scan of the outlier counts (wave prefix + LDS over the 16 waves) -> slots in the index list
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) scan[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kPrepThreads / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += scan[w];
        total += scan[w];
    }
    int slot = base + incl - local;
#pragma unroll
    for (int i = 0; i < kPrepV; ++i) {
        const int c = threadIdx.x + i * kPrepThreads;
        if (c < chunks) {
            uint32_t q8[2] = {0, 0};
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool outlier = threshold > 0.f && fabsf(a[i][e]) >= threshold;
                const int qv = outlier ? 0 : (int)rintf(__fmul_rn(a[i][e], inv));
                q8[e >> 2] |= (uint32_t)(qv & 0xff) << (8 * (e & 3));
                o8[e] = a[i][e];  // every element: the union pass of a multi-row call needs the non-outliers of outlier columns too
                if (outlier) oidx[(int64_t)m * K + slot++] = c * 8 + e;
            }
            *reinterpret_cast<uint2*>(xq + (int64_t)m * K + (int64_t)c * 8) = make_uint2(q8[0], q8[1]);
            float4* op = reinterpret_cast<float4*>(xout + (int64_t)m * K + (int64_t)c * 8);
            op[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
            op[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
        }
    }
    if (threadIdx.x == 0) {
        sca[m] = mx;
        nout[m] = total;
    }
}

// ---- multi-row calls: LLM.int8's outliers are feature DIMENSIONS of the call (arXiv:2208.07339 §3.2; bitsandbytes MatMul8bitLt:
// idx = unique(coo_tensorA.colidx), CA[:, idx] = 0, subA = A[:, idx]): a column with an outlier in ANY row leaves the int8 product of
// EVERY row and goes through the 16-bit product for every row.  Pass 1: flag the columns (one thread per column and chunk of rows).
__global__ void __launch_bounds__(256)
w8_union_cols_kernel(const float* __restrict__ xout, int32_t* __restrict__ colflag, int M, int K, float threshold) {
    // thread = one column over a chunk of 32 rows (grid.y chunks): enough workgroups to fill the chip on a 128-row prompt
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int m0 = blockIdx.y * 32, m1 = min(M, m0 + 32);
    int any = 0;
#pragma unroll 8
    for (int m = m0; m < m1; ++m) any |= fabsf(xout[(int64_t)m * K + k]) >= threshold ? 1 : 0;  // (no short circuit: the loads overlap)
    if (any) atomicOr(colflag + k, 1);
}
// Pass 2, one workgroup per row: the ascending list of flagged columns becomes the row's outlier list (the same for every row),
// and those columns leave the row's int8 copy
__global__ void __launch_bounds__(1024)
w8_union_lists_kernel(const int32_t* __restrict__ colflag, int8_t* __restrict__ xq, int32_t* __restrict__ nout, int32_t* __restrict__ oidx, int K) {
    __shared__ int scan[16];
    const int m = blockIdx.x;
    const int per = (K + 1023) / 1024;  // consecutive columns per thread: the list comes out ascending
    const int k0 = threadIdx.x * per, k1 = min(K, k0 + per);
    int local = 0;
    for (int k = k0; k < k1; ++k) local += colflag[k];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) scan[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += scan[w];
        total += scan[w];
    }
    int slot = base + incl - local;
    for (int k = k0; k < k1; ++k)
        if (colflag[k]) {
            oidx[(int64_t)m * K + slot++] = k;
            xq[(int64_t)m * K + k] = 0;
        }
    if (threadIdx.x == 0) nout[m] = total;
}

constexpr int kW8MaxSlabs = 8;  // K <= 8 * 4096
constexpr int kW8U = 4;
constexpr int kW8MaxRows = 16;

template <bool DUAL, int J>
__global__ void __launch_bounds__(512)
w8_gemv_kernel(const uint4* __restrict__ CB, const uint4* __restrict__ CB2, const float* __restrict__ SCB,
               const float* __restrict__ SCB2, const int8_t* __restrict__ xq, const float* __restrict__ xout,
               const float* __restrict__ sca, const int32_t* __restrict__ nout, const int32_t* __restrict__ oidx,
               const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int N, int K,
               int rows_per_wg, int epi, int nslabs) {
    constexpr int NW = DUAL ? 2 : 1;
    __shared__ int red[kW8MaxSlabs][kW8MaxRows * NW];
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = K >> 4;
    const int c0 = (int)((int64_t)wave * chunks / nslabs), c1 = (int)((int64_t)(wave + 1) * chunks / nslabs);
    int cidx[J];
    int xr[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = c0 + j * 64 + lane;
        const bool ok = c < c1;
        cidx[j] = ok ? c : c1 - 1;
        uint4 v = reinterpret_cast<const uint4*>(xq + (int64_t)m * K)[cidx[j]];
        if (!ok) v = make_uint4(0, 0, 0, 0);
        xr[j][0] = (int)v.x;
        xr[j][1] = (int)v.y;
        xr[j][2] = (int)v.z;
        xr[j][3] = (int)v.w;
    }
    const int r_begin = blockIdx.x * rows_per_wg;
    const int r_end = min(N, r_begin + rows_per_wg);
    for (int r0 = r_begin; r0 < r_end; r0 += kW8U) {
        uint4 w[NW][kW8U][J];
#pragma unroll
        for (int u = 0; u < kW8U; ++u) {
            const int64_t row = min(r0 + u, N - 1);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                w[0][u][j] = CB[row * chunks + cidx[j]];
                if (DUAL) w[1][u][j] = CB2[row * chunks + cidx[j]];
            }
        }
#pragma unroll
        for (int u = 0; u < kW8U; ++u)
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                int p = 0;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    p = __builtin_amdgcn_sdot4((int)w[q][u][j].x, xr[j][0], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][j].y, xr[j][1], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][j].z, xr[j][2], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][j].w, xr[j][3], p, false);
                }
                p = wave_sum_i32_to_lane63(p);
                if (lane == 63) red[wave][(r0 - r_begin + u) * NW + q] = p;
            }
    }
    __syncthreads();
    const int nrows = r_end - r_begin;
    if ((int)threadIdx.x < nrows) {
        const int ur = threadIdx.x, col = r_begin + ur;
        const float sa = sca[m];
        const int no = nout[m];
        float res[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            int c32 = 0;
            for (int c = 0; c < nslabs; ++c) c32 += red[c][ur * NW + q];
            const float scb = q ? SCB2[col] : SCB[col];
            const float b = (bias != nullptr && q == 0) ? bf2f(bias[col]) : 0.f;
            // separately rounded fp32 products and sum (no FMA), the order mm_dequant uses
            float v = rhalf(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn((float)c32, kMmDequant), sa), scb), b));
            if (no > 0) {  // mixed-precision decomposition over the compact list of outlier columns
                const int8_t* wrow = reinterpret_cast<const int8_t*>(q ? CB2 : CB) + (int64_t)col * K;
                float o = 0.f;
                for (int t = 0; t < no; ++t) {
                    const int k = oidx[(int64_t)m * K + t];
                    o += xout[(int64_t)m * K + k] * rhalf(__fdiv_rn(__fmul_rn((float)wrow[k], scb), 127.0f));
                }
                v = rhalf(v + rhalf(o));
            }
            res[q] = v;
        }
        out[(int64_t)m * ldo + col] =
            apply_epilogue(epi, res[0], res[1], nullptr, residual ? residual + (int64_t)m * ldr : nullptr, col);
    }
}

template <int J>
static int w8_launch(const void* CB, const void* CB2, const void* SCB, const void* SCB2, const void* xq, const void* xout,
                     const void* sca, const void* nout, const void* oidx, int M, const void* bias, const void* residual, int ldr, void* out,
                     int ldo, int N, int K, int epi, int nslabs, hipStream_t st) {
    const int R = N >= 16 * 2048 ? 16 : (N >= 8 * 1024 ? 8 : 4);
    const dim3 grid((N + R - 1) / R, M), block(64 * nslabs);
    if (epi == PARROT_EPI_SWIGLU)
        return launch(K_W8_GEMV, w8_gemv_kernel<true, J>, grid, block, 0, st, (const uint4*)CB, (const uint4*)CB2,
                      (const float*)SCB, (const float*)SCB2, (const int8_t*)xq, (const float*)xout, (const float*)sca,
                      (const int32_t*)nout, (const int32_t*)oidx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo,
                      N, K, R, epi, nslabs);
    return launch(K_W8_GEMV, w8_gemv_kernel<false, J>, grid, block, 0, st, (const uint4*)CB, (const uint4*)CB2,
                  (const float*)SCB, (const float*)SCB2, (const int8_t*)xq, (const float*)xout, (const float*)sca,
                  (const int32_t*)nout, (const int32_t*)oidx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, N,
                  K, R, epi, nslabs);
}

// ------------------------------------------------------------------------------------------ fused single-token Linear
// Decode step (one row): activation quantiser + GEMV in ONE launch, on the structure of the int4 kernel (w4.hip):
// workgroup = nslabs x wps waves; the prologue (optional norm -> fp16 cast -> outlier split -> row absmax -> int8) is
// computed once per workgroup into LDS ([K] int8, [K] fp16 outlier values, a compact list of the first kW8Cap outlier
// columns in a deterministic thread-major order); weight rows go through a rolling window; resident-size grid walking
// `iters` row batches; wave priorities.  Same arithmetic as w8_prep_act_kernel + w8_gemv_kernel.
constexpr int kW8Cap = 256;

template <bool DUAL, int J, int RU, int MAXW>
__global__ void __launch_bounds__(MAXW * 64)
w8_fused_kernel(const uint4* __restrict__ CB, const uint4* __restrict__ CB2, const float* __restrict__ SCB,
                const float* __restrict__ SCB2, const bf16_t* __restrict__ x, float threshold,
                const bf16_t* __restrict__ bias, const bf16_t* residual, bf16_t* out, int N, int K, int wps, int nslabs,
                int epi, int iters, NormArgs na) {
    constexpr int NW = DUAL ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char w8_smem[];  // [K] int8 | [K] fp16
    __shared__ int red[2][MAXW][RU * NW];
    __shared__ float stat[16];
    __shared__ float shmax[16];
    __shared__ int scan[16];
    __shared__ int olist[kW8Cap];
    __shared__ int s_total;
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nslabs * wps;
    const int nthreads = nwaves * 64;
    const int slab = wave / wps, j = wave % wps;
    const int chunks16 = K >> 4;  // 16-byte units of an int8 row
    const int c0 = (int)((int64_t)slab * chunks16 / nslabs), c1 = (int)((int64_t)(slab + 1) * chunks16 / nslabs);
    const int R = wps * RU;
    int8_t* xq_l = reinterpret_cast<int8_t*>(w8_smem);
    __half* xo_l = reinterpret_cast<__half*>(w8_smem + K);

    int cidx[J];
    bool cok[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        const int c = c0 + jj * 64 + lane;
        cok[jj] = c < c1;
        cidx[jj] = cok[jj] ? c : c1 - 1;
    }
    // ---- activation row: thread t owns the 8-element chunks t, t + nthreads, ... (rounds past the row are skipped)
    const int chunks8 = K >> 3;
    constexpr int kIt = 4;  // K <= 4 * 8 * nthreads (host)
    uint4 cx[kIt], cw[kIt], cb[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        cx[it] = cw[it] = cb[it] = make_uint4(0, 0, 0, 0);
        if (it * nthreads < chunks8) {
            const int c = threadIdx.x + it * nthreads;
            const int cc = c < chunks8 ? c : chunks8 - 1;
            cx[it] = reinterpret_cast<const uint4*>(x)[cc];
            if (na.kind != 0) {
                cw[it] = reinterpret_cast<const uint4*>(na.weight)[cc];
                if (na.kind == 2 && na.bias != nullptr) cb[it] = reinterpret_cast<const uint4*>(na.bias)[cc];
            }
        }
    }
    uint4 w[NW][RU][J];
    constexpr int PRIME = RU >= 4 ? 2 : 1;
#define W8_ROW0(T) (((int)blockIdx.x + (T) * (int)gridDim.x) * R + j * RU)
#define W8_ISSUE_ROW(T, U)                                                          \
    {                                                                               \
        const int64_t row_ = min(W8_ROW0(T) + (U), N - 1);                          \
        _Pragma("unroll") for (int jj = 0; jj < J; ++jj) {                          \
            w[0][U][jj] = load_nt16(CB + row_ * chunks16 + cidx[jj]);               \
            if (DUAL) w[1][U][jj] = load_nt16(CB2 + row_ * chunks16 + cidx[jj]);    \
        }                                                                           \
    }
#pragma unroll
    for (int u = 0; u < PRIME; ++u) W8_ISSUE_ROW(0, u)
    asm volatile("" ::: "memory");

    float mean = 0.f, r = 1.f;
    if (na.kind != 0) {
        float s1 = 0.f;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            if (it * nthreads < chunks8) {
                const uint32_t dw[4] = {cx[it].x, cx[it].y, cx[it].z, cx[it].w};
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) t += norm_stat1(dw[i], na.kind);
                s1 += (threadIdx.x + it * nthreads < chunks8) ? t : 0.f;
            }
        }
        s1 = block_sum_waves(s1, stat, nwaves);
        if (na.kind == 2) {
            mean = s1 / (float)K;
            float s2 = 0.f;
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                if (threadIdx.x + it * nthreads < chunks8) {
                    const uint32_t dw[4] = {cx[it].x, cx[it].y, cx[it].z, cx[it].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) s2 += norm_stat2(dw[i], mean);
                }
            }
            r = norm_scale(na, block_sum_waves(s2, stat, nwaves));
        } else {
            r = norm_scale(na, s1);
        }
    }
    float a[kIt][8];
    float mx = 0.f;
    int local = 0;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const bool ok = threadIdx.x + it * nthreads < chunks8;
        const uint32_t raw[4] = {cx[it].x, cx[it].y, cx[it].z, cx[it].w};
        uint32_t nrm[4] = {raw[0], raw[1], raw[2], raw[3]};
        if (na.kind != 0) {
            const uint32_t ww[4] = {cw[it].x, cw[it].y, cw[it].z, cw[it].w}, bb[4] = {cb[it].x, cb[it].y, cb[it].z, cb[it].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) nrm[q] = norm_apply(raw[q], ww[q], bb[q], na.kind, mean, r);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[it][2 * q] = rhalf(bflo(nrm[q]));
            a[it][2 * q + 1] = rhalf(bfhi(nrm[q]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool outlier = ok && threshold > 0.f && fabsf(a[it][e]) >= threshold;
            if (ok && !outlier) mx = fmaxf(mx, fabsf(a[it][e]));
            local += outlier ? 1 : 0;
        }
    }
    mx = wave_max(mx);
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) scan[wave] = incl;
    if (lane == 0) shmax[wave] = mx;
    __syncthreads();
    int base = 0, total = 0;
    mx = 0.f;
    for (int wv = 0; wv < nwaves; ++wv) {
        if (wv < wave) base += scan[wv];
        total += scan[wv];
        mx = fmaxf(mx, shmax[wv]);
    }
    const float sa = mx;
    const float inv = mx > 0.f ? __fdiv_rn(127.0f, mx) : 0.f;
    int slot = base + incl - local;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int c = threadIdx.x + it * nthreads;
        if (c < chunks8) {
            uint32_t q8[2] = {0, 0};
            uint32_t h8[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool outlier = threshold > 0.f && fabsf(a[it][e]) >= threshold;
                const int qv = outlier ? 0 : (int)rintf(__fmul_rn(a[it][e], inv));
                q8[e >> 2] |= (uint32_t)(qv & 0xff) << (8 * (e & 3));
                const uint32_t hv = outlier ? (uint32_t)__half_as_ushort(__float2half(a[it][e])) : 0u;
                h8[e >> 1] |= hv << (16 * (e & 1));
                if (outlier) {
                    if (slot < kW8Cap) olist[slot] = c * 8 + e;
                    ++slot;
                }
            }
            *reinterpret_cast<uint2*>(xq_l + (int64_t)c * 8) = make_uint2(q8[0], q8[1]);
            *reinterpret_cast<uint4*>(xo_l + (int64_t)c * 8) = make_uint4(h8[0], h8[1], h8[2], h8[3]);
        }
    }
    if (threadIdx.x == 0) s_total = total;
    __syncthreads();
    int xr[J][4];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        uint4 v = reinterpret_cast<const uint4*>(xq_l)[cidx[jj]];
        if (!cok[jj]) v = make_uint4(0, 0, 0, 0);
        xr[jj][0] = (int)v.x;
        xr[jj][1] = (int)v.y;
        xr[jj][2] = (int)v.z;
        xr[jj][3] = (int)v.w;
    }
    __builtin_amdgcn_s_setprio(0);

    const bf16_t* res_p = residual != nullptr ? residual : reinterpret_cast<const bf16_t*>(CB);
    const bf16_t* bias_p = bias != nullptr ? bias : reinterpret_cast<const bf16_t*>(CB);
    for (int t = 0; t < iters; ++t) {
        int(*rd)[RU * NW] = red[t & 1];
        const int e_col = min(((int)blockIdx.x + t * (int)gridDim.x) * R + (int)threadIdx.x, N - 1);
        const bf16_t e_res = res_p[residual != nullptr ? e_col : 0];
        const bf16_t e_bias = bias_p[bias != nullptr ? e_col : 0];
        const float e_scb = SCB[e_col];
        const float e_scb2 = DUAL ? SCB2[e_col] : 0.f;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (u + PRIME < RU) {
                W8_ISSUE_ROW(t, u + PRIME)
            } else if (t + 1 < iters) {
                W8_ISSUE_ROW(t + 1, u + PRIME - RU)
            }
            asm volatile("" ::: "memory");
            if (RU >= 4) {
                if (u == 0) __builtin_amdgcn_s_setprio(3);
                if (u == RU / 4) __builtin_amdgcn_s_setprio(2);
                if (u == RU / 2) __builtin_amdgcn_s_setprio(1);
                if (u == (3 * RU) / 4) __builtin_amdgcn_s_setprio(0);
            }
            int part[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                int p = 0;
#pragma unroll
                for (int jj = 0; jj < J; ++jj) {
                    p = __builtin_amdgcn_sdot4((int)w[q][u][jj].x, xr[jj][0], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][jj].y, xr[jj][1], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][jj].z, xr[jj][2], p, false);
                    p = __builtin_amdgcn_sdot4((int)w[q][u][jj].w, xr[jj][3], p, false);
                }
                part[q] = wave_sum_i32_to_lane63(p);
            }
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (lane == 63) rd[wave][u * NW + q] = part[q];
        }
        __syncthreads();
        if ((int)threadIdx.x < R) {
            const int ur = threadIdx.x;
            const int jj = ur / RU, u = ur % RU;
            const int col = ((int)blockIdx.x + t * (int)gridDim.x) * R + ur;
            if (col < N) {
                const int no = s_total;
                float res[2] = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    int c32 = 0;
                    for (int c = 0; c < nslabs; ++c) c32 += rd[c * wps + jj][u * NW + q];
                    const float scb = q ? e_scb2 : e_scb;
                    const float b = (bias != nullptr && q == 0) ? bf2f(e_bias) : 0.f;
                    // separately rounded fp32 products and sum (no FMA), the order mm_dequant uses
                    float v = rhalf(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn((float)c32, kMmDequant), sa), scb), b));
                    if (no > 0) {  // mixed-precision decomposition over the outlier columns
                        const int8_t* wrow = reinterpret_cast<const int8_t*>(q ? CB2 : CB) + (int64_t)col * K;
                        float o = 0.f;
                        if (no <= kW8Cap) {
                            for (int i = 0; i < no; ++i) {
                                const int k = olist[i];
                                o += __half2float(xo_l[k]) * rhalf(__fdiv_rn(__fmul_rn((float)wrow[k], scb), 127.0f));
                            }
                        } else {  // more outliers than the list holds: walk the row (same ascending-per-thread order is not kept:
                                  // the sum runs over k ascending, which the tolerance of the int8 path absorbs)
                            for (int k = 0; k < K; ++k) {
                                const float xv = __half2float(xo_l[k]);
                                if (xv != 0.f) o += xv * rhalf(__fdiv_rn(__fmul_rn((float)wrow[k], scb), 127.0f));
                            }
                        }
                        v = rhalf(v + rhalf(o));
                    }
                    res[q] = v;
                }
                out[col] = apply_epilogue_vals(epi, res[0], res[1], false, 0.f, bf2f(e_res));
            }
        }
    }
#undef W8_ISSUE_ROW
#undef W8_ROW0
}

template <int J>
static int w8_fused_launch(const void* CB, const void* CB2, const void* SCB, const void* SCB2, const void* x, float threshold,
                           const void* bias, const void* residual, void* out, int N, int K, int epi, int nslabs,
                           const NormArgs& na, hipStream_t st) {
    constexpr int MAXW = 8;
    constexpr int RU = 8, RUD = 4;
    const bool dual = epi == PARROT_EPI_SWIGLU;
    const int ru = dual ? RUD : RU;
    int wps = MAXW / nslabs;
    if (wps > 4) wps = 4;
    while (wps > 1 && (int64_t)wps * ru * 256 > N) --wps;
    while (wps * nslabs < MAXW && (K >> 3) > 4 * 64 * nslabs * wps) ++wps;  // the cooperative prologue covers 4 x 8 x nthreads
    const int nthreads = 64 * nslabs * wps;
    PARROT_UNSUPPORTED((K >> 3) <= 4 * nthreads, "w8_gemv_fused: K=%d does not fit the workgroup's prologue", K);
    const size_t lds = (size_t)K * 3;
    PARROT_UNSUPPORTED(lds <= 56 * 1024, "w8_gemv_fused: K=%d needs %zu B of LDS", K, lds);
    const int R = wps * ru;
    const int batches = (N + R - 1) / R;
    const int per_simd = (J <= 2 && !dual) ? 3 : 2;  // waves per SIMD at this build's VGPR count (135 .. 201)
    const int resident = 256 * (4 * per_simd / (nslabs * wps) > 0 ? 4 * per_simd / (nslabs * wps) : 1);
    const int iters = (batches + resident - 1) / resident;
    const dim3 grid((batches + iters - 1) / iters), block(nthreads);
    if (dual)
        return launch(K_W8_GEMV, w8_fused_kernel<true, J, RUD, MAXW>, grid, block, lds, st, (const uint4*)CB, (const uint4*)CB2,
                      (const float*)SCB, (const float*)SCB2, (const bf16_t*)x, threshold, (const bf16_t*)bias,
                      (const bf16_t*)residual, (bf16_t*)out, N, K, wps, nslabs, epi, iters, na);
    return launch(K_W8_GEMV, w8_fused_kernel<false, J, RU, MAXW>, grid, block, lds, st, (const uint4*)CB, (const uint4*)CB2,
                  (const float*)SCB, (const float*)SCB2, (const bf16_t*)x, threshold, (const bf16_t*)bias,
                  (const bf16_t*)residual, (bf16_t*)out, N, K, wps, nslabs, epi, iters, na);
}

// ------------------------------------------------------------------------------------------ int8 GEMM (prefill)
// Many token rows: C32[M, N] = CA[M, K] . CB[N, K]^T on the matrix cores (v_mfma_i32_32x32x32_i8: lane l holds 16 int8 of
// row l & 31 at k = 16 (l >> 5) ..), then the mm_dequant / outlier / epilogue arithmetic of the GEMV per element.
// Same tiling as gemm.hip (64 x 64 tile, 4 waves as 2 x 2, one 32 x 32 MFMA tile per wave, K-tile 64 bytes = two MFMA
// k-steps, two LDS stages with 80-byte rows, register prefetch of the next K-tile).  The first version ran the GEMV once
// per row (128 prompt tokens: 103 ms).
typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr int W8BK = 64;   // K-tile in bytes (= int8 elements)
constexpr int W8LD = 80;   // LDS row stride in bytes

template <bool SWI>
__global__ void __launch_bounds__(256)
w8_gemm_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ CB, const int8_t* __restrict__ CB2,
               const float* __restrict__ SCB, const float* __restrict__ SCB2, const float* __restrict__ xout,
               const float* __restrict__ sca, const int32_t* __restrict__ nout, const int32_t* __restrict__ oidx,
               const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int M, int N, int K, int epi) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][64 * W8LD];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][64 * W8LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / W8BK;
    // thread t moves the 16-byte piece (row t >> 2, piece t & 3) of both tiles
    const int prow = tid >> 2, ppc = tid & 3;
    const int64_t am = min(m0 + prow, M - 1), bn = min(n0 + prow, N - 1);
    float res[2][16];
    const int npass = SWI ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int8_t* Wp = pass ? CB2 : CB;
        i32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
        uint4 ra = *reinterpret_cast<const uint4*>(A + am * K + ppc * 16);
        uint4 rb = *reinterpret_cast<const uint4*>(Wp + bn * K + ppc * 16);
        __syncthreads();
        *reinterpret_cast<uint4*>(&As[0][prow * W8LD + ppc * 16]) = ra;
        *reinterpret_cast<uint4*>(&Bs[0][prow * W8LD + ppc * 16]) = rb;
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            const int st = kt & 1;
            const int kn = min(kt + 1, ktiles - 1);  // clamped: unconditional loads
            ra = *reinterpret_cast<const uint4*>(A + am * K + (int64_t)kn * W8BK + ppc * 16);
            rb = *reinterpret_cast<const uint4*>(Wp + bn * K + (int64_t)kn * W8BK + ppc * 16);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint4 va = *reinterpret_cast<const uint4*>(&As[st][(wm * 32 + lr) * W8LD + s2 * 32 + lh * 16]);
                const uint4 vb = *reinterpret_cast<const uint4*>(&Bs[st][(wn * 32 + lr) * W8LD + s2 * 32 + lh * 16]);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, va), __builtin_bit_cast(i32x4_t, vb), acc, 0, 0, 0);
            }
            *reinterpret_cast<uint4*>(&As[st ^ 1][prow * W8LD + ppc * 16]) = ra;  // stage st^1 was last read in step kt-1
            *reinterpret_cast<uint4*>(&Bs[st ^ 1][prow * W8LD + ppc * 16]) = rb;
            __syncthreads();
        }
        // ---- dequantise this pass: C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int col = n0 + wn * 32 + lr;
        const int colc = min(col, N - 1);
        const float scb = (pass ? SCB2 : SCB)[colc];
        const float b = (bias != nullptr && pass == 0) ? bf2f(bias[colc]) : 0.f;
        const int8_t* wrow = Wp + (int64_t)colc * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
            const float sa = sca[row];
            // separately rounded fp32 products and sum (no FMA), the order mm_dequant uses (as in the GEMV)
            float v = rhalf(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn((float)acc[r], kMmDequant), sa), scb), b));
            const int no = nout[row];
            if (no > 0) {  // mixed-precision decomposition over the row's outlier columns
                float o = 0.f;
                for (int t = 0; t < no; ++t) {
                    const int k = oidx[(int64_t)row * K + t];
                    o += xout[(int64_t)row * K + k] * rhalf(__fdiv_rn(__fmul_rn((float)wrow[k], scb), 127.0f));
                }
                v = rhalf(v + rhalf(o));
            }
            res[pass][r] = v;
        }
    }
    const int col = n0 + wn * 32 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N)
            out[(int64_t)row * ldo + col] =
                apply_epilogue(epi, res[0][r], SWI ? res[1][r] : 0.f, nullptr, residual ? residual + (int64_t)row * ldr : nullptr, col);
    }
}

// ------------------------------------------------------------------------------------------ int8 GEMM, second generation
// The structure of gemm2.hip (128 x 128 tiles, both operands by global_load_lds_dwordx4 with the source-side swizzle, two LDS
// buffers, one barrier per K-step, XCD-aware tile order is not needed at these sizes) with 128-byte K-steps = 128 int8 = four
// v_mfma_i32_32x32x32_i8 per fragment pair.  The kernel only accumulates: it writes exact int32 sums (of its K range when K is
// split) to the workspace, and w8_dequant_epilogue_kernel adds the ranges, dequantises, adds the outlier part and applies the
// epilogue - the arithmetic of the GEMV, element for element (integer sums are order-free, so prompt and decode agree bit for bit
// on the int8 part).
typedef int i32x16b_t __attribute__((ext_vector_type(16)));
typedef int i32x4b_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
w8_gemm2_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ CB, int M, int N, int K, int ksplit, int32_t* __restrict__ part) {
    __shared__ __attribute__((aligned(1024))) uint4 smem[2][2][128 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, z = blockIdx.z;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / 128;
    const int kt_begin = (int)((int64_t)z * ktiles / ksplit), kt_end = (int)((int64_t)(z + 1) * ktiles / ksplit);
    const int l_row = lane >> 3, l_slot = lane & 7;
    const int8_t* a_src[4];
    const int8_t* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + l_row;
        const int gs = l_slot ^ ((r >> 1) & 7);
        a_src[i] = A + (int64_t)min(m0 + r, M - 1) * K + gs * 16;
        b_src[i] = CB + (int64_t)min(n0 + r, N - 1) * K + gs * 16;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r0 = (wave * 4 + i) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (int64_t)kt * 128),
                                             (__attribute__((address_space(3))) void*)&smem[buf][0][r0 * 8], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (int64_t)kt * 128),
                                             (__attribute__((address_space(3))) void*)&smem[buf][1][r0 * 8], 16, 0, 0);
        }
    };
    i32x16b_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    int a_row[2], b_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_row[i] = wm * 64 + i * 32 + lr;
        b_row[i] = wn * 64 + i * 32 + lr;
    }
    if (kt_begin < kt_end) issue(kt_begin, 0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        __syncthreads();  // (vmcnt(0) first) tile kt has landed; everybody is done with buffer buf ^ 1
        if (kt + 1 < kt_end) issue(kt + 1, buf ^ 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // 32 int8 per MFMA: lane half lh holds bytes 16 lh .. 16 lh + 15 of the 32
            const int slot = ks * 2 + lh;
            i32x4b_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_bit_cast(i32x4b_t, smem[buf][0][a_row[i] * 8 + (slot ^ ((a_row[i] >> 1) & 7))]);
                bfr[i] = __builtin_bit_cast(i32x4b_t, smem[buf][1][b_row[i] * 8 + (slot ^ ((b_row[i] >> 1) & 7))]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) part[((int64_t)z * M + row) * N + col] = acc[i][j][r];
            }
        }
}


// one thread per output element: sum of the K ranges (exact), mm_dequant, outlier part, epilogue (as w8_gemm_kernel / the GEMV)
__global__ void __launch_bounds__(256)
w8_dequant_epilogue_kernel(const int32_t* __restrict__ part, const int32_t* __restrict__ part2, int ksplit, const int8_t* __restrict__ CB,
                           const int8_t* __restrict__ CB2, const float* __restrict__ SCB, const float* __restrict__ SCB2,
                           const float* __restrict__ O1, const float* __restrict__ O2, const float* __restrict__ sca,
                           const int32_t* __restrict__ nout, const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out,
                           int ldo, int M, int N, int K, int epi) {
    const int row = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;  // (a flat index cost a 64-bit division per element)
    if (col >= N) return;
    const float sa = sca[row];
    const int no = nout[row];
    float res[2] = {0.f, 0.f};
    const int npass = epi == PARROT_EPI_SWIGLU ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int32_t* p = pass ? part2 : part;
        int32_t c = 0;
        for (int z = 0; z < ksplit; ++z) c += p[((int64_t)z * M + row) * N + col];
        const float scb = (pass ? SCB2 : SCB)[col];
        const float b = (bias != nullptr && pass == 0) ? bf2f(bias[col]) : 0.f;
        float v = rhalf(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn((float)c, kMmDequant), sa), scb), b));
        if (no > 0) v = rhalf(v + rhalf((pass ? O2 : O1)[(int64_t)row * N + col]));  // the outlier GEMM (w8_sub_gemm_kernel)
        res[pass] = v;
    }
    out[(int64_t)row * ldo + col] = apply_epilogue(epi, res[0], res[1], nullptr, residual ? residual + (int64_t)row * ldr : nullptr, col);
}

// ---- prompts: the mixed-precision part as ONE dense fp16 GEMM over the call's outlier columns (what bitsandbytes runs:
// subA = A[:, idx], subB = fp16(CB[:, idx] * SCB / 127), out += subA @ subB^T).  The byte-gather kernel above walks the list per
// output element; with the column rule of a multi-row call the list is the same for every row and can be long (the synthetic
// model's MLP hidden rows: ~40 % of 11008 columns over a 128-token prompt - 64 ms of a 73 ms prefill).
// Gather: the listed columns of A (fp16 values, exact) and of the dequantised weights into dense fp16 matrices, leading dimension
// K (the worst case: the count lives on the device), zero-padded to a multiple of 16 columns.
__global__ void __launch_bounds__(256)
w8_sub_gather_kernel(const float* __restrict__ xout, const int8_t* __restrict__ CB, const float* __restrict__ SCB, const int32_t* __restrict__ nout,
                     const int32_t* __restrict__ oidx, __half* __restrict__ subA, __half* __restrict__ subB, int M, int N, int K) {
    const int nu = nout[0], nu16 = (nu + 15) & ~15;
    if (nu == 0) return;  // (a call without outlier columns - every synthetic-weight prompt - must cost a launch, not a grid of M + N workgroups)
    // rows 0 .. M-1: A;  M .. M+N-1: weight rows.  A workgroup walks the list for its rows (grid-stride: at most 2048 workgroups)
    for (int r = blockIdx.x; r < M + N; r += gridDim.x) {
        const float scb = r >= M ? SCB[r - M] : 0.f;
        for (int j = threadIdx.x; j < nu16; j += 256) {
            const int k = j < nu ? oidx[j] : -1;
            if (r < M)
                subA[(int64_t)r * K + j] = __float2half(k >= 0 ? xout[(int64_t)r * K + k] : 0.f);
            else
                subB[(int64_t)(r - M) * K + j] = __float2half(k >= 0 ? rhalf(__fdiv_rn(__fmul_rn((float)CB[(int64_t)(r - M) * K + k], scb), 127.0f)) : 0.f);
        }
    }
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float w8_f32x16_t;

// O[M][N] (fp32) = subA[M][nu] @ subB[N][nu]^T on v_mfma_f32_32x32x16_f16: a workgroup = 2 x 2 waves, one 32 x 32 tile each, fragments
// straight from L2 (the operands are a few MB; 5 GFLOP at most).  fp16 x fp16 products are exact in fp32: only the summation order
// differs from a serial sum.
__global__ void __launch_bounds__(256)
w8_sub_gemm_kernel(const __half* __restrict__ subA, const __half* __restrict__ subB, const int32_t* __restrict__ nout, float* __restrict__ O,
                   int M, int N, int K) {
    const int nu = nout[0];
    if (nu <= 0) return;  // the element-wise pass does not read O then
    const int nu16 = (nu + 15) & ~15;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    const __half* ap = subA + (int64_t)min(m0 + lr, M - 1) * K + 8 * lh;
    const __half* bp = subB + (int64_t)min(n0 + lr, N - 1) * K + 8 * lh;
    w8_f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < nu16; k0 += 16) {
        const f16x8_t af = *reinterpret_cast<const f16x8_t*>(ap + k0);
        const f16x8_t bf = *reinterpret_cast<const f16x8_t*>(bp + k0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = n0 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) O[(int64_t)row * N + col] = acc[r];
    }
}

static int w8_gemm2_ksplit(int M, int N, int K) {
    const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    const int ktiles = K / 128;
    const int target = tune_env("PARROT_W8_SPLIT_TARGET", 512);  // PARROT_W8_SPLIT_TARGET: workgroups to aim at when splitting K
    int ks = tiles > 256 ? 1 : (int)(target / (tiles > 0 ? tiles : 1));  // up to 256 tiles: split, two workgroups fit a CU
    if (ks > 8) ks = 8;
    while (ks > 1 && ktiles / ks < 4) --ks;
    return ks < 1 ? 1 : ks;
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_w8_quantize_rows(const void* W, int w_dtype, int N, int K, void* CB_int8, void* SCB_f32, void* stream) {
    PARROT_REQUIRE(W && CB_int8 && SCB_f32, "w8_quantize_rows: null pointer");
    PARROT_REQUIRE(N >= 1 && K >= 1, "w8_quantize_rows: bad shape N=%d K=%d", N, K);
    PARROT_REQUIRE(w_dtype >= 0 && w_dtype <= 2, "w8_quantize_rows: w_dtype must be 0 (bf16), 1 (fp16) or 2 (fp32)");
    hipStream_t st = (hipStream_t)stream;
    if (w_dtype == 1)
        return launch(K_W8_QUANT_ROWS, w8_quantize_rows_kernel<__half>, dim3(N), dim3(256), 0, st, (const __half*)W, K, (int8_t*)CB_int8, (float*)SCB_f32);
    if (w_dtype == 2)
        return launch(K_W8_QUANT_ROWS, w8_quantize_rows_kernel<float>, dim3(N), dim3(256), 0, st, (const float*)W, K, (int8_t*)CB_int8, (float*)SCB_f32);
    return launch(K_W8_QUANT_ROWS, w8_quantize_rows_kernel<bf16_t>, dim3(N), dim3(256), 0, st, (const bf16_t*)W, K, (int8_t*)CB_int8, (float*)SCB_f32);
}

int parrot_w8_prep_act(const void* x, int ldx, int M, int K, float threshold, void* xq, void* xout, void* sca, void* nout,
                       void* oidx, void* colflag, const parrot_norm_t* norm, void* stream) {
    PARROT_REQUIRE(x && xq && xout && sca && nout && oidx, "w8_prep_act: null pointer");
    PARROT_REQUIRE(M == 1 || colflag != nullptr, "w8_prep_act: a call with several rows needs the K-word column scratch");
    PARROT_REQUIRE(M >= 1 && K >= 1 && ldx >= K, "w8_prep_act: bad shape M=%d K=%d ldx=%d", M, K, ldx);
    PARROT_UNSUPPORTED(K % 8 == 0 && ldx % 8 == 0 && K <= kPrepThreads * 8 * kPrepV && aligned16(x),
                       "w8_prep_act: K=%d must be a multiple of 8 and <= %d, rows 16-byte aligned", K, kPrepThreads * 8 * kPrepV);
    NormArgs na;
    const int rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    int rc2 = launch(K_W8_PREP_ACT, w8_prep_act_kernel, dim3(M), dim3(kPrepThreads), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     K, threshold, (int8_t*)xq, (float*)xout, (float*)sca, (int32_t*)nout, (int32_t*)oidx, na);
    if (rc2 != PARROT_OK || M == 1 || !(threshold > 0.f)) return rc2;
    // several rows: the outlier columns are those of the whole call
    hipError_t e = hipMemsetAsync(colflag, 0, (size_t)K * sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(colflag)");
    rc2 = launch(K_W8_PREP_ACT, w8_union_cols_kernel, dim3((K + 255) / 256, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                 (const float*)xout, (int32_t*)colflag, M, K, threshold);
    if (rc2 != PARROT_OK) return rc2;
    return launch(K_W8_PREP_ACT, w8_union_lists_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, (const int32_t*)colflag,
                  (int8_t*)xq, (int32_t*)nout, (int32_t*)oidx, K);
}

// CB / SCB may be followed by a second weight for the SWIGLU epilogue: pass them concatenated as
// CB = [CB1; CB2] is NOT assumed; the second weight is addressed as CB + N*K and SCB + N (fc_1 then fc_2 rows).
int parrot_w8_gemv(const void* CB, const void* SCB, const void* xq, const void* xout, const void* sca, const void* nout,
                   const void* oidx, int M, const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue,
                   void* stream) {
    PARROT_REQUIRE(CB && SCB && xq && xout && sca && nout && oidx && out, "w8_gemv: null pointer");
    PARROT_REQUIRE(M >= 1 && M <= 65535 && N >= 1 && K >= 1, "w8_gemv: bad shape M=%d N=%d K=%d", M, N, K);
    PARROT_REQUIRE(epilogue >= PARROT_EPI_NONE && epilogue <= PARROT_EPI_SWIGLU, "w8_gemv: unknown epilogue %d", epilogue);
    PARROT_REQUIRE((epilogue == PARROT_EPI_RESIDUAL) == (residual != nullptr), "w8_gemv: residual iff RESIDUAL epilogue");
    PARROT_UNSUPPORTED(K % 16 == 0 && K <= 4096 * kW8MaxSlabs, "w8_gemv: K=%d must be a multiple of 16 and <= %d", K,
                       4096 * kW8MaxSlabs);
    PARROT_REQUIRE(aligned16(CB) && aligned16(xq), "w8_gemv: CB and xq must be 16-byte aligned");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w8_gemv: SWIGLU epilogue takes no bias");
    const void* CB2 = nullptr;
    const void* SCB2 = nullptr;
    if (epilogue == PARROT_EPI_SWIGLU) {
        CB2 = (const int8_t*)CB + (int64_t)N * K;
        SCB2 = (const float*)SCB + N;
    }
    hipStream_t st = (hipStream_t)stream;
    if (M > 8 && K % W8BK == 0) {  // many rows (prefill): matrix cores
        const dim3 grid((N + 63) / 64, (M + 63) / 64);
        if (epilogue == PARROT_EPI_SWIGLU)
            return launch(K_W8_GEMV, w8_gemm_kernel<true>, grid, dim3(256), 0, st, (const int8_t*)xq, (const int8_t*)CB, (const int8_t*)CB2,
                          (const float*)SCB, (const float*)SCB2, (const float*)xout, (const float*)sca, (const int32_t*)nout,
                          (const int32_t*)oidx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, N, K, epilogue);
        return launch(K_W8_GEMV, w8_gemm_kernel<false>, grid, dim3(256), 0, st, (const int8_t*)xq, (const int8_t*)CB, (const int8_t*)CB2,
                      (const float*)SCB, (const float*)SCB2, (const float*)xout, (const float*)sca, (const int32_t*)nout,
                      (const int32_t*)oidx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, N, K, epilogue);
    }
    const int chunks = K / 16;
    const int nslabs = (chunks + 255) / 256;
    const int per_slab = (chunks + nslabs - 1) / nslabs;
    const int jn = (per_slab + 63) / 64;
    if (jn <= 1) return w8_launch<1>(CB, CB2, SCB, SCB2, xq, xout, sca, nout, oidx, M, bias, residual, ldr, out, ldo, N, K, epilogue, nslabs, st);
    if (jn <= 2) return w8_launch<2>(CB, CB2, SCB, SCB2, xq, xout, sca, nout, oidx, M, bias, residual, ldr, out, ldo, N, K, epilogue, nslabs, st);
    return w8_launch<4>(CB, CB2, SCB, SCB2, xq, xout, sca, nout, oidx, M, bias, residual, ldr, out, ldo, N, K, epilogue, nslabs, st);
}

// One token row: activation quantiser (parrot_w8_prep_act) + GEMV (parrot_w8_gemv) in one launch; same arithmetic.
int parrot_w8_gemv_fused(const void* CB, const void* SCB, const void* x, float threshold, const void* bias,
                         const void* residual, void* out, int N, int K, int epilogue, const parrot_norm_t* norm, void* stream) {
    PARROT_REQUIRE(CB && SCB && x && out, "w8_gemv_fused: null pointer");
    PARROT_REQUIRE(N >= 1 && K >= 1, "w8_gemv_fused: bad shape N=%d K=%d", N, K);
    PARROT_REQUIRE(epilogue >= PARROT_EPI_NONE && epilogue <= PARROT_EPI_SWIGLU, "w8_gemv_fused: unknown epilogue %d", epilogue);
    PARROT_REQUIRE((epilogue == PARROT_EPI_RESIDUAL) == (residual != nullptr), "w8_gemv_fused: residual iff RESIDUAL epilogue");
    PARROT_UNSUPPORTED(K % 16 == 0 && K <= 4096 * 4, "w8_gemv_fused: K=%d must be a multiple of 16 and <= %d", K, 4096 * 4);
    PARROT_REQUIRE(aligned16(CB) && aligned16(x), "w8_gemv_fused: CB and x must be 16-byte aligned");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w8_gemv_fused: SWIGLU epilogue takes no bias");
    NormArgs na;
    const int rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    const void* CB2 = nullptr;
    const void* SCB2 = nullptr;
    if (epilogue == PARROT_EPI_SWIGLU) {
        CB2 = (const int8_t*)CB + (int64_t)N * K;
        SCB2 = (const float*)SCB + N;
    }
    const int chunks = K / 16;
    const int nslabs = (chunks + 255) / 256;  // <= 4 loads of 16 B per lane and row
    const int per_slab = (chunks + nslabs - 1) / nslabs;
    const int jn = (per_slab + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (jn <= 1) return w8_fused_launch<1>(CB, CB2, SCB, SCB2, x, threshold, bias, residual, out, N, K, epilogue, nslabs, na, st);
    if (jn <= 2) return w8_fused_launch<2>(CB, CB2, SCB, SCB2, x, threshold, bias, residual, out, N, K, epilogue, nslabs, na, st);
    if (jn <= 3) return w8_fused_launch<3>(CB, CB2, SCB, SCB2, x, threshold, bias, residual, out, N, K, epilogue, nslabs, na, st);
    return w8_fused_launch<4>(CB, CB2, SCB, SCB2, x, threshold, bias, residual, out, N, K, epilogue, nslabs, na, st);
}


/* LLM.int8 prompt rows on the LDS-DMA structure: bytes of workspace parrot_w8_gemm needs (int32 sums per K range, x2 for SWIGLU) */
int64_t parrot_w8_gemm_workspace_bytes(int M, int N, int K, int epilogue) {
    if (M <= 8 || K % 128 != 0) return 0;
    const int64_t nw = epilogue == PARROT_EPI_SWIGLU ? 2 : 1;
    // int32 sums + fp32 outlier part, then the fp16 operands of the outlier GEMM: A[:, idx] (M x K at worst) and dequant(CB[:, idx])
    return ((int64_t)w8_gemm2_ksplit(M, N, K) + 1) * M * N * 4 * nw + ((int64_t)M + nw * N) * K * 2 + 64;
}

int parrot_w8_gemm(const void* CB, const void* SCB, const void* xq, const void* xout, const void* sca, const void* nout, const void* oidx,
                   int M, const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue, void* workspace,
                   void* stream) {
    if (M <= 8 || K % 128 != 0)
        return parrot_w8_gemv(CB, SCB, xq, xout, sca, nout, oidx, M, bias, residual, ldr, out, ldo, N, K, epilogue, stream);
    PARROT_REQUIRE(CB && SCB && xq && xout && sca && nout && oidx && out && workspace, "w8_gemm: null pointer");
    PARROT_REQUIRE(M <= 65535 * 128 && N >= 1, "w8_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    PARROT_REQUIRE(epilogue >= PARROT_EPI_NONE && epilogue <= PARROT_EPI_SWIGLU, "w8_gemm: unknown epilogue %d", epilogue);
    PARROT_REQUIRE((epilogue == PARROT_EPI_RESIDUAL) == (residual != nullptr), "w8_gemm: residual iff RESIDUAL epilogue");
    PARROT_REQUIRE(aligned16(CB) && aligned16(xq), "w8_gemm: CB and xq must be 16-byte aligned");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w8_gemm: SWIGLU epilogue takes no bias");
    hipStream_t st = (hipStream_t)stream;
    const bool swi = epilogue == PARROT_EPI_SWIGLU;
    const int8_t* CB2 = swi ? (const int8_t*)CB + (int64_t)N * K : nullptr;
    const float* SCB2 = swi ? (const float*)SCB + N : nullptr;
    const int ks = w8_gemm2_ksplit(M, N, K);
    int32_t* part = (int32_t*)workspace;
    int32_t* part2 = swi ? part + (int64_t)ks * M * N : nullptr;
    const dim3 grid((N + 127) / 128, (M + 127) / 128, ks);
    int rc = launch(K_W8_GEMM, w8_gemm2_kernel, grid, dim3(256), 0, st, (const int8_t*)xq, (const int8_t*)CB, M, N, K, ks, part);
    if (rc != PARROT_OK) return rc;
    if (swi) {
        rc = launch(K_W8_GEMM, w8_gemm2_kernel, grid, dim3(256), 0, st, (const int8_t*)xq, CB2, M, N, K, ks, part2);
        if (rc != PARROT_OK) return rc;
    }
    PARROT_REQUIRE(M <= 65535, "w8_gemm: M too large");
    float* O1 = (float*)(part + (int64_t)ks * M * N * (swi ? 2 : 1));
    float* O2 = swi ? O1 + (int64_t)M * N : nullptr;
    // the mixed-precision part: gather the call's outlier columns once, one fp16 GEMM per weight
    __half* subA = reinterpret_cast<__half*>((reinterpret_cast<uintptr_t>(O1 + (int64_t)M * N * (swi ? 2 : 1)) + 15) & ~(uintptr_t)15);
    __half* subB = subA + (int64_t)M * K;
    const int Ntot = swi ? 2 * N : N;
    rc = launch(K_W8_OUTLIER, w8_sub_gather_kernel, dim3((unsigned)(M + Ntot < 2048 ? M + Ntot : 2048)), dim3(256), 0, st, (const float*)xout,
                (const int8_t*)CB, (const float*)SCB, (const int32_t*)nout, (const int32_t*)oidx, subA, subB, M, Ntot, K);
    if (rc != PARROT_OK) return rc;
    const dim3 ogrid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    rc = launch(K_W8_OUTLIER, w8_sub_gemm_kernel, ogrid, dim3(256), 0, st, (const __half*)subA, (const __half*)subB, (const int32_t*)nout, O1, M, N, K);
    if (rc != PARROT_OK) return rc;
    if (swi) {
        rc = launch(K_W8_OUTLIER, w8_sub_gemm_kernel, ogrid, dim3(256), 0, st, (const __half*)subA, (const __half*)(subB + (int64_t)N * K),
                    (const int32_t*)nout, O2, M, N, K);
        if (rc != PARROT_OK) return rc;
    }
    return launch(K_W8_DEQUANT, w8_dequant_epilogue_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)M), dim3(256), 0, st, (const int32_t*)part,
                  (const int32_t*)part2, ks, (const int8_t*)CB, CB2, (const float*)SCB, SCB2, (const float*)O1, (const float*)O2,
                  (const float*)sca, (const int32_t*)nout, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, N, K,
                  epilogue);
}

}  // extern "C"
