// Dense bf16 Linear for the decode step: y = epilogue(x @ W^T + b), W (N, K) row-major bf16.
// Reference: torch.nn.Linear on the bf16-true path (lit_gpt/model.py:29,188,190,281-282,293-295).
//
// Memory-bound GEMV.  One wavefront streams one K-slab (<= 4096 k) of a block of output rows:
// each lane loads 16 B (8 bf16) per instruction, 64 lanes = 1 KiB contiguous, up to 8 instructions per
// row; its activations for those k stay in registers for every row.  fp32 accumulation through
// v_dot2c_f32_bf16, wave reduction, cross-slab reduction in LDS, fused epilogue.
#include "parrot_common.h"

namespace parrot {

constexpr int kDenseJ = 8;        // 16-B loads per lane per row (max)
constexpr int kDenseMaxSlabs = 8; // K <= 32768
constexpr int kDenseU = 2;        // rows in flight
constexpr int kDenseMaxRows = 16;

template <int M, bool DUAL, int J>
__global__ void __launch_bounds__(512)
bf16_gemv_kernel(const uint4* __restrict__ W, const uint4* __restrict__ W2, const bf16_t* __restrict__ x, int ldx,
                 const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int N, int K,
                 int rows_per_wg, int epi, int nslabs, NormArgs na) {
    constexpr int NW = DUAL ? 2 : 1;
    __shared__ float red[kDenseMaxSlabs][kDenseMaxRows * M * NW];
    __shared__ float stat[16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = K >> 3;  // 16-B units per row
    const int c0 = (int)((int64_t)wave * chunks / nslabs), c1 = (int)((int64_t)(wave + 1) * chunks / nslabs);

    int cidx[J];
    bool cok[J];
    uint32_t xr[M][J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = c0 + j * 64 + lane;
        cok[j] = c < c1;
        cidx[j] = cok[j] ? c : c1 - 1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            uint4 v = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx)[cidx[j]];
            if (!cok[j]) v = make_uint4(0, 0, 0, 0);
            xr[m][j][0] = v.x;
            xr[m][j][1] = v.y;
            xr[m][j][2] = v.z;
            xr[m][j][3] = v.w;
        }
    }
    if (na.kind != 0) {  // fused RMSNorm / LayerNorm of the input rows (wave-uniform branch)
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float s1 = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) s1 += norm_stat1(xr[m][j][i], na.kind);  // lanes past the slab hold zeros
            s1 = block_sum_waves(s1, stat, nslabs);
            float mean = 0.f, r;
            if (na.kind == 2) {
                mean = s1 / (float)na.d;
                float s2 = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s2 += cok[j] ? norm_stat2(xr[m][j][i], mean) : 0.f;
                r = norm_scale(na, block_sum_waves(s2, stat, nslabs));
            } else {
                r = norm_scale(na, s1);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const uint4 wv = reinterpret_cast<const uint4*>(na.weight)[cidx[j]];
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (na.kind == 2 && na.bias != nullptr) bv = reinterpret_cast<const uint4*>(na.bias)[cidx[j]];
                const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[m][j][i] = cok[j] ? norm_apply(xr[m][j][i], ww[i], bb[i], na.kind, mean, r) : 0u;
            }
        }
    }
    const int r_begin = blockIdx.x * rows_per_wg;
    const int r_end = min(N, r_begin + rows_per_wg);
    const int64_t row16 = chunks;

    for (int r0 = r_begin; r0 < r_end; r0 += kDenseU) {
        // every load is unconditional (lanes past the slab end re-read its last chunk against x = 0):
        // a per-load branch would make hipcc wait vmcnt(0) per element
        uint4 w[NW][kDenseU][J];
#pragma unroll
        for (int u = 0; u < kDenseU; ++u) {
            const int64_t row = min(r0 + u, N - 1);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                w[0][u][j] = load_nt16(W + row * row16 + cidx[j]);
                if (DUAL) w[1][u][j] = load_nt16(W2 + row * row16 + cidx[j]);
            }
        }
#pragma unroll
        for (int u = 0; u < kDenseU; ++u)
#pragma unroll
            for (int q = 0; q < NW; ++q)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const uint4 ww = w[q][u][j];
                        p0 = dot2_bf16(ww.x, xr[m][j][0], p0);
                        p1 = dot2_bf16(ww.y, xr[m][j][1], p1);
                        p0 = dot2_bf16(ww.z, xr[m][j][2], p0);
                        p1 = dot2_bf16(ww.w, xr[m][j][3], p1);
                    }
                    const float v = wave_sum_to_lane63(p0 + p1);
                    if (lane == 63) red[wave][((r0 - r_begin + u) * M + m) * NW + q] = v;
                }
    }
    __syncthreads();
    const int nrows = r_end - r_begin;
    if ((int)threadIdx.x < nrows * M) {
        const int ur = threadIdx.x / M, m = threadIdx.x % M;
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < nslabs; ++c) {
            a0 += red[c][(ur * M + m) * NW];
            if (DUAL) a1 += red[c][(ur * M + m) * NW + 1];
        }
        const int col = r_begin + ur;
        out[(int64_t)m * ldo + col] =
            apply_epilogue(epi, a0, a1, bias, residual ? residual + (int64_t)m * ldr : nullptr, col);
    }
}

template <int M, int J>
static int bf16_gemv_launch_j(const void* W, const void* W2, const void* x, int ldx, const void* bias,
                              const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                              hipStream_t st) {
    const int chunks = K / 8;
    const int nslabs = (chunks + 64 * kDenseJ - 1) / (64 * kDenseJ);
    const int R = N >= 16 * 2048 ? 16 : (N >= 8 * 1024 ? 8 : 4);
    const dim3 grid((N + R - 1) / R), block(64 * nslabs);
    if (epi == PARROT_EPI_SWIGLU)
        return launch(K_BF16_GEMV_DUAL, bf16_gemv_kernel<M, true, J>, grid, block, 0, st, (const uint4*)W, (const uint4*)W2,
                      (const bf16_t*)x, ldx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, N, K,
                      R, epi, nslabs, na);
    return launch(K_BF16_GEMV, bf16_gemv_kernel<M, false, J>, grid, block, 0, st, (const uint4*)W, (const uint4*)W2,
                  (const bf16_t*)x, ldx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, N, K, R,
                  epi, nslabs, na);
}

template <int M>
static int bf16_gemv_launch(const void* W, const void* W2, const void* x, int ldx, const void* bias,
                            const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                            hipStream_t st) {
    const int chunks = K / 8;
    const int nslabs = (chunks + 64 * kDenseJ - 1) / (64 * kDenseJ);
    const int per_slab = (chunks + nslabs - 1) / nslabs;  // 16-B units of the largest slab
    const int jn = (per_slab + 63) / 64;
    if (jn <= 1) return bf16_gemv_launch_j<M, 1>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, st);
    if (jn <= 2) return bf16_gemv_launch_j<M, 2>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, st);
    if (jn <= 4) return bf16_gemv_launch_j<M, 4>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, st);
    return bf16_gemv_launch_j<M, 8>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, st);
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_bf16_gemv(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias,
                     const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue,
                     const parrot_norm_t* norm, void* stream) {
    int rc = check_linear_args("bf16_gemv", W, W2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(K % 8 == 0, "bf16_gemv: K=%d must be a multiple of 8", K);
    PARROT_UNSUPPORTED(K <= 64 * 8 * kDenseJ * kDenseMaxSlabs, "bf16_gemv: K=%d too large", K);
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "bf16_gemv: SWIGLU epilogue takes no bias");
    NormArgs na;
    rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xb = (const bf16_t*)x;
    const bf16_t* rb = (const bf16_t*)residual;
    bf16_t* ob = (bf16_t*)out;
    for (int m0 = 0; m0 < M; m0 += 2) {
        const void* xm = xb + (int64_t)m0 * ldx;
        const void* rm = rb ? rb + (int64_t)m0 * ldr : nullptr;
        void* om = ob + (int64_t)m0 * ldo;
        if (M - m0 >= 2)
            rc = bf16_gemv_launch<2>(W, W2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, st);
        else
            rc = bf16_gemv_launch<1>(W, W2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, st);
        if (rc != PARROT_OK) return rc;
    }
    return PARROT_OK;
}

}  // extern "C"
